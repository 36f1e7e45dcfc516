"""GPU get_rotations / get_unique (alphazero_amd.augment) against the
reference's own outputs (tests/golden/augment_S*.npz) and, on larger random
inputs, against the numpy oracle.  Bit-exact, float32 sums included."""
import glob
import os

import numpy as np
import pytest

import augment_oracle as A

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "augment_S*.npz")))


def _same(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape, (a.shape, b.shape)
    for f in ("board", "pi", "r"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_gpu_matches_reference_outputs(path):
    from alphazero_amd import augment

    g = np.load(path)
    data = g["data"]
    _same(augment.get_rotations(data), g["rotations"])
    _same(augment.get_unique(data), g["unique"])
    _same(augment.get_unique(g["rotations"]), g["unique_of_rotations"])


def _random_samples(S, n, seed, n_boards):
    rng = np.random.default_rng(seed)
    size = S * S
    dt = np.dtype([("board", "f4", (2, S, S)), ("pi", "f4", (size,)), ("r", "f4")])
    pool = np.zeros((n_boards, 2, size), np.float32)
    for k in range(n_boards):
        stones = rng.choice(size, size=rng.integers(0, 12), replace=False)
        for t, a in enumerate(stones):
            pool[k, t % 2, a] = 1
    data = np.zeros(n, dt)
    # heavily skewed: one board (the empty opening) is shared by a third of the records
    pick = np.where(rng.random(n) < 0.33, 0, rng.integers(0, n_boards, size=n))
    pool[0] = 0
    data["board"] = pool[pick].reshape(n, 2, S, S)
    p = rng.random((n, size)).astype(np.float32)
    data["pi"] = p / p.sum(axis=1, keepdims=True)
    data["r"] = rng.choice(np.array([1.0, -1.0, 1e-3], np.float32), size=n)
    return data


@pytest.mark.parametrize("S,n,n_boards", [(15, 3000, 700), (20, 500, 60), (9, 1, 1), (6, 0, 1)])
def test_gpu_matches_oracle_on_skewed_duplicates(S, n, n_boards):
    from alphazero_amd import augment

    data = _random_samples(S, n, 7 * S + n, n_boards)
    _same(augment.get_rotations(data), A.get_rotations(data))
    _same(augment.get_unique(data), A.get_unique(data))


def test_rotations_then_unique_round_trip_property():
    # every orbit of the 8 symmetries collapses to the same set of boards whatever member it starts from
    from alphazero_amd import augment

    data = _random_samples(7, 50, 3, 50)
    a = augment.get_unique(augment.get_rotations(data))
    turned = augment.get_rotations(data)[50:][::7].copy()  # the quarter-turned records
    b = augment.get_unique(augment.get_rotations(turned))
    assert np.array_equal(a["board"], b["board"])
