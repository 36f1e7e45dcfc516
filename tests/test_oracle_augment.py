"""The numpy restatement of get_rotations / get_unique (oracle/augment_oracle.py)
against outputs of the reference's own functions (tests/golden/augment_S*.npz,
made by tests/golden/make_augment_golden.py from src/tools.py:46-88)."""
import glob
import os

import numpy as np
import pytest

import augment_oracle as A

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "augment_S*.npz")))


def _same(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    for f in ("board", "pi", "r"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f  # bit patterns


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_matches_reference_outputs(path):
    g = np.load(path)
    data = g["data"]
    _same(A.get_rotations(data), g["rotations"])
    _same(A.get_unique(data), g["unique"])
    _same(A.get_unique(g["rotations"]), g["unique_of_rotations"])


def test_fixtures_present():
    assert len(FILES) == 3


def test_symmetries_are_the_dihedral_group():
    # 7 derived maps + identity = 8 distinct permutations, closed under composition
    for S in (5, 6):
        src = [np.arange(S * S)] + A.symmetry_sources(S)
        keys = {tuple(s) for s in src}
        assert len(keys) == 8
        for a in src:
            for b in src:
                assert tuple(a[b]) in keys


def test_f32_sum_matches_numpy_reduce():
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 7, 8, 9, 17, 127, 128, 129, 130, 300, 1000, 5000):
        a = rng.choice(np.array([1.0, -1.0, 1e-3], np.float32), size=n)
        assert A._f32_sum_like_numpy(a).view(np.uint32) == np.add.reduce(a).view(np.uint32), n
        b = rng.random(n).astype(np.float32)
        assert A._f32_sum_like_numpy(b).view(np.uint32) == np.add.reduce(b).view(np.uint32), n
