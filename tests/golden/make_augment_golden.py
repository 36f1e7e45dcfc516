#!/usr/bin/env python3
"""Generates tests/golden/augment_S<shape>.npz from the reference's own
`get_rotations` / `get_unique` (/root/reference/src/tools.py:46-88).

Run in the build container only (the reference does not travel):
    python tests/golden/make_augment_golden.py
tools.py does `import cmcts`; this build's drop-in module of that name
(alphazero_amd.cmcts) is what it gets.  The fixtures hold inputs and the
reference's outputs, nothing of its source.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
import alphazero_amd.cmcts as _cmcts  # noqa: E402

sys.modules["cmcts"] = _cmcts
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/src")
import tools  # noqa: E402  (the reference module)


def make_samples(shape, n, seed, dup_groups):
    """Random sparse positions with float32 pi / r; `dup_groups` lists group sizes of
    boards repeated with different pi and r (the duplicate-averaging case)."""
    rng = np.random.default_rng(seed)
    size = shape * shape
    dt = np.dtype([("board", "f4", (2, shape, shape)), ("pi", "f4", (size,)), ("r", "f4")])
    data = np.zeros(n, dt)
    for k in range(n):
        stones = rng.choice(size, size=rng.integers(0, min(size, 9)), replace=False)
        for t, a in enumerate(stones):
            data["board"][k, t % 2].reshape(-1)[a] = 1
        p = rng.random(size).astype(np.float32)
        data["pi"][k] = p / p.sum()
        data["r"][k] = rng.choice(np.array([1.0, -1.0, 1e-3], np.float32))
    pos = 0
    for g in dup_groups:
        idx = rng.choice(n, size=g, replace=False)
        data["board"][idx] = data["board"][idx[0]]
        pos += g
    return data


def main():
    for shape, n, groups in ((5, 40, (2, 3, 5)), (7, 300, (2, 2, 3, 9, 17, 140)), (13, 64, (2, 4))):
        data = make_samples(shape, n, 100 + shape, groups)
        rot = tools.get_rotations(data)
        uniq = tools.get_unique(data)
        uniq_rot = tools.get_unique(rot)
        np.savez_compressed(os.path.join(HERE, f"augment_S{shape}.npz"),
                            data=data, rotations=rot, unique=uniq, unique_of_rotations=uniq_rot)
        print(shape, len(data), len(rot), len(uniq), len(uniq_rot))


if __name__ == "__main__":
    main()
