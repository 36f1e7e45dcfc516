"""A host in plain C on the C ABI (examples/host_search.c: no Python, no torch, its own HIP runtime
linked in) plays searches and moves; the oracle replays the same games and every number the
program printed must match: the drop-in boundary works for a compiled caller, which is what the
reference's own Cmcts users (mcts/src/main.cpp) are."""
import os
import subprocess

import numpy as np
import pytest

import az_oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "host_search")


def test_c_host_is_built_and_fails_loudly_without_a_gpu():
    import torch

    assert os.path.exists(BIN), "examples/host_search is missing: run __graft_entry__.build()"
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([BIN, "9", "2", "10", "1", "3"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "az_create" in out.stderr and "failed" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("shape,trees,sims,plies,seed", [(15, 5, 120, 4, 11), (9, 3, 60, 6, 2), (20, 2, 200, 2, 77)])
def test_c_host_matches_the_oracle(shape, trees, sims, plies, seed):
    out = subprocess.run([BIN, str(shape), str(trees), str(sims), str(plies), str(seed)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    size = shape * shape
    ev = O.hash_evaluator(seed)
    ots = []
    for t in range(trees):
        ot = O.OracleTree(shape, cpuct=4.0)
        ot.set_eps(0.0)
        ot.make_move((7 * t + 3) % size)
        ots.append(ot)
    rows = iter(lines)
    for ply in range(plies):
        for t, ot in enumerate(ots):
            ot.simulate(sims, ev)
            counts = ot.root_counts()
            move = int(counts.argmax())
            want = [ply, t, int(ot.root_n()), move, int(((np.arange(size) + 1) * counts).sum())]
            assert [int(x) for x in next(rows).split()] == want
        for ot in ots:
            ot.make_move(int(ot.root_counts().argmax()))
    stats = next(rows).split()
    assert stats[0] == "stats"
    want = [sum(ot.stats()[k] for ot in ots) for k in ("sims", "expansions", "terminals", "select_calls")]
    assert [int(x) for x in stats[1:]] == want
    assert next(rows).startswith("version ")
