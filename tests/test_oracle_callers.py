"""The loop restatements the GPU driver tests are checked with
(tests/helpers.py: oracle_self_play_game, oracle_eval_game) against games
played by the reference's own self_play_game / eval_game over the same oracle
trees (tests/golden/caller_games.npz, made by tests/golden/make_caller_golden.py)."""
import os

import numpy as np
import pytest

import az_oracle_py as O
from helpers import LegacyStream, oracle_eval_game, oracle_exact_evaluator, oracle_self_play_game

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "caller_games.npz"))
S = int(G["shape"])


def _tree(eps, opening):
    t = O.OracleTree(S, cpuct=4.0)
    t.set_alpha_default()
    t.set_eps(eps)
    for m in opening:
        t.make_move(int(m))
    return t


@pytest.mark.parametrize("name", ["sp0", "sp1", "sp2"])
def test_self_play_restatement_matches_reference_function(name):
    sims, tau, seed = (int(x) for x in G[f"{name}_params"])
    eps = float(G[f"{name}_eps"])
    noise = G[f"{name}_noise"] if f"{name}_noise" in G else None
    t0, t1 = _tree(eps, G[f"{name}_opening"]), _tree(eps, G[f"{name}_opening"])
    recs, n0 = oracle_self_play_game(t0, t1, sims, tau, oracle_exact_evaluator(S), LegacyStream(seed), noise)
    want = G[f"{name}_data"]
    assert len(recs) == len(want)
    for k, (board, pi, r) in enumerate(recs):
        assert np.array_equal(board, want["board"][k]), k
        assert np.array_equal(pi.view(np.uint32), want["pi"][k].view(np.uint32)), k
        assert np.float32(r) == want["r"][k], k
    assert t0.winner == float(G[f"{name}_winner"])


@pytest.mark.parametrize("name", ["ev0", "ev1"])
def test_eval_restatement_matches_reference_function(name):
    s_first, s_second = (int(x) for x in G[f"{name}_params"])
    first, second = _tree(0.0, G[f"{name}_opening"]), _tree(0.0, G[f"{name}_opening"])
    ev = oracle_exact_evaluator(S)
    oracle_eval_game(first, s_first, ev, second, s_second, ev)
    assert first.winner == float(G[f"{name}_winner"])
    assert first.move_cnt == len(G[f"{name}_opening"]) + len(G[f"{name}_moves"])
    from helpers import replay_board
    assert np.array_equal(first.raw_board().astype(np.float32),
                          replay_board(S, list(G[f"{name}_opening"]) + list(G[f"{name}_moves"])))
