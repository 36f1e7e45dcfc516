"""Continuous self-play (self_play_stream): finished games hand their tree slot
to the next game.  Every game must still be the reference's self_play_game
(src/self_play.py:134-202): checked against the fixed-batch driver when no slot
is reused, against the oracle game loop when games are deterministic, and
through replay properties (one stone per ply, sampled moves have pi > 0, greedy
records are one-hot, labels follow the rules oracle's winner) when sampled
games of different ages share a batch."""
import numpy as np
import pytest

import az_oracle_py as O
from helpers import ExactTorchEvaluator, oracle_exact_evaluator, oracle_self_play_game

pytestmark = pytest.mark.gpu


def _batches(G, S, sims, eps=0.0):
    from alphazero_amd import selfplay as sp

    ev = ExactTorchEvaluator()
    return (sp.TreeBatch(G, S, ev, cpuct=4.0, eps=eps, nodes_per_tree=sims * 6),
            sp.TreeBatch(G, S, ev, cpuct=4.0, eps=eps, nodes_per_tree=sims * 6))


def _same(a, b):
    assert len(a) == len(b)
    for g, (x, y) in enumerate(zip(a, b)):
        assert x.dtype == y.dtype and len(x) == len(y), g
        for f in ("board", "pi", "r"):
            assert np.array_equal(x[f], y[f]), (g, f)


def test_stream_without_reuse_equals_fixed_batch_driver():
    from alphazero_amd import selfplay as sp

    G, S, sims, tau = 5, 7, 30, 3
    b0, b1 = _batches(G, S, sims)
    fixed = sp.self_play_games_device(b0, b1, sims, tau, rng=np.random.default_rng(11))
    b0.close(); b1.close()
    b0, b1 = _batches(G, S, sims)
    stream = sp.self_play_stream(b0, b1, sims, tau, G, rng=np.random.default_rng(11))
    b0.close(); b1.close()
    _same(fixed, stream)


def test_stream_reuses_slots_for_deterministic_games():
    from alphazero_amd import selfplay as sp

    G, S, sims, n_games = 3, 7, 40, 8
    b0, b1 = _batches(G, S, sims)
    games = sp.self_play_stream(b0, b1, sims, -1, n_games, init_moves=(24,), flush_plies=5)
    b0.close(); b1.close()
    t0, t1 = O.OracleTree(S, cpuct=4.0), O.OracleTree(S, cpuct=4.0)
    for t in (t0, t1):
        t.set_alpha_default()
        t.set_eps(0.0)
        t.make_move(24)
    recs, _ = oracle_self_play_game(t0, t1, sims, -1, oracle_exact_evaluator(S), np.random.default_rng(0))
    assert len(games) == n_games
    for g in range(n_games):
        assert len(games[g]) == len(recs), g
        for k, (board, pi, r) in enumerate(recs):
            assert np.array_equal(games[g]["board"][k], board), (g, k)
            assert np.array_equal(games[g]["pi"][k], pi), (g, k)
            assert games[g]["r"][k] == np.float32(r), (g, k)


@pytest.mark.parametrize("eps", [0.0, 0.25])
def test_stream_sampled_games_replay(eps):
    from alphazero_amd import selfplay as sp

    G, S, sims, tau, n_games = 4, 7, 30, 2, 11
    size = S * S
    b0, b1 = _batches(G, S, sims, eps)
    games = sp.self_play_stream(b0, b1, sims, tau, n_games, rng=np.random.default_rng(3), flush_plies=7)
    b0.close(); b1.close()
    assert len(games) == n_games
    distinct = set()
    for g, rec in enumerate(games):
        n1 = len(rec) // 2              # the first player moves first: it has as many records or one more
        n0 = len(rec) - n1
        plies = []                      # (side, record) in the order played
        for k in range(n0):
            plies.append((0, rec[k]))
            if k < n1:
                plies.append((1, rec[n0 + k]))
        tree = O.OracleTree(S, cpuct=4.0)
        absolute = np.zeros((2, size), np.float32)
        moves = []
        for t, (side, r) in enumerate(plies):
            assert tree.winner == -1, (g, t)  # the game went on only while undecided
            seen = r["board"].reshape(2, size)
            mine = seen if side == 0 else seen[::-1]  # plane 0 of a record = side to move
            assert np.array_equal(mine, absolute), (g, t)
            if t + 1 < len(plies):
                nxt_side, nxt = plies[t + 1]
                nb = nxt["board"].reshape(2, size)
                nb = nb if nxt_side == 0 else nb[::-1]
                diff = nb - absolute
                assert diff.sum() == 1 and (diff >= 0).all() and diff[side].sum() == 1, (g, t)
                move = int(diff[side].argmax())
            else:
                assert t // 2 > tau
                move = int(r["pi"].argmax())  # last ply of a game longer than tau pairs: greedy, one-hot
            if t // 2 > tau:
                assert r["pi"][move] == 1 and r["pi"].sum() == 1, (g, t)
            else:
                assert r["pi"][move] > 0 and abs(float(r["pi"].sum()) - 1) < 1e-3, (g, t)
            absolute[side, move] = 1
            tree.make_move(move)
            moves.append(move)
        assert tree.winner != -1, g
        want0 = 1.0 if tree.winner == 0 else (-1.0 if tree.winner == 1 else 1e-3)
        want1 = 1.0 if tree.winner == 1 else (-1.0 if tree.winner == 0 else 1e-3)
        assert np.all(rec["r"][:n0] == np.float32(want0)) and np.all(rec["r"][n0:] == np.float32(want1)), g
        distinct.add(tuple(moves))
    assert len(distinct) > 1  # sampling made the games differ
