"""GPU parity of the batched self-play and arena drivers against the
reference-shaped game loops run on the CPU oracle (same evaluator, same noise,
same move rule): identical training records, outcomes and tallies."""
import numpy as np
import pytest

import az_oracle_py as O
from helpers import (ExactTorchEvaluator, exact_eval_numpy, oracle_eval_game, oracle_exact_evaluator,
                     oracle_self_play_game)

pytestmark = pytest.mark.gpu


def _noise(rows, size, alpha, seed):
    return np.random.default_rng(seed).dirichlet(np.full(size, alpha), size=rows)


@pytest.mark.parametrize("S,sims,tau,eps", [(9, 40, -1, 0.25), (7, 30, -1, 0.0), (15, 25, -1, 0.25)])
def test_self_play_games_match_reference_loop(S, sims, tau, eps):
    from alphazero_amd import selfplay as sp

    G, size = 4, S * S
    ev = ExactTorchEvaluator()
    b0 = sp.TreeBatch(G, S, ev, cpuct=4.0, eps=eps, nodes_per_tree=sims * 6)
    b1 = sp.TreeBatch(G, S, ev, cpuct=4.0, eps=eps, nodes_per_tree=sims * 6)
    noise = None
    if eps > 0:
        noise = _noise(sims + 2, size, b0.eng.get_alpha(), 3)
        b0.set_noise_table(noise)
        b1.set_noise_table(noise)
    # give every game its own opening so the four games differ
    for ply in range(2):
        act = np.array([O.opening(size, g * 5 + 1)[ply] for g in range(G)], np.int32)
        b0.make_move(act)
        b1.make_move(act)
    games = sp.self_play_games(b0, b1, sims, tau, rng=np.random.default_rng(0))
    oev = oracle_exact_evaluator(S)
    for g in range(G):
        t0, t1 = O.OracleTree(S, cpuct=4.0), O.OracleTree(S, cpuct=4.0)
        for t in (t0, t1):
            t.set_alpha_default()
            t.set_eps(eps)
            for ply in range(2):
                t.make_move(O.opening(size, g * 5 + 1)[ply])
        recs, n0 = oracle_self_play_game(t0, t1, sims, tau, oev, np.random.default_rng(0), noise)
        got = games[g]
        assert got.dtype == sp.sample_dtype(S)
        assert len(got) == len(recs), (g, len(got), len(recs))
        for k, (board, pi, r) in enumerate(recs):
            assert np.array_equal(got["board"][k], board), (g, k)
            assert np.array_equal(got["pi"][k], pi), (g, k)
            assert got["r"][k] == np.float32(r), (g, k)
        assert float(b0.winners()[g]) == t0.winner
    b0.close()
    b1.close()


def test_self_play_sampled_moves_single_game():
    # temperature phase: moves drawn from pi with the caller's generator
    from alphazero_amd import selfplay as sp

    S, sims, tau = 7, 30, 3
    ev = ExactTorchEvaluator()
    b0 = sp.TreeBatch(1, S, ev, cpuct=4.0, eps=0.0, nodes_per_tree=400)
    b1 = sp.TreeBatch(1, S, ev, cpuct=4.0, eps=0.0, nodes_per_tree=400)
    games = sp.self_play_games(b0, b1, sims, tau, rng=np.random.default_rng(42))
    t0, t1 = O.OracleTree(S, cpuct=4.0), O.OracleTree(S, cpuct=4.0)
    for t in (t0, t1):
        t.set_eps(0.0)
    recs, n0 = oracle_self_play_game(t0, t1, sims, tau, oracle_exact_evaluator(S), np.random.default_rng(42))
    assert len(games[0]) == len(recs)
    for k, (board, pi, r) in enumerate(recs):
        assert np.array_equal(games[0]["pi"][k], pi) and np.array_equal(games[0]["board"][k], board)
        assert games[0]["r"][k] == np.float32(r)
    # sampled plies keep the full distribution, greedy plies are one-hot (src/self_play.py:159-164)
    assert (games[0]["pi"][0] > 0).sum() > 1
    assert (games[0]["pi"][-1] > 0).sum() == 1
    b0.close()
    b1.close()


@pytest.mark.parametrize("eps", [0.0, 0.25])
@pytest.mark.parametrize("driver", ["host", "device"])
def test_eval_games_match_reference_loop(driver, eps):
    """Arena of two different networks; with eps = 0.25 the Dirichlet noise stays on
    as in the reference's arena (src/eval.py:86,92), fed from one fixture table."""
    from alphazero_amd import selfplay as sp

    S, size, G = 7, 49, 6
    sims_b, sims_l = 30, 40

    class Shifted(ExactTorchEvaluator):
        """a second, different 'network': the same function on a shifted key"""

        def __call__(self, boards):
            v, p = super().__call__(boards)
            return (-v).contiguous(), p.flip(1).contiguous()

    def shifted_numpy(b, player):
        v, p = exact_eval_numpy(b)
        return -v, p[::-1].copy()

    best = sp.TreeBatch(G, S, ExactTorchEvaluator(), cpuct=4.0, eps=eps, nodes_per_tree=600)
    latest = sp.TreeBatch(G, S, Shifted(), cpuct=2.0, eps=eps, nodes_per_tree=600)
    noise = None
    if eps > 0:
        noise = _noise(max(sims_b, sims_l) + 2, size, best.eng.get_alpha(), 17)
        best.set_noise_table(noise)
        latest.set_noise_table(noise)
    for ply in range(2):
        act = np.array([O.opening(size, g * 3 + 2)[ply] for g in range(G)], np.int32)
        best.make_move(act)
        latest.make_move(act)
    wb, wl, dr = (sp.eval_games if driver == "host" else sp.eval_games_device)(best, latest, sims_b, sims_l)
    ev_b = oracle_exact_evaluator(S)
    ev_l = O.callback_evaluator(shifted_numpy, S, log_priors=False)
    exp = [0, 0, 0]
    for g in range(G):
        tb, tl = O.OracleTree(S, cpuct=4.0), O.OracleTree(S, cpuct=2.0)
        for t in (tb, tl):
            t.set_alpha_default()
            t.set_eps(eps)
            for ply in range(2):
                t.make_move(O.opening(size, g * 3 + 2)[ply])
        first_player = tb.player
        if g % 2 == 0:
            oracle_eval_game(tb, sims_b, ev_b, tl, sims_l, ev_l, noise)
        else:
            oracle_eval_game(tl, sims_l, ev_l, tb, sims_b, ev_b, noise)
        w = tb.winner
        assert float(best.winners()[g]) == w
        if w == first_player:
            exp[0 if g % 2 == 0 else 1] += 1
        elif w == 1 - first_player:
            exp[1 if g % 2 == 0 else 0] += 1
        else:
            exp[2] += 1
    assert (wb, wl, dr) == tuple(exp)
    best.close()
    latest.close()


def test_cmcts_facade_runs_reference_shaped_game(tmp_path):
    """The drop-in facade with the reference's own call sequence (src/self_play.py:155-186),
    a TorchScript net file as `params`, compared for record shapes and label rules."""
    import torch

    import alphazero_amd.cmcts as cmcts
    from alphazero_amd.model import PolicyValueNet

    S = 9
    torch.manual_seed(0)
    net = PolicyValueNet(S, conv_channels=8, front_layers=2, policy_layers=1, value_layers=1).eval()
    path = str(tmp_path / "net.pt")
    torch.jit.trace(net, torch.rand(1, 2, S, S)).save(path)
    ms = []
    for _ in range(2):
        m = cmcts.mcts(cpuct=4.0, shape=S, nodes_per_tree=2048)
        m.set_alpha_default()
        m.set_threads(0)
        m.set_params(path)
        m.eps = 0.25
        ms.append(m)
    assert cmcts.test(1) == 2 and isinstance(cmcts.version(), str) and cmcts.build_timestamp()
    dt = np.dtype([("board", "f4", (2, S, S)), ("pi", "f4", (S * S,)), ("r", "f4")])
    data = []
    for i in range(6):
        for m in ms:
            m.simulate(20)
            pi = m.get_prob()
            board = m.get_board()
            assert pi.dtype == np.float32 and pi.shape == (S * S,) and abs(pi.sum() - 1) < 1e-3
            assert board.dtype == np.float32 and board.shape == (2, S, S)
            move = pi.argmax()
            data.append(np.array((board, pi, -1), dtype=dt))
            ms[0].make_move(move)
            ms[1].make_move(move)
    assert ms[0].move_cnt == 12 and ms[0].winner == -1 and ms[0].player == 0
    with pytest.raises(RuntimeError, match="Invalid move"):
        ms[0].make_move(int(np.stack(data)["pi"][0].argmax()))
    with pytest.raises(RuntimeError, match="Invalid value"):
        ms[0].cuda = 1
    with pytest.raises(RuntimeError, match="Epsilon must be in range"):
        ms[0].eps = 2
    m = cmcts.mcts(shape=S)
    with pytest.raises(RuntimeError, match="Predictor missing!"):
        m.simulate(3)
    m.set_params(path)
    m.eps = 0.0
    m.simulate(1)
    with pytest.raises(RuntimeError, match="Not a distribution"):
        m.get_prob()
    assert "Move count: 0" in repr(m) and "Moves: \n" in repr(m)
    m.make_movexy(x=2, y=1)
    m.make_move(0)
    # State::repr (state.cpp:113-150): the move list and the board rows
    assert f"Move count: 2\nMoves: {S + 2}, 0, \nBoard size: {S * S}\nBoard:\no _ _" in repr(m)
    m.clear()
    assert "Moves: \n" in repr(m)


# ---- games played by the reference's own caller functions (tests/golden/caller_games.npz)
import os as _os

_CALLER = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "caller_games.npz"))


@pytest.mark.parametrize("name", ["sp0", "sp1", "sp2"])
@pytest.mark.parametrize("driver", ["host", "device"])
def test_drivers_reproduce_reference_self_play_game(name, driver):
    from alphazero_amd import selfplay as sp
    from helpers import LegacyStream

    S = int(_CALLER["shape"])
    sims, tau, seed = (int(x) for x in _CALLER[f"{name}_params"])
    eps = float(_CALLER[f"{name}_eps"])
    ev = ExactTorchEvaluator()
    b0 = sp.TreeBatch(1, S, ev, cpuct=4.0, eps=eps, nodes_per_tree=sims * 8)
    b1 = sp.TreeBatch(1, S, ev, cpuct=4.0, eps=eps, nodes_per_tree=sims * 8)
    if f"{name}_noise" in _CALLER:
        b0.set_noise_table(_CALLER[f"{name}_noise"])
        b1.set_noise_table(_CALLER[f"{name}_noise"])
    play = sp.self_play_games if driver == "host" else sp.self_play_games_device
    games = play(b0, b1, sims, tau, init_moves=[int(m) for m in _CALLER[f"{name}_opening"]], rng=LegacyStream(seed))
    want = _CALLER[f"{name}_data"]
    got = games[0]
    assert got.dtype == want.dtype and len(got) == len(want)
    assert np.array_equal(got["board"], want["board"])
    assert np.array_equal(got["pi"].view(np.uint32), want["pi"].view(np.uint32))
    assert np.array_equal(got["r"], want["r"])
    assert float(b0.winners()[0]) == float(_CALLER[f"{name}_winner"])
    b0.close()
    b1.close()


@pytest.mark.parametrize("name", ["ev0", "ev1"])
def test_eval_games_reproduce_reference_eval_game(name):
    from alphazero_amd import selfplay as sp
    from helpers import replay_board

    S = int(_CALLER["shape"])
    s_first, s_second = (int(x) for x in _CALLER[f"{name}_params"])
    opening = [int(m) for m in _CALLER[f"{name}_opening"]]
    ev = ExactTorchEvaluator()
    best = sp.TreeBatch(1, S, ev, cpuct=4.0, eps=0.0, nodes_per_tree=64 * 8)
    latest = sp.TreeBatch(1, S, ev, cpuct=4.0, eps=0.0, nodes_per_tree=64 * 8)
    tallies = sp.eval_games(best, latest, s_first, s_second, init_moves=opening)  # game 0: `best` opens
    winner = float(_CALLER[f"{name}_winner"])
    assert float(best.winners()[0]) == winner
    first_player = len(opening) % 2
    assert tallies == ((1, 0, 0) if winner == first_player else (0, 1, 0) if winner == 1 - first_player else (0, 0, 1))
    assert np.array_equal(best.eng.get_raw_boards()[0].astype(np.float32),
                          replay_board(S, opening + [int(m) for m in _CALLER[f"{name}_moves"]]))
    best.close()
    latest.close()
