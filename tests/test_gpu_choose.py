"""GPU move choice / record fields (az_choose_moves), the device-resident
self-play driver and the device Dirichlet table.

Move choice is bit-exact: pi against Cmcts::get_prob, greedy = first argmax,
sampling against numpy.random.choice evaluated for the same uniform number
(and against numpy's own legacy generator where its sum-to-one check lets the
call through).  The Dirichlet generator is the build's own (GSL's stream is not
reproducible): checked against a numpy mirror of the same counter-based
algorithm at 1e-9 and against the distribution's moments."""
import numpy as np
import pytest

import az_oracle_py as O
from helpers import ExactTorchEvaluator, oracle_exact_evaluator, oracle_self_play_game

pytestmark = pytest.mark.gpu


def _engine_with_searches(S, trees, sims, seed=7):
    from alphazero_amd import engine as E

    eng = E.Engine(trees, S, nodes_per_tree=sims + 8, cpuct=4.0)
    eng.set_eps(0.0)
    size = S * S
    for ply in range(3):
        eng.make_move(np.array([O.opening(size, g * 3 + 2)[ply % 2] + (ply // 2) * 2 for g in range(trees)], np.int32) % size)
    eng.simulate_fused(sims, E.EVAL_HASH, 0.0, 0.0, seed)
    return eng


@pytest.mark.parametrize("S,sims", [(7, 60), (15, 200), (20, 90)])
def test_choose_moves_greedy_and_sampled(S, sims):
    import torch

    trees, size = 6, S * S
    eng = _engine_with_searches(S, trees, sims)
    active = np.array([1, 1, 0, 1, 1, 1], np.uint8)
    eng.set_active(active)
    pis = [eng.get_prob(t) for t in range(trees)]
    boards = [eng.get_board(t) for t in range(trees)]
    d_act = torch.full((trees,), -7, dtype=torch.int32, device="cuda")
    d_pi = torch.full((trees, size), -5.0, dtype=torch.float32, device="cuda")
    d_board = torch.full((trees, 2, S, S), -5.0, dtype=torch.float32, device="cuda")

    eng.choose_moves(None, True, d_act.data_ptr(), d_pi.data_ptr(), d_board.data_ptr())
    eng.sync_check()
    act, pi, board = d_act.cpu().numpy(), d_pi.cpu().numpy(), d_board.cpu().numpy()
    for t in range(trees):
        if not active[t]:
            assert act[t] == -1 and (pi[t] == -5.0).all() and (board[t] == -5.0).all()
            continue
        m = int(pis[t].argmax())
        assert act[t] == m
        onehot = np.zeros(size, np.float32)
        onehot[m] = 1
        assert np.array_equal(pi[t], onehot)
        assert np.array_equal(board[t], boards[t])

    rs = np.random.RandomState(11)
    for rep in range(8):
        state = rs.get_state()
        u = rs.random_sample(trees)
        d_u = torch.from_numpy(u).cuda()
        eng.choose_moves(d_u.data_ptr(), False, d_act.data_ptr(), d_pi.data_ptr(), None)
        eng.sync_check()
        act, pi = d_act.cpu().numpy(), d_pi.cpu().numpy()
        for t in range(trees):
            if not active[t]:
                assert act[t] == -1
                continue
            assert np.array_equal(pi[t], pis[t])
            cdf = np.cumsum(pis[t].astype(np.float64))
            cdf /= cdf[-1]
            assert act[t] == np.searchsorted(cdf, u[t], side="right")
            # numpy's own call on the same stream position, when its sum check passes
            one = np.random.RandomState()
            one.set_state(state)
            one.random_sample(t)  # skip the numbers of the trees before this one
            try:
                assert act[t] == one.choice(size, p=pis[t].astype(np.float64))
            except ValueError as e:
                assert "sum to 1" in str(e)
    eng.close()


def test_choose_moves_not_a_distribution():
    import torch
    from alphazero_amd import engine as E

    eng = E.Engine(3, 7, nodes_per_tree=16, cpuct=4.0)
    eng.set_eps(0.0)
    eng.simulate_fused(1, E.EVAL_HASH, 0.0, 0.0, 1)  # expands the root only: nodeN = 0
    d_act = torch.zeros(3, dtype=torch.int32, device="cuda")
    eng.choose_moves(None, True, d_act.data_ptr())
    with pytest.raises(E.EngineError, match="Not a distribution"):
        eng.sync_check()
    eng.close()


@pytest.mark.parametrize("S,sims,tau", [(7, 30, 3), (9, 40, -1)])
def test_device_self_play_matches_host_driver_and_reference_loop(S, sims, tau):
    from alphazero_amd import selfplay as sp

    G, size = 5, S * S
    ev = ExactTorchEvaluator()

    def fresh():
        b0 = sp.TreeBatch(G, S, ev, cpuct=4.0, eps=0.0, nodes_per_tree=sims * 6)
        b1 = sp.TreeBatch(G, S, ev, cpuct=4.0, eps=0.0, nodes_per_tree=sims * 6)
        for ply in range(2):
            act = np.array([O.opening(size, g * 5 + 1)[ply] for g in range(G)], np.int32)
            b0.make_move(act)
            b1.make_move(act)
        return b0, b1

    b0, b1 = fresh()
    host = sp.self_play_games(b0, b1, sims, tau, rng=np.random.default_rng(5))
    w_host = b0.winners().copy()
    b0.close(); b1.close()
    b0, b1 = fresh()
    dev = sp.self_play_games_device(b0, b1, sims, tau, rng=np.random.default_rng(5))
    assert np.array_equal(b0.winners(), w_host)
    b0.close(); b1.close()
    assert len(host) == len(dev) == G
    for g in range(G):
        assert host[g].dtype == dev[g].dtype
        assert len(host[g]) == len(dev[g]), g
        assert np.array_equal(host[g]["board"], dev[g]["board"]), g
        assert np.array_equal(host[g]["pi"], dev[g]["pi"]), g
        assert np.array_equal(host[g]["r"], dev[g]["r"]), g
    if tau < 0:  # greedy games do not consume the generator: game by game against the oracle loop
        oev = oracle_exact_evaluator(S)
        for g in range(G):
            t0, t1 = O.OracleTree(S, cpuct=4.0), O.OracleTree(S, cpuct=4.0)
            for t in (t0, t1):
                t.set_alpha_default()
                t.set_eps(0.0)
                for ply in range(2):
                    t.make_move(O.opening(size, g * 5 + 1)[ply])
            recs, _ = oracle_self_play_game(t0, t1, sims, tau, oev, np.random.default_rng(0))
            assert len(recs) == len(dev[g])
            for k, (board, pi, r) in enumerate(recs):
                assert np.array_equal(dev[g]["board"][k], board)
                assert np.array_equal(dev[g]["pi"][k], pi)
                assert dev[g]["r"][k] == np.float32(r)


# ------------------------------------------------------------ Dirichlet table
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94d049bb133111eb)
    x ^= x >> np.uint64(31)
    return x


def _u01(seed, row, a, j):
    with np.errstate(over="ignore"):
        base = _mix64(_mix64(np.uint64(seed)) ^ ((row.astype(np.uint64) << np.uint64(32)) | a.astype(np.uint64)))
        x = _mix64(base + np.uint64(0x9e3779b97f4a7c15) * (np.asarray(j, np.uint64) + np.uint64(1)))
    return ((x >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _dirichlet_mirror(rows, size, sp, alpha, seed):
    """The generator of k_dirichlet in numpy (same counters, same formulas, same
    order of the row sum: per-thread partial sums, then an xor butterfly)."""
    E = sp // 64
    row, a = np.meshgrid(np.arange(rows), np.arange(size), indexing="ij")
    a1 = alpha + 1.0 if alpha < 1.0 else alpha
    d = a1 - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    g = np.full((rows, size), d)
    done = np.zeros((rows, size), bool)
    j = np.zeros((rows, size), np.uint64)
    for _ in range(64):
        u1, u2, u3 = _u01(seed, row, a, j), _u01(seed, row, a, j + np.uint64(1)), _u01(seed, row, a, j + np.uint64(2))
        x = np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586 * u2)
        v = 1.0 + c * x
        ok = v > 0
        v3 = np.where(ok, v * v * v, 1.0)
        acc = ok & (np.log(u3) < 0.5 * x * x + d - d * v3 + d * np.log(v3)) & ~done
        g = np.where(acc, d * v3, g)
        j = np.where(done | acc, j, j + np.uint64(3))
        done |= acc
        if done.all():
            break
    if alpha < 1.0:
        g = g * np.power(_u01(seed, row, a, np.uint64(0xFFFFFFFF)), 1.0 / alpha)
    pad = np.zeros((rows, sp))
    pad[:, :size] = g
    part = np.zeros((rows, 64))
    for k in range(E):
        part = part + pad.reshape(rows, 64, E)[:, :, k]
    o = 1
    while o < 64:
        part = part + part[:, np.arange(64) ^ o]
        o <<= 1
    return g / part[:, :1]


@pytest.mark.parametrize("S", [7, 15])
def test_device_dirichlet_table(S):
    from alphazero_amd import engine as E

    size = S * S
    eng = E.Engine(2, S, nodes_per_tree=64)
    eng.set_alpha_default()
    alpha = eng.get_alpha()
    rows = 4096
    eng.fill_noise_dirichlet(rows, 1234)
    t = eng.get_noise_table(rows)
    assert t.shape == (rows, size)
    assert np.all(t >= 0) and np.allclose(t.sum(axis=1), 1.0, atol=1e-12)
    # same seed, same table; another seed, another table
    eng.fill_noise_dirichlet(rows, 1234)
    assert np.array_equal(eng.get_noise_table(rows), t)
    eng.fill_noise_dirichlet(rows, 99)
    assert not np.array_equal(eng.get_noise_table(64), t[:64])
    # the numpy mirror of the generator (device libm differs in the last bits)
    e4 = (size + 255) // 256
    sp = ((size + 64 * e4 - 1) // (64 * e4)) * 64 * e4  # padded actions per record (az_kernels.hip)
    ref = _dirichlet_mirror(64, size, sp, alpha, 1234)
    close = np.isclose(t[:64], ref, rtol=1e-9, atol=1e-300)
    assert close.mean() > 0.999, close.mean()
    # moments of Dirichlet(alpha,...,alpha): mean 1/size, var (size-1)/(size^2 (size*alpha+1))
    a0 = size * alpha
    mean, var = 1.0 / size, (size - 1) / (size * size * (a0 + 1))
    assert abs(t.mean() - mean) < 1e-12
    col_mean = t.mean(axis=0)
    assert np.all(np.abs(col_mean - mean) < 6 * np.sqrt(var / rows))
    emp_var = t.var(axis=0).mean()
    assert abs(emp_var - var) / var < 0.05, (emp_var, var)
    # a search consumes the device table like a host one
    eng.set_eps(0.25)
    eng.simulate_fused(32, E.EVAL_HASH, 0.0, 0.0, 3)
    eng.sync_check()
    eng.close()
