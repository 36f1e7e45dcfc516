"""GPU search parity: fused and stepped MCTS kernels against the CPU oracle.

Bar (north star): with fixed priors and fixed noise, visit counts, selected
moves and terminal outcomes bit-exact; value sums W compared bit-exact as well
(stricter than the 1e-5 relative bound asked for Q = W/N), priors P bit-exact.
"""
import numpy as np
import pytest

import az_oracle_py as O

pytestmark = pytest.mark.gpu


def _openings(size, n):
    return [O.opening(size, g) for g in range(n)]


def _noise_table(rows, size, alpha, seed):
    rng = np.random.default_rng(seed)
    return rng.dirichlet(np.full(size, alpha), size=rows)


def _compare_roots(eng, trees, tag):
    counts, root_n = eng.get_counts()
    for t, ot in enumerate(trees):
        assert root_n[t] == ot.root_n(), (tag, t, root_n[t], ot.root_n())
        assert np.array_equal(counts[t].astype(np.int64), ot.root_counts()), (tag, t)
        node = eng.get_node(t, [])
        if ot.root_n() >= 0:
            assert node is not None
            assert np.array_equal(node[2], ot.root_w()), (tag, t, "W")
            assert np.array_equal(node[3], ot.root_p()), (tag, t, "P")
    return counts


def _play_fused(S, B, sims, plies, ev_kind, eps, cpuct, threads=0, nodes=None, seed=7):
    from alphazero_amd import engine as E

    size = S * S
    eng = E.Engine(B, S, nodes_per_tree=nodes or (sims * 4 + 64), cpuct=cpuct)
    eng.set_alpha_default()
    eng.set_eps(eps)
    eng.set_threads(threads)
    noise = None
    if eps > 0:
        noise = _noise_table(sims + 8, size, eng.get_alpha(), seed)
        eng.set_noise_table(noise)
    if ev_kind == "uniform":
        ev = O.uniform_evaluator(np.float32(1.0 / size))
        fused = dict(eval_kind=E.EVAL_UNIFORM, const_prior=float(np.float32(1.0 / size)), const_value=0.0)
    else:
        ev = O.hash_evaluator(1234)
        fused = dict(eval_kind=E.EVAL_HASH, hash_seed=1234)
    trees = []
    for g in range(B):
        t = O.OracleTree(S, cpuct=cpuct)
        t.set_alpha_default()
        t.set_eps(eps)
        t.set_threads(threads)
        trees.append(t)
    # distinct two-ply openings so the trees diverge
    for ply in range(2):
        act = np.array([O.opening(size, g)[ply] for g in range(B)], np.int32)
        eng.make_move(act)
        for t, a in zip(trees, act):
            t.make_move(int(a))
    finished = np.zeros(B, bool)
    for ply in range(plies):
        eng.set_active((~finished).astype(np.uint8))
        eng.simulate_fused(sims, **fused)
        for t, ot in enumerate(trees):
            if not finished[t]:
                ot.simulate(sims, ev, noise)
        counts = _compare_roots(eng, trees, f"ply {ply}")
        # a deeper node too: the most visited child of tree 0
        a0 = int(counts[0].argmax())
        on = trees[0].node_at([a0])
        gn = eng.get_node(0, [a0])
        assert (on is None) == (gn is None)
        if on is not None:
            assert gn[0] == on[0]
            assert np.array_equal(gn[1].astype(np.int64), on[1])
            assert np.array_equal(gn[2], on[2]) and np.array_equal(gn[3], on[3])
        act = np.full(B, -1, np.int32)
        for t in range(B):
            if not finished[t]:
                act[t] = int(counts[t].argmax())  # first maximum, like pi.argmax()
        eng.make_move(act)
        for t, ot in enumerate(trees):
            if act[t] >= 0:
                ot.make_move(int(act[t]))
        pl, mc, wi = eng.get_states()
        for t, ot in enumerate(trees):
            assert (pl[t], mc[t], float(wi[t])) == (ot.player, ot.move_cnt, ot.winner), (ply, t)
            if ot.winner != -1:
                finished[t] = True
        if finished.all():
            break
    # counters agree with the oracle's
    st = eng.get_stats()
    tot = dict(sims=0, expansions=0, terminals=0, select_calls=0)
    for ot in trees:
        for k, v in ot.stats().items():
            tot[k] += v
    assert st == tot
    eng.close()
    return st


@pytest.mark.parametrize("S", [15, 13, 20, 9])
def test_fused_uniform_eps0(S):
    _play_fused(S, B=6, sims=200, plies=6, ev_kind="uniform", eps=0.0, cpuct=4.0)


@pytest.mark.parametrize("S", [15, 20, 7])
def test_fused_hash_eps0(S):
    st = _play_fused(S, B=6, sims=300, plies=8, ev_kind="hash", eps=0.0, cpuct=4.0)
    assert st["select_calls"] / st["sims"] > 1.5  # the hash evaluator drives the search deeper


@pytest.mark.parametrize("S,ev", [(15, "uniform"), (15, "hash"), (13, "hash")])
def test_fused_with_noise(S, ev):
    _play_fused(S, B=5, sims=200, plies=6, ev_kind=ev, eps=0.25, cpuct=4.0)


def test_fused_800_sims_config2_sample():
    # the headline configuration's arithmetic on a sample of trees: 800 sims/move
    _play_fused(15, B=4, sims=800, plies=5, ev_kind="uniform", eps=0.0, cpuct=4.0)
    _play_fused(15, B=3, sims=800, plies=4, ev_kind="hash", eps=0.25, cpuct=4.0)


@pytest.mark.timeout(400)
def test_fused_config2_full_size_every_tree():
    # the headline configuration at its full size: all 1024 trees x 800 sims/move, every
    # tree's root (N, W, P, nodeN), states and the engine counters against the oracle
    _play_fused(15, B=1024, sims=800, plies=6, ev_kind="uniform", eps=0.0, cpuct=4.0, nodes=4096)


@pytest.mark.timeout(400)
def test_fused_20x20_at_8192_trees():
    # one GPU's share of the 20x20 configuration: 8192 concurrent trees, noise at every expansion
    _play_fused(20, B=8192, sims=100, plies=3, ev_kind="hash", eps=0.25, cpuct=4.0, nodes=480)


def test_fused_paths_deeper_than_64_levels():
    # cpuct = 0 with a constant value: every score ties, the first legal action wins at every
    # level and each search extends one line by a node -- down to level 80 on 20x20 (the
    # part of a search path beyond 64 levels is kept in LDS instead of registers)
    st = _play_fused(20, B=2, sims=500, plies=2, ev_kind="uniform", eps=0.0, cpuct=0.0)
    assert st["select_calls"] / st["sims"] > 30


def test_fused_more_visits_than_the_sqrt_table_holds():
    # nodeN beyond the 2048-entry table of sqrt(nodeN + 1e-8): the operation itself takes over
    _play_fused(9, B=2, sims=2600, plies=2, ev_kind="hash", eps=0.0, cpuct=4.0, nodes=2600 * 3)


@pytest.mark.parametrize("lean", ["0", "1", "2"])
@pytest.mark.parametrize("S,B,sims,plies,ev,eps,cpuct", [(15, 10, 300, 4, "uniform", 0.25, 4.0), (15, 10, 300, 4, "hash", 0.0, 4.0),
                                                       (20, 6, 200, 3, "hash", 0.25, 4.0), (7, 8, 150, 30, "hash", 0.0, 2.0),
                                                       (15, 3, 800, 3, "uniform", 0.0, 4.0), (20, 2, 300, 2, "uniform", 0.0, 0.0)])
def test_fused_kernel_forms_match_the_oracle(S, B, sims, plies, ev, eps, cpuct, lean, monkeypatch):
    """The launcher picks the fused kernel's form by evaluator and tree count (early child load + large
    sqrt table for about one wavefront per SIMD, the lean form for many trees, the compact form -- 128 registers,
    four wavefronts per SIMD -- for very many); AZ_FUSED_LEAN forces one: every form against the oracle on either evaluator, with and without noise, past the lean
    form's 512-entry sqrt table, down a path deeper than 64 levels and through finished games."""
    monkeypatch.setenv("AZ_FUSED_LEAN", lean)
    _play_fused(S, B=B, sims=sims, plies=plies, ev_kind=ev, eps=eps, cpuct=cpuct)


def test_fused_small_board_to_the_end():
    # 6x6 / 7x7 games run into terminals, draws and full boards inside the search
    for S in (6, 7):
        _play_fused(S, B=8, sims=150, plies=S * S, ev_kind="hash", eps=0.0, cpuct=2.0)


def test_fused_thread_fanout_rounding():
    # threads > 1: per-worker count rounds up and the noise stream restarts per worker
    # (cmcts.cpp:98-119,132-133); workers run back to back, so compare with the
    # oracle run as `workers` consecutive single-thread calls
    from alphazero_amd import engine as E

    S, size, sims, threads = 9, 81, 50, 4
    per = sims // threads + 1  # 13 -> 52 searches
    noise = _noise_table(per + 2, size, 0.1, 3)
    eng = E.Engine(2, S, nodes_per_tree=400, cpuct=3.0)
    eng.set_eps(0.25)
    eng.set_threads(threads)
    eng.set_noise_table(noise)
    eng.simulate_fused(sims, eval_kind=E.EVAL_HASH, hash_seed=9)
    ot = O.OracleTree(S, cpuct=3.0)
    ot.set_eps(0.25)
    ev = O.hash_evaluator(9)
    for _ in range(threads):
        ot.simulate(per, ev, noise)
    counts, root_n = eng.get_counts()
    assert root_n[0] == ot.root_n() == threads * per - 1
    assert np.array_equal(counts[0].astype(np.int64), ot.root_counts())
    assert np.array_equal(counts[1], counts[0])
    eng.close()


def test_get_prob_and_errors():
    from alphazero_amd import engine as E

    eng = E.Engine(2, 15, nodes_per_tree=64, cpuct=4.0)
    eng.set_eps(0.0)
    p15 = float(np.float32(1 / 225))
    # simulate(1) on a fresh root only expands it: nodeN == 0 -> "Not a distribution"
    eng.simulate_fused(1, const_prior=p15)
    with pytest.raises(E.EngineError) as ei:
        eng.get_prob(0)
    assert ei.value.code == E.ERR_NOT_DIST and "Not a distribution" in str(ei.value)
    eng.simulate_fused(40, const_prior=p15)
    ot = O.OracleTree(15, cpuct=4.0)
    ot.set_eps(0.0)
    ev = O.uniform_evaluator(np.float32(1 / 225))
    ot.simulate(1, ev)
    ot.simulate(40, ev)
    assert np.array_equal(eng.get_prob(0), ot.get_prob())
    with pytest.raises(E.EngineError) as ei:
        eng.simulate_fused(0)
    assert ei.value.code == E.ERR_BAD_SIMS
    with pytest.raises(E.EngineError) as ei:
        eng.set_eps(1.5)
    assert ei.value.code == E.ERR_EPS_RANGE
    with pytest.raises(E.EngineError) as ei:
        eng.set_cuda(1)
    assert ei.value.code == E.ERR_INVALID_VALUE and "Invalid value" in str(ei.value)
    with pytest.raises(E.EngineError) as ei:
        eng.make_movexy(0, 15, 0)
    assert ei.value.code == E.ERR_BAD_XY
    # node pool exhaustion is reported, not silently ignored
    with pytest.raises(E.EngineError) as ei:
        eng.simulate_fused(200, const_prior=p15)
    assert ei.value.code == E.ERR_POOL_FULL
    eng.close()


def test_clear_and_reuse():
    from alphazero_amd import engine as E

    eng = E.Engine(3, 15, nodes_per_tree=300, cpuct=4.0)
    eng.set_eps(0.0)
    p15 = float(np.float32(1 / 225))
    eng.simulate_fused(100, const_prior=p15)
    c1, n1 = eng.get_counts()
    eng.make_move(np.array([3, 4, 5], np.int32))
    eng.clear(1)
    pl, mc, wi = eng.get_states()
    assert mc.tolist() == [1, 0, 1] and pl.tolist() == [1, 0, 1]
    eng.clear()
    used, mx = eng.pool_usage(0)
    assert mx == 0
    eng.simulate_fused(100, const_prior=p15)
    c2, n2 = eng.get_counts()
    assert np.array_equal(c1, c2) and np.array_equal(n1, n2)
    eng.close()


def test_restart_finished_on_device_equals_host_clear_and_openings():
    """az_make_move_greedy and az_restart_finished (greedy move, then clear + opening
    moves of finished games, without a host round trip) against argmax +
    az_make_move + az_clear_masked from the host on a twin engine."""
    import torch
    from alphazero_amd import engine as E

    S, B, sims = 6, 24, 40
    size = S * S
    op = np.array([[O.opening(size, g)[ply] for g in range(B)] for ply in range(2)], np.int32)
    engines = [E.Engine(B, S, nodes_per_tree=2000, cpuct=2.0) for _ in range(2)]
    for eng in engines:
        eng.set_eps(0.0)
        for ply in range(2):
            eng.make_move(op[ply])
    d_open = torch.from_numpy(op).cuda()
    d_count = torch.zeros(1, dtype=torch.int32, device="cuda")
    restarted = 0
    for ply in range(3 * size):
        states = []
        for k, eng in enumerate(engines):
            eng.simulate_fused(sims, eval_kind=E.EVAL_HASH, hash_seed=21)
            counts, root_n = eng.get_counts()
            if k == 0:
                eng.make_move_greedy()  # first maximum of the visit counts, chosen on the GPU
                eng.restart_finished(d_open.data_ptr(), 2, d_count.data_ptr())
                eng.sync_check()
            else:
                eng.make_move(counts.argmax(axis=1).astype(np.int32))
                fin = eng.get_states()[2] != -1.0
                restarted += int(fin.sum())
                if fin.any():
                    eng.clear_masked(fin.astype(np.uint8))
                    for p in range(2):
                        eng.make_move(np.where(fin, op[p], -1).astype(np.int32))
            states.append((counts, root_n) + tuple(eng.get_states()))
        for a, b in zip(*states):
            assert np.array_equal(a, b), ply
    assert restarted > B  # every slot has seen more than one game
    assert int(d_count.item()) == restarted
    for eng in engines:
        eng.close()


def test_noise_table_too_short_is_reported():
    """Every expansion of a simulate() call takes the next row of the noise table
    (node.cpp:38 draws a fresh Dirichlet vector each time): running out of rows is an
    error, for the fused and for the stepped search, and the tree keeps its last valid state."""
    from alphazero_amd import engine as E

    S, size = 7, 49
    noise = _noise_table(10, size, 0.3, 1)
    for stepped in (False, True):
        eng = E.Engine(2, S, nodes_per_tree=200, cpuct=4.0)
        eng.set_eps(0.25)
        eng.set_noise_table(noise)
        with pytest.raises(E.EngineError) as ei:
            if stepped:
                import torch
                from test_gpu_step import _step_sims
                _step_sims(eng, torch, 30, lambda b: O.hash_eval(S, 3, b))
            else:
                eng.simulate_fused(30, eval_kind=E.EVAL_HASH, hash_seed=3)
        assert ei.value.code == E.ERR_NOISE_ROWS
        eng.close()


@pytest.mark.timeout(400)
def test_fused_random_configurations():
    """Differential sweep: board edge, search length, evaluator, noise share and cpuct
    drawn at random (fixed seed); every tree's root statistics, states and the
    counters against the oracle, as in the fixed cases above.  (Single worker only:
    with threads > 1 the oracle, like the reference, runs real concurrent threads and
    is not deterministic; that mode's arithmetic is covered by
    test_fused_thread_fanout_rounding.)"""
    rng = np.random.default_rng(20240611)
    for _ in range(60):
        S = int(rng.integers(5, 21))
        sims = int(rng.integers(20, 220))
        _play_fused(S, B=3, sims=sims, plies=int(rng.integers(2, 6)),
                    ev_kind=str(rng.choice(["uniform", "hash"])), eps=float(rng.choice([0.0, 0.1, 0.25, 1.0])),
                    cpuct=float(rng.choice([0.0, 0.5, 4.0, 10.0])), nodes=sims * 8 + 64,
                    seed=int(rng.integers(1, 1000)))


def test_timing_slots_do_not_block():
    """az_timing_mark / az_timing_elapsed: start and end events of several launches are
    recorded without waiting, the times are read afterwards."""
    from alphazero_amd import engine as E

    eng = E.Engine(64, 9, nodes_per_tree=400, cpuct=4.0)
    eng.set_eps(0.0)
    eng.simulate_fused(5, eval_kind=E.EVAL_HASH, hash_seed=1)  # first launch pays for loading the code object
    for slot, sims in enumerate((20, 200)):
        eng.timing_mark(slot, 0)
        eng.simulate_fused(sims, eval_kind=E.EVAL_HASH, hash_seed=1, sync=False)
        eng.timing_mark(slot, 1)
    eng.sync_check()
    short, long_ = eng.timing_elapsed(0), eng.timing_elapsed(1)
    assert 0 < short < long_
    with pytest.raises(E.EngineError):
        eng.timing_elapsed(7)  # never marked
    eng.close()
