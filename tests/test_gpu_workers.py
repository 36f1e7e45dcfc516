"""threads > 1 as concurrent workers in lock step (az_set_schedule, az_select_workers /
az_expand_backup_workers, az_simulate_net) against the oracle's same-schedule mode
(azo_set_schedule(LOCKSTEP): one legal interleaving of the reference's std::threads,
cmcts.cpp:98-119): visit counts, value sums, priors, nodeN and the counters bit-exact, through
re-rooting, with workers that meet in one unexpanded node, with the per-worker count rounding and
the per-worker noise stream."""
import numpy as np
import pytest

import az_oracle_py as O
from helpers import ExactTorchEvaluator, oracle_exact_evaluator

pytestmark = pytest.mark.gpu


def _oracle(S, T, eps):
    t = O.OracleTree(S, cpuct=4.0)
    t.set_alpha_default()
    t.set_eps(eps)
    t.set_threads(T)
    t.set_schedule(1)
    return t


def _compare(batch, trees, paths=((),)):
    counts, root_n = batch.eng.get_counts()
    for g, ot in enumerate(trees):
        assert root_n[g] == ot.root_n(), g
        assert np.array_equal(counts[g].astype(np.int64), ot.root_counts()), g
        for path in paths:
            got, want = batch.eng.get_node(g, list(path)), ot.node_at(list(path))
            assert (got is None) == (want is None), (g, path)
            if got is not None:
                assert got[0] == want[0], (g, path)                      # nodeN (visit deficit included)
                assert np.array_equal(got[1].astype(np.int64), want[1]), (g, path)
                assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), (g, path)
    return counts


@pytest.mark.parametrize("S,T,sims", [(9, 2, 31), (9, 4, 50), (9, 8, 3), (15, 4, 100), (15, 8, 40), (15, 2, 64),
                                      (20, 4, 60), (20, 8, 33), (20, 2, 25),
                                      # the most workers the engine takes, a count that ends inside a chunk of eight, and
                                      # more workers than the board has cells (terminal leaves, crowds in one new node)
                                      (9, 64, 200), (7, 17, 60), (5, 64, 130), (15, 64, 64), (20, 33, 70)])
@pytest.mark.parametrize("eps", [0.0, 0.25])
def test_concurrent_workers_match_the_oracle_lockstep(S, T, sims, eps):
    from alphazero_amd import selfplay as sp

    G, plies, size = 5, 4, S * S
    batch = sp.TreeBatch(G, S, ExactTorchEvaluator(), cpuct=4.0, eps=eps, threads=T, nodes_per_tree=(sims + 2 * T) * plies + 64)
    noise = np.random.default_rng(S + T).dirichlet(np.full(size, batch.eng.get_alpha()), size=sims + 2)
    if eps:
        batch.set_noise_table(noise)
    trees = [_oracle(S, T, eps) for _ in range(G)]
    first = np.array([O.opening(size, 5 * g + 1)[0] for g in range(G)], np.int32)
    batch.make_move(first)
    for ot, a in zip(trees, first):
        ot.make_move(int(a))
    oev = oracle_exact_evaluator(S)
    for ply in range(plies):
        batch.simulate(sims)
        for ot in trees:
            ot.simulate(sims, oev, noise if eps else None)
        counts = _compare(batch, trees)
        best = counts.argmax(axis=1)
        if not counts.any():  # n < threads on a fresh root: every worker only met the unexpanded root
            raw = batch.eng.get_raw_boards().reshape(G, 2, size).sum(axis=1)
            best = (raw == 0).argmax(axis=1)
        for g, ot in enumerate(trees):
            a = int(best[g])
            got, want = batch.eng.get_node(g, [a]), ot.node_at([a])
            assert (got is None) == (want is None)
            if got is not None:
                assert got[0] == want[0] and np.array_equal(got[1].astype(np.int64), want[1]) and np.array_equal(got[2], want[2])
        st = batch.eng.get_stats()
        want = {k: sum(ot.stats()[k] for ot in trees) for k in st}
        assert st == want, ply
        batch.make_move(best.astype(np.int32))
        for ot, a in zip(trees, best):
            ot.make_move(int(a))
    batch.close()


class _Sharp:
    """Nearly all prior mass on the first empty cell: the workers of a step crowd into the same new child."""
    log_priors = False
    name = "sharp"

    def __call__(self, boards):
        import torch

        B, _, S, _ = boards.shape
        size = S * S
        occupied = boards.reshape(B, 2, size).sum(1) > 0
        first_empty = (~occupied).to(torch.int32).argmax(dim=1)
        p = torch.full((B, size), 1e-4, dtype=torch.float32, device=boards.device)
        p[torch.arange(B, device=boards.device), first_empty] = 0.9
        return torch.zeros(B, dtype=torch.float32, device=boards.device), p


def _sharp_numpy(board, player):
    size = board.shape[-1] ** 2
    p = np.full(size, 1e-4, np.float32)
    p[np.flatnonzero(board.reshape(2, size).sum(axis=0) == 0)[0]] = 0.9
    return 0.0, p


def test_workers_that_meet_in_an_unexpanded_node_leave_a_visit_deficit():
    from alphazero_amd import engine as E
    from alphazero_amd import selfplay as sp

    S, T, sims = 9, 4, 24
    batch = sp.TreeBatch(3, S, _Sharp(), cpuct=4.0, eps=0.0, threads=T, nodes_per_tree=512)
    trees = [_oracle(S, T, 0.0) for _ in range(3)]
    oev = O.callback_evaluator(_sharp_numpy, S, log_priors=False)
    for ply in range(3):
        batch.simulate(sims)
        for ot in trees:
            ot.simulate(sims, oev)
        counts = _compare(batch, trees)
        a = int(counts[0].argmax())
        child, want = batch.eng.get_node(0, [a]), trees[0].node_at([a])
        assert child[0] == want[0] and np.array_equal(child[1].astype(np.int64), want[1]) and np.array_equal(child[2], want[2])
        if ply == 0:
            assert child[0] < counts[0][a] - 1  # the deficit: fewer selects from the child than visits of the action minus one
        # two levels down as well
        b = int(child[1].argmax())
        g2, w2 = batch.eng.get_node(0, [a, b]), trees[0].node_at([a, b])
        assert (g2 is None) == (w2 is None)
        if g2 is not None:
            assert g2[0] == w2[0] and np.array_equal(g2[1].astype(np.int64), w2[1]) and np.array_equal(g2[2], w2[2])
        assert batch.eng.get_stats() == {k: sum(ot.stats()[k] for ot in trees) for k in ("sims", "expansions", "terminals", "select_calls")}
        move = counts.argmax(axis=1).astype(np.int32)
        batch.make_move(move)  # re-rooting onto a node that carries a deficit
        for ot, m in zip(trees, move):
            ot.make_move(int(m))
    # the single-worker kernels walk such a tree correctly (sequential schedule) ...
    batch.eng.set_schedule(E.WORKERS_SEQUENTIAL)
    batch.eng.set_threads(0)
    for ot in trees:
        ot.set_threads(0)
    batch.simulate(10)
    for ot in trees:
        ot.simulate(10, oev)
    _compare(batch, trees)
    # ... and the in-kernel-evaluator search, which cannot, refuses it
    with pytest.raises(E.EngineError) as ei:
        batch.eng.simulate_fused(5, E.EVAL_HASH, hash_seed=1)
    assert ei.value.code == E.ERR_STATE
    batch.eng.clear()
    batch.eng.simulate_fused(5, E.EVAL_HASH, hash_seed=1)
    batch.close()


@pytest.mark.parametrize("B,S,T,sims", [(6, 15, 4, 50), (300, 15, 4, 22), (6, 20, 8, 5), (40, 9, 3, 31)])
def test_simulate_net_runs_the_workers_in_lock_step(B, S, T, sims):
    """az_simulate_net with threads > 1 queues exactly the launches of az_select_workers / az_net_eval /
    az_expand_backup_workers (per-worker count rounded up as cmcts.cpp:98-107; n < threads: n workers of one
    search): every statistic bit-identical, and ceil(n / T) evaluator steps instead of n."""
    import torch
    from alphazero_amd import engine as E
    from test_gpu_configs import _fused

    size, plies = S * S, 2
    _, ev = _fused(S, seed=3)
    engs = []
    for _ in range(2):
        e = E.Engine(B, S, nodes_per_tree=(sims + 2 * T) * plies + 64, cpuct=4.0)
        e.set_alpha_default()
        e.set_eps(0.25)
        e.set_threads(T)
        assert e.get_schedule() == E.WORKERS_CONCURRENT
        e.set_noise_table(np.random.default_rng(4).dirichlet(np.full(size, e.get_alpha()), size=sims + 2))
        e.make_move(np.array([O.opening(size, 7 * g + 1)[0] for g in range(B)], np.int32))
        engs.append(e)
    a, b = engs
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    nw = min(T, sims)
    per = 1 if sims < T else (sims // T if sims % T == 0 else sims // T + 1)
    boards = torch.zeros((B * nw, 2, S, S), dtype=torch.float32, device=dev)
    kinds = torch.zeros(B * nw, dtype=torch.int32, device=dev)
    for ply in range(plies):
        a.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        b.begin_simulate(stream)
        for _ in range(per):
            b.select_workers(nw, boards.data_ptr(), kinds.data_ptr(), stream)
            value, logp = ev(boards, kinds)
            b.expand_backup_workers(nw, value.data_ptr(), logp.data_ptr(), True, stream)
        b.sync_check(stream)
        ca, ra = a.get_counts()
        cb, rb = b.get_counts()
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), ply
        assert ra.max() > 0 or sims < T  # n < threads: n workers of one search, which all meet the unexpanded root
        for t in range(0, B, max(1, B // 6)):
            na, nb = a.get_node(t, []), b.get_node(t, [])
            assert np.array_equal(na[2], nb[2]) and np.array_equal(na[3], nb[3]), (ply, t)
        assert a.get_stats() == b.get_stats()
        assert a.get_stats()["sims"] == B * nw * per * (ply + 1)
        act = ca.argmax(axis=1).astype(np.int32)
        if not ca.any():
            act = (a.get_raw_boards().reshape(B, 2, size).sum(axis=1) == 0).argmax(axis=1).astype(np.int32)
        a.make_move(act)
        b.make_move(act)
    a.close()
    b.close()


def test_simulate_net_lock_step_matches_the_oracle():
    """The engine-driven network search with four concurrent workers against the oracle in lock-step mode,
    whose predictor returns what the fused kernel computes for the position it is asked about."""
    import torch
    from alphazero_amd import engine as E
    from test_gpu_configs import _fused

    S, B, T, sims, plies = 15, 3, 4, 60, 3
    size = S * S
    _, ev = _fused(S, seed=5)
    dev = torch.device("cuda", 0)

    def predictor(raw, player):
        boards = torch.from_numpy(raw.astype(np.float32)).reshape(1, 2, S, S).to(dev)
        value, logp = ev(boards)
        return float(value.cpu()[0]), logp.cpu().numpy()[0]

    oev = O.callback_evaluator(predictor, S, log_priors=True)
    eng = E.Engine(B, S, nodes_per_tree=(sims + 8) * plies + 64, cpuct=4.0)
    eng.set_alpha_default()
    eng.set_eps(0.25)
    eng.set_threads(T)
    noise = np.random.default_rng(6).dirichlet(np.full(size, eng.get_alpha()), size=sims + 2)
    eng.set_noise_table(noise)
    trees = [_oracle(S, T, 0.25) for _ in range(B)]
    act = np.array([O.opening(size, 11 * g + 3)[0] for g in range(B)], np.int32)
    eng.make_move(act)
    for t, a in zip(trees, act):
        t.make_move(int(a))
    stream = torch.cuda.current_stream(dev).cuda_stream
    for ply in range(plies):
        eng.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        counts, root_n = eng.get_counts()
        for t, ot in enumerate(trees):
            ot.simulate(sims, oev, noise)
            assert root_n[t] == ot.root_n()
            assert np.array_equal(counts[t].astype(np.int64), ot.root_counts()), (ply, t)
            np.testing.assert_allclose(eng.get_node(t, [])[3], ot.root_p(), rtol=1e-6, atol=0)  # device expf vs host expf
        act = counts.argmax(axis=1).astype(np.int32)
        eng.make_move(act)
        for t, a in zip(trees, act):
            t.make_move(int(a))
    eng.close()


@pytest.mark.timeout(400)
@pytest.mark.parametrize("S,T,sims", [(15, 4, 48), (20, 8, 24)])
def test_concurrent_workers_at_full_width_against_the_oracle_on_sampled_trees(S, T, sims):
    """8192 trees with several concurrent workers each (the width the self-play drivers run at, where the
    trees are split into groups on two streams): sampled trees bit-exact against the oracle in lock-step
    mode through three moves, and the bookkeeping invariants on every tree."""
    from alphazero_amd import selfplay as sp

    G, plies, size = 8192, 3, S * S
    batch = sp.TreeBatch(G, S, ExactTorchEvaluator(), cpuct=4.0, eps=0.25, threads=T, nodes_per_tree=(sims + 2 * T) * plies + 64)
    noise = np.random.default_rng(S * T).dirichlet(np.full(size, batch.eng.get_alpha()), size=sims + 2)
    batch.set_noise_table(noise)
    sample = [0, 1, 31, 32, 2047, 4095, 4096, 4097, 6143, 8160, 8191]
    trees = {g: _oracle(S, T, 0.25) for g in sample}
    first = np.array([O.opening(size, g)[0] for g in range(G)], np.int32)
    batch.make_move(first)
    for g, ot in trees.items():
        ot.make_move(int(first[g]))
    oev = oracle_exact_evaluator(S)
    carried = np.zeros(G, np.int64)
    for ply in range(plies):
        batch.simulate(sims)
        counts, root_n = batch.eng.get_counts()
        assert np.array_equal(counts.sum(axis=1), root_n)
        raw = batch.eng.get_raw_boards().reshape(G, 2, size)
        assert not np.any(counts[(raw[:, 0] + raw[:, 1]) > 0])
        for g, ot in trees.items():
            ot.simulate(sims, oev, noise)
            assert root_n[g] == ot.root_n(), (ply, g)
            assert np.array_equal(counts[g].astype(np.int64), ot.root_counts()), (ply, g)
            got, want = batch.eng.get_node(g, []), ot.node_at([])
            assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), (ply, g)
            a = int(counts[g].argmax())
            got, want = batch.eng.get_node(g, [a]), ot.node_at([a])
            assert got[0] == want[0] and np.array_equal(got[1].astype(np.int64), want[1]) and np.array_equal(got[2], want[2]), (ply, g)
        act = counts.argmax(axis=1).astype(np.int32)
        batch.make_move(act)
        for g, ot in trees.items():
            ot.make_move(int(act[g]))
    st = batch.eng.get_stats()
    # workers that meet in a node another worker of the step expands back its value up without expanding
    assert st["sims"] == G * plies * sims and st["sims"] * 0.9 < st["expansions"] + st["terminals"] <= st["sims"]
    assert np.all(batch.eng.get_errors() == 0)
    batch.close()


def test_more_workers_than_the_engine_takes_are_refused():
    from alphazero_amd import engine as E

    e = E.Engine(2, 9, nodes_per_tree=64, cpuct=4.0)
    e.set_threads(E.MAX_WORKERS)
    with pytest.raises(E.EngineError) as ei:
        e.set_threads(E.MAX_WORKERS + 1)
    assert ei.value.code == E.ERR_ARG and "at most 64" in str(ei.value)
    assert e.get_threads() == E.MAX_WORKERS  # the refused call changed nothing
    e.close()


def test_pool_exhaustion_under_concurrent_workers_stops_the_tree_until_it_is_cleared():
    """A worker that finds the node pool empty raises the tree's error: the workers after it in the step do
    nothing, later steps skip the tree, the other trees of the batch go on, and a cleared tree searches again
    and matches the oracle."""
    from alphazero_amd import engine as E
    from alphazero_amd import selfplay as sp

    S, T, B = 9, 4, 3
    batch = sp.TreeBatch(B, S, ExactTorchEvaluator(), cpuct=4.0, eps=0.0, threads=T, nodes_per_tree=24)
    eng = batch.eng
    with pytest.raises(RuntimeError, match=r"node pool exhausted \(tree 0; 3 trees in error\)"):  # (TreeBatch raises as the facade does)
        batch.simulate(80)
    assert list(eng.get_errors()) == [E.ERR_POOL_FULL] * B
    counts0, root_n0 = eng.get_counts()
    assert 0 < root_n0.min() and root_n0.max() < 80  # stopped half way
    with pytest.raises(RuntimeError, match="node pool exhausted"):  # still pending: the trees are skipped
        batch.simulate(8)
    counts1, root_n1 = eng.get_counts()
    assert np.array_equal(counts0, counts1) and np.array_equal(root_n0, root_n1)
    eng.clear(1)
    assert list(eng.get_errors()) == [E.ERR_POOL_FULL, 0, E.ERR_POOL_FULL]
    with pytest.raises(RuntimeError, match=r"2 trees in error"):
        batch.simulate(8)
    ot = _oracle(S, T, 0.0)
    ot.simulate(8, oracle_exact_evaluator(S))
    counts, root_n = eng.get_counts()
    assert root_n[1] == ot.root_n() and np.array_equal(counts[1].astype(np.int64), ot.root_counts())
    assert root_n[0] == root_n0[0] and root_n[2] == root_n0[2]
    eng.clear()
    assert not eng.get_errors().any()
    batch.simulate(8)
    assert list(eng.get_counts()[1]) == [ot.root_n()] * B
    batch.close()


def _net_engines(B, S, T, sims, plies, seed):
    from alphazero_amd import engine as E

    size = S * S
    engs = []
    for _ in range(2):
        e = E.Engine(B, S, nodes_per_tree=(sims + 2 * T) * plies + 64, cpuct=4.0)
        e.set_alpha_default()
        e.set_eps(0.25)
        e.set_threads(T)
        e.set_noise_table(np.random.default_rng(seed).dirichlet(np.full(size, e.get_alpha()), size=sims + 2))
        engs.append(e)
    return engs


@pytest.mark.parametrize("B,S,T,sims", [(1, 15, 4, 200), (1, 20, 8, 120), (28, 15, 8, 40), (51, 9, 4, 33), (3, 13, 2, 75),
                                        (40, 6, 5, 50), (85, 5, 2, 30), (7, 19, 3, 2)])
def test_one_launch_cluster_search_equals_the_stepped_launches(B, S, T, sims, monkeypatch):
    """az_simulate_net with concurrent workers on a batch small enough for a cluster of T + 1 workgroups per tree
    (k_net_search_workers: the master's descents and backups overlap with the evaluator teams) against the
    stepped launches (AZ_NET_SEARCH=0): every statistic of every tree bit-identical, through re-rooting, with
    idle trees, and on small boards to the end of the games (terminal leaves, which no team evaluates)."""
    import torch
    from alphazero_amd import engine as E
    from test_gpu_configs import _fused

    size, plies = S * S, (S * S - 6 if S <= 6 else 3)
    _, ev = _fused(S, seed=11)
    a, b = _net_engines(B, S, T, sims, plies, seed=B + S)
    stream = torch.cuda.current_stream().cuda_stream
    first = np.array([O.opening(size, 3 * g + 2)[0] for g in range(B)], np.int32)
    active = (np.arange(B) % 5 != 3).astype(np.uint8) if B > 4 else None
    for e in (a, b):
        e.make_move(first)
        e.set_active(active)
    for ply in range(plies):
        monkeypatch.delenv("AZ_NET_SEARCH", raising=False)
        a.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        assert a.get_net_path() == E.NET_PATH_CLUSTER, (a.get_net_path(), B, S, T, sims)
        monkeypatch.setenv("AZ_NET_SEARCH", "0")
        b.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        assert b.get_net_path() == E.NET_PATH_STEPPED
        ca, ra = a.get_counts()
        cb, rb = b.get_counts()
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), ply
        assert a.get_stats() == b.get_stats()
        _, _, winner = a.get_states()
        live = [t for t in range(B) if winner[t] == -1 and (active is None or active[t])]
        for t in live[:: max(1, len(live) // 5)]:
            na, nb = a.get_node(t, []), b.get_node(t, [])
            assert na[0] == nb[0] and np.array_equal(na[2], nb[2]) and np.array_equal(na[3], nb[3]), (ply, t)
            best = int(ca[t].argmax())
            na, nb = a.get_node(t, [best]), b.get_node(t, [best])
            assert (na is None) == (nb is None)
            if na is not None:
                assert na[0] == nb[0] and np.array_equal(na[1], nb[1]) and np.array_equal(na[2], nb[2]), (ply, t)
        act = ca.argmax(axis=1).astype(np.int32)
        raw = a.get_raw_boards().reshape(B, 2, size).sum(axis=1)
        empty = (raw == 0).argmax(axis=1).astype(np.int32)
        act = np.where(ca.sum(axis=1) > 0, act, empty)
        act[winner != -1] = -1  # finished games stay as they are
        if active is not None:
            act[active == 0] = -1
        a.make_move(act)
        b.make_move(act)
    if S == 5 and sims >= 30:  # (the board fills up: the last searches end on finished games)
        assert a.get_stats()["terminals"] > 0  # searches that ended on a finished game: leaves no team evaluates
    a.close()
    b.close()


def test_one_launch_cluster_search_reports_pool_exhaustion_like_the_stepped_launches(monkeypatch):
    import torch
    from alphazero_amd import engine as E
    from test_gpu_configs import _fused

    S, T, B, sims = 9, 4, 3, 80
    _, ev = _fused(S, seed=2)
    stream = torch.cuda.current_stream().cuda_stream
    got = []
    for path in ("cluster", "stepped"):
        if path == "stepped":
            monkeypatch.setenv("AZ_NET_SEARCH", "0")
        e = E.Engine(B, S, nodes_per_tree=24, cpuct=4.0)
        e.set_eps(0.0)
        e.set_threads(T)
        with pytest.raises(E.EngineError, match=r"node pool exhausted \(tree 0; 3 trees in error\)") as ei:
            e.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        assert ei.value.code == E.ERR_POOL_FULL
        assert e.get_net_path() == (E.NET_PATH_CLUSTER if path == "cluster" else E.NET_PATH_STEPPED)
        got.append((e.get_counts(), e.get_stats(), list(e.get_errors())))
        e.clear()
        e.simulate_net(8, ev.params.data_ptr(), ev.fused_layers, stream)
        got[-1] += (e.get_counts(),)
        e.close()
    (c0, s0, e0, d0), (c1, s1, e1, d1) = got
    assert np.array_equal(c0[0], c1[0]) and np.array_equal(c0[1], c1[1]) and s0 == s1 and e0 == e1
    assert np.array_equal(d0[0], d1[0]) and np.array_equal(d0[1], d1[1])
