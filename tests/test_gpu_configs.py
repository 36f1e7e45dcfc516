"""The benchmark configurations at their stated workloads, and the engine-driven
network search (az_simulate_net), on the GPU:
  * one full 15x15 self-play game through the drop-in facade with a TorchScript file
    of the reference-configuration network (SIMS=100, cpuct 4, eps 0.25, TAU 10);
  * 8192 concurrent trees driven by the fused network kernel;
  * per-tree error handling when several trees fail in one launch.
(The arena with noise and two networks is in test_gpu_drivers.py.)
"""
import numpy as np
import pytest

import az_oracle_py as O

pytestmark = pytest.mark.gpu


def _fused(S, seed=0, **kw):
    import torch

    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet

    torch.manual_seed(seed)
    net = PolicyValueNet(S, **kw).eval()
    # random-init batch-norm statistics would be the identity: give them some spread
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    return net, FusedNetEvaluator(net, S, 0)


def _stepped(eng, torch, ev, n):
    """n searches through the three separate entry points (the path az_simulate_net replaces)."""
    dev = torch.device("cuda", eng.device)
    boards = torch.zeros((eng.n_trees, 2, eng.shape, eng.shape), dtype=torch.float32, device=dev)
    kinds = torch.zeros(eng.n_trees, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    eng.begin_simulate(stream)
    for _ in range(n):
        eng.select(boards.data_ptr(), kinds.data_ptr(), stream)
        value, logp = ev(boards, kinds)
        eng.expand_backup(value.data_ptr(), logp.data_ptr(), True, stream)
    eng.sync_check(stream)


@pytest.mark.parametrize("S,threads", [(15, 0), (9, 3), (20, 0)])
def test_simulate_net_equals_the_three_step_path(S, threads):
    """az_simulate_net queues exactly the launches of az_select / az_net_eval /
    az_expand_backup: every statistic of every tree is bit-identical."""
    import torch
    from alphazero_amd import engine as E

    B, sims, plies, size = 6, 50, 3, S * S
    _, ev = _fused(S, seed=3)
    engs = []
    for _ in range(2):
        e = E.Engine(B, S, nodes_per_tree=sims * plies + 64, cpuct=4.0)
        e.set_alpha_default()
        e.set_eps(0.25)
        e.set_schedule(E.WORKERS_SEQUENTIAL)  # this test steps the workers one after another by hand
        e.set_threads(threads)
        e.set_noise_table(np.random.default_rng(4).dirichlet(np.full(size, e.get_alpha()), size=sims + 2))
        e.make_move(np.array([O.opening(size, 7 * g + 1)[0] for g in range(B)], np.int32))
        engs.append(e)
    a, b = engs
    stream = torch.cuda.current_stream().cuda_stream
    for ply in range(plies):
        a.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        if threads > 1:  # the workers of cmcts.cpp:98-107, one after another
            per = sims // threads + (1 if sims % threads else 0)
            for _ in range(threads):
                _stepped(b, torch, ev, per)
        else:
            _stepped(b, torch, ev, sims)
        ca, ra = a.get_counts()
        cb, rb = b.get_counts()
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), ply
        for t in range(B):
            na, nb = a.get_node(t, []), b.get_node(t, [])
            assert np.array_equal(na[2], nb[2]) and np.array_equal(na[3], nb[3]), (ply, t)
        assert a.get_stats() == b.get_stats()
        act = ca.argmax(axis=1).astype(np.int32)
        a.make_move(act)
        b.make_move(act)
    a.close()
    b.close()


@pytest.mark.parametrize("one_launch", [False, True])
@pytest.mark.parametrize("B,S,plies", [(77, 5, 24), (130, 7, 3), (33, 16, 3), (41, 15, 2), (21, 20, 2), (29, 13, 2), (18, 19, 2)])
def test_simulate_net_position_mask_with_idle_and_terminal_leaves(B, S, plies, one_launch, net_form):
    """az_simulate_net tells its evaluator which positions to evaluate through the bit mask az_select
    fills (the three-step path scans the leaf codes): tree counts that are no multiple of 32, idle trees
    and, on small boards, searches that end on finished games; every statistic stays bit-identical.
    one_launch: the kernel form is left to the library, which then runs a batch this small as ONE launch
    per call (a workgroup per tree selects, evaluates and backs up): the same comparison for that path."""
    import torch
    from alphazero_amd import engine as E

    sims, size = 40, S * S  # (5x5: the board fills up, searches end on finished games and the games themselves end)
    _, ev = _fused(S, seed=5)
    # the one-wavefront kernels are the ones that read the mask; batches this small would get the team kernel
    if not one_launch:
        net_form({13: "w43", 14: "w43", 15: "w43", 16: "wide", 17: "w43", 18: "wide", 19: "w43", 20: "w43"}.get(S, "winograd"))
    active = (np.arange(B) % 5 != 2).astype(np.uint8)
    engs = []
    for _ in range(2):
        e = E.Engine(B, S, nodes_per_tree=sims * plies + 64, cpuct=4.0)
        e.set_alpha_default()
        e.set_eps(0.25)
        e.set_noise_table(np.random.default_rng(9).dirichlet(np.full(size, e.get_alpha()), size=sims + 2))
        e.make_move(np.array([O.opening(size, 3 * g + 2)[0] for g in range(B)], np.int32))
        e.set_active(active)
        engs.append(e)
    a, b = engs
    stream = torch.cuda.current_stream().cuda_stream
    terminals = 0
    for ply in range(plies):
        a.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        a.sync_check(stream)
        _stepped(b, torch, ev, sims)
        ca, ra = a.get_counts()
        cb, rb = b.get_counts()
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), ply
        assert a.get_stats() == b.get_stats()
        assert not ca[active == 0].any()  # idle trees were not searched
        for t in range(0, B, 7):
            na, nb = a.get_node(t, []), b.get_node(t, [])
            if na is None or nb is None:  # a tree that was never searched has no root node
                assert na is None and nb is None and active[t] == 0
                continue
            assert np.array_equal(na[2], nb[2]) and np.array_equal(na[3], nb[3]), (ply, t)
        terminals = a.get_stats()["terminals"]
        act = np.where(active == 1, ca.argmax(axis=1), -1).astype(np.int32)  # a negative action leaves the tree alone
        for eng in (a, b):
            eng.make_move(act)
        # games that are over leave the search
        active = (active.astype(bool) & (a.get_states()[2] == -1.0)).astype(np.uint8)
        for eng in (a, b):
            eng.set_active(active)
    if S == 5:
        assert terminals > 0 and not active.any()  # searches reached finished games; every game is over
    a.close()
    b.close()


@pytest.mark.parametrize("B,S,threads,groups", [(200, 15, 0, "3"), (97, 9, 0, "2"), (200, 15, 4, "4"), (70, 20, 3, "2"), (1030, 13, 0, "4")])
def test_simulate_net_tree_groups_do_not_change_results(B, S, threads, groups, monkeypatch, net_form):
    """az_simulate_net may split the trees into groups that run on streams of their own (so that one group's tree
    kernels overlap the other's evaluation); trees are independent, so every statistic is the one-group result.
    Uneven splits, fewer 32-tree blocks than groups, sequential and concurrent workers."""
    import torch
    from alphazero_amd import engine as E

    sims, plies, size = 24, 2, S * S
    _, ev = _fused(S, seed=7)
    stream = torch.cuda.current_stream().cuda_stream
    # (one kernel form for every batch size: left to itself az_net_eval gives small batches a kernel of their own,
    # whose float32 sums round differently; the engine only forms groups far above those sizes)
    net_form("w43" if S >= 13 else "direct")
    results = []
    for g in ("1", groups):
        monkeypatch.setenv("AZ_NET_GROUPS", g)
        monkeypatch.setenv("AZ_NET_SEARCH", "0")
        e = E.Engine(B, S, nodes_per_tree=(sims + 8) * plies + 64, cpuct=4.0)
        e.set_alpha_default()
        e.set_eps(0.25)
        e.set_threads(threads)
        e.set_noise_table(np.random.default_rng(4).dirichlet(np.full(size, e.get_alpha()), size=sims + 2))
        e.make_move(np.array([O.opening(size, 7 * t + 1)[0] for t in range(B)], np.int32))
        e.set_active((np.arange(B) % 7 != 3).astype(np.uint8))
        out = []
        for ply in range(plies):
            e.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
            counts, root_n = e.get_counts()
            out.append((counts.copy(), root_n.copy(), e.get_stats(), [e.get_node(t, [])[2].copy() for t in (0, B // 2 + (B // 2 % 7 == 3), B - 1 - ((B - 1) % 7 == 3))]))
            e.make_move(np.where(root_n > 0, counts.argmax(axis=1), -1).astype(np.int32))
        results.append(out)
        e.close()
    for a, b in zip(*results):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
        assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
        assert a[1].max() > 0


def test_simulate_net_matches_oracle():
    """The engine-driven network search against the CPU oracle, whose predictor returns
    what the fused kernel computes for the position it is asked about (batch of one):
    visit counts and node totals must agree exactly, priors to the accuracy of the
    two expf implementations (device vs host libm, node.cpp:42)."""
    import torch
    from alphazero_amd import engine as E

    S, B, sims, plies = 15, 4, 80, 3
    size = S * S
    _, ev = _fused(S, seed=5)
    dev = torch.device("cuda", 0)

    def predictor(raw, player):
        boards = torch.from_numpy(raw.astype(np.float32)).reshape(1, 2, S, S).to(dev)
        value, logp = ev(boards)
        return float(value.cpu()[0]), logp.cpu().numpy()[0]

    oev = O.callback_evaluator(predictor, S, log_priors=True)
    eng = E.Engine(B, S, nodes_per_tree=sims * plies + 64, cpuct=4.0)
    eng.set_alpha_default()
    eng.set_eps(0.25)
    noise = np.random.default_rng(6).dirichlet(np.full(size, eng.get_alpha()), size=sims + 2)
    eng.set_noise_table(noise)
    trees = []
    for g in range(B):
        t = O.OracleTree(S, cpuct=4.0)
        t.set_alpha_default()
        t.set_eps(0.25)
        trees.append(t)
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
            assert root_n[t] == ot.root_n(), (ply, t)
            assert np.array_equal(counts[t].astype(np.int64), ot.root_counts()), (ply, t)
            node = eng.get_node(t, [])
            np.testing.assert_allclose(node[3], ot.root_p(), rtol=1e-6, atol=0)
            np.testing.assert_allclose(node[2], ot.root_w(), rtol=1e-5, atol=1e-6)  # Q within 1e-5
        act = counts.argmax(axis=1).astype(np.int32)
        eng.make_move(act)
        for ot, a in zip(trees, act):
            ot.make_move(int(a))
    st = eng.get_stats()
    tot = {k: sum(ot.stats()[k] for ot in trees) for k in st}
    assert st == tot
    eng.close()


def test_config1_full_game_through_the_facade(tmp_path):
    """15x15, SIMS=100, the reference-configuration network (general_config.py:108-129 /
    config.py: 4 + 2 + 2 layers of 16 channels) as a TorchScript file, cpuct 4,
    eps 0.25, TAU 10, default alpha, threads 0: one whole self_play_game
    (src/self_play.py:134-202) written against the drop-in module, record rules checked."""
    import torch

    import alphazero_amd.cmcts as cmcts
    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet
    from alphazero_amd.selfplay import choice_from_uniform, sample_dtype

    S, SIMS, TAU = 15, 100, 10
    size = S * S
    torch.manual_seed(0)
    net = PolicyValueNet(S).eval()
    path = str(tmp_path / "net15.pt")
    torch.jit.trace(net, torch.rand(1, 2, S, S)).save(path)
    ms = []
    for _ in range(2):  # src/self_play.py:69-79
        m = cmcts.mcts(cpuct=4.0, shape=S, nodes_per_tree=8192)
        m.set_alpha_default()
        m.set_threads(0)
        m.set_params(path)
        m.eps = 0.25
        ms.append(m)
    mcts0, mcts1 = ms
    rs = np.random.RandomState(0)
    dt = sample_dtype(S)
    data0, data1 = [], []
    for i in range(size):  # src/self_play.py:155-186
        for searcher, mine in ((mcts0, data0), (mcts1, data1)):
            searcher.simulate(SIMS)
            pi = searcher.get_prob()
            board = searcher.get_board()
            if i > TAU:
                move = int(pi.argmax())
                pi = np.zeros(size, np.float32)
                pi[move] = 1
            else:
                move = choice_from_uniform(pi, rs.random_sample())
            mine.append(np.array((board, pi, -1), dtype=dt))
            mcts0.make_move(move)
            mcts1.make_move(move)
            if mcts0.winner != -1:
                break
        if mcts0.winner != -1:
            break
    assert isinstance(mcts0._evaluator, FusedNetEvaluator)  # the file holds the reference architecture
    data0, data1 = np.stack(data0), np.stack(data1)
    if mcts0.winner == 0:  # src/self_play.py:188-200
        data0["r"] = 1
    elif mcts0.winner == 1:
        data1["r"] = 1
    else:
        data0["r"] = data1["r"] = 1e-3
    data = np.concatenate([data0, data1])
    n_plies = mcts0.move_cnt
    assert mcts0.winner in (0.0, 1.0, 0.5) and mcts1.winner == mcts0.winner and mcts1.move_cnt == n_plies
    assert len(data) == n_plies and len(data0) - len(data1) in (0, 1)
    assert data.dtype == dt and data["board"].shape == (n_plies, 2, S, S) and data["pi"].shape == (n_plies, size)
    # record k of the first player was taken after 2k stones, of the second after 2k+1; plane 0 = side to move
    for k, rec in enumerate(data0):
        assert rec["board"].sum() == 2 * k and rec["board"][0].sum() == k
    for k, rec in enumerate(data1):
        assert rec["board"].sum() == 2 * k + 1 and rec["board"][0].sum() == k
    np.testing.assert_allclose(data["pi"].sum(axis=1), 1.0, atol=1e-3)
    for d in (data0, data1):
        for k, rec in enumerate(d):
            hot = (rec["pi"] > 0).sum()
            assert hot == 1 if k > TAU else hot >= 1
            assert np.all(rec["board"].reshape(2, -1).sum(axis=0)[rec["pi"] > 0] == 0)  # only empty cells are visited
    if mcts0.winner == 0.5:
        assert np.all(data["r"] == np.float32(1e-3))
    else:
        win, lose = (data0, data1) if mcts0.winner == 0 else (data1, data0)
        assert np.all(win["r"] == 1) and np.all(lose["r"] == -1)


@pytest.mark.timeout(400)
def test_config3_8192_trees_driven_by_the_fused_network():
    """8192 concurrent 15x15 trees, random-init reference network, noise at every
    expansion from the GPU sampler: the kernel's outputs on a sample of real leaf
    boards against PyTorch float64, and the bookkeeping invariants on all trees."""
    import torch
    from alphazero_amd import selfplay as sp

    G, S, sims, plies = 8192, 15, 24, 3
    size = S * S
    net, ev = _fused(S, seed=0)
    batch = sp.TreeBatch(G, S, ev, cpuct=4.0, eps=0.25, nodes_per_tree=256, noise_seed=1)
    eng = batch.eng
    for ply in range(2):
        batch.make_move(np.array([O.opening(size, g)[ply] for g in range(G)], np.int32))
    # one step by hand: the boards the network sees at full width
    stream = torch.cuda.current_stream(batch.dev).cuda_stream
    batch._ensure_noise(sims)
    eng.set_active(None)
    eng.begin_simulate(stream)
    eng.select(batch.boards.data_ptr(), batch.kinds.data_ptr(), stream)
    value, logp = ev(batch.boards, batch.kinds)
    assert int((batch.kinds == 1).sum()) == G  # every fresh root is a leaf to expand
    idx = torch.from_numpy(np.linspace(0, G - 1, 96).astype(np.int64)).to(batch.dev)
    with torch.no_grad():
        v64, lp64 = net.double().to(batch.dev)(batch.boards[idx].double())
    np.testing.assert_allclose(value[idx].cpu().numpy(), v64.reshape(-1).cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(logp[idx].cpu().numpy(), lp64.reshape(len(idx), -1).cpu().numpy(), rtol=0, atol=2e-5)
    eng.expand_backup(value.data_ptr(), logp.data_ptr(), True, stream)
    eng.sync_check(stream)
    carried = np.zeros(G, np.int64)  # nodeN the root brings along (the step above expanded it: 0)
    for ply in range(plies):
        batch.simulate(sims)  # az_simulate_net
        counts, root_n = eng.get_counts()
        assert np.array_equal(counts.sum(axis=1), root_n)
        assert np.array_equal(root_n, carried + sims)
        assert counts.min() >= 0
        raw = eng.get_raw_boards().reshape(G, 2, size)
        assert not np.any(counts[(raw[:, 0] + raw[:, 1]) > 0])  # occupied cells are never visited
        act = counts.argmax(axis=1).astype(np.int32)
        carried = counts[np.arange(G), act].astype(np.int64) - 1
        batch.make_move(act)
    st = eng.get_stats()
    assert st["sims"] == G * (1 + plies * sims)
    assert st["expansions"] + st["terminals"] == st["sims"]
    assert np.all(eng.get_errors() == 0)
    _, move_cnt, winner = eng.get_states()
    assert np.all(move_cnt == 2 + plies) and np.all(winner == -1)
    batch.close()


@pytest.mark.timeout(400)
def test_config4_share_8192_trees_of_20x20_driven_by_the_fused_network():
    """One GPU's share of config 4 through the network path: 8192 concurrent 20x20 trees searched by
    az_simulate_net with the two-wavefront F(4x4,3x3) kernel.  The kernel's outputs at full width on
    real leaf boards (first step: every root; a later step: leaves two and three levels down, through
    the position mask) against PyTorch float64, and the bookkeeping invariants on all trees."""
    import torch
    from alphazero_amd import selfplay as sp

    G, S, sims, plies = 8192, 20, 16, 3
    size = S * S
    net, ev = _fused(S, seed=0)
    net64 = net.double().to("cuda")
    batch = sp.TreeBatch(G, S, ev, cpuct=4.0, eps=0.25, nodes_per_tree=160, noise_seed=1)
    eng = batch.eng
    for ply in range(2):
        batch.make_move(np.array([O.opening(size, g)[ply] for g in range(G)], np.int32))
    stream = torch.cuda.current_stream(batch.dev).cuda_stream
    batch._ensure_noise(sims)
    eng.set_active(None)
    eng.begin_simulate(stream)
    idx = torch.from_numpy(np.linspace(0, G - 1, 96).astype(np.int64)).to(batch.dev)

    def check_step(expect_all):
        eng.select(batch.boards.data_ptr(), batch.kinds.data_ptr(), stream)
        value, logp = ev(batch.boards, batch.kinds)
        expand = batch.kinds == 1
        assert int(expand.sum()) == G if expect_all else int(expand.sum()) > G // 2
        pick = idx[expand[idx]]
        with torch.no_grad():
            v64, lp64 = net64(batch.boards[pick].double())
        np.testing.assert_allclose(value[pick].cpu().numpy(), v64.reshape(-1).cpu().numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(logp[pick].cpu().numpy(), lp64.reshape(len(pick), -1).cpu().numpy(), rtol=0, atol=2e-5)
        eng.expand_backup(value.data_ptr(), logp.data_ptr(), True, stream)
        eng.sync_check(stream)
        return len(pick)

    assert check_step(True) == 96  # every fresh root is a leaf to expand
    for _ in range(5):
        checked = check_step(False)  # leaves below the root
    assert checked > 48
    carried = np.full(G, 5, np.int64)  # nodeN the root brings along: the five searches above (its own expansion counts none)
    for ply in range(plies):
        batch.simulate(sims)  # az_simulate_net: k_backup_select + k_net_w43t<20> with the position mask
        counts, root_n = eng.get_counts()
        assert np.array_equal(counts.sum(axis=1), root_n)
        assert np.array_equal(root_n, carried + sims)
        assert counts.min() >= 0
        raw = eng.get_raw_boards().reshape(G, 2, size)
        assert not np.any(counts[(raw[:, 0] + raw[:, 1]) > 0])  # occupied cells are never visited
        act = counts.argmax(axis=1).astype(np.int32)
        carried = counts[np.arange(G), act].astype(np.int64) - 1
        batch.make_move(act)
    st = eng.get_stats()
    assert st["sims"] == G * (6 + plies * sims)
    assert st["expansions"] + st["terminals"] == st["sims"]
    assert np.all(eng.get_errors() == 0)
    _, move_cnt, winner = eng.get_states()
    assert np.all(move_cnt == 2 + plies) and np.all(winner == -1)
    batch.close()


def test_errors_of_several_trees_in_one_launch_stay_until_cleared():
    """Pool exhaustion stops a search half way, so the tree must not be searched again
    before it is cleared; an illegal move leaves its tree intact and is cleared by the report."""
    import torch
    from alphazero_amd import engine as E

    S, B = 9, 3
    eng = E.Engine(B, S, nodes_per_tree=8, cpuct=4.0)
    eng.set_eps(0.0)
    with pytest.raises(E.EngineError, match=r"node pool exhausted \(tree 0; 3 trees in error\)") as ei:
        eng.simulate_fused(50, E.EVAL_HASH, hash_seed=3)
    assert ei.value.code == E.ERR_POOL_FULL
    assert list(eng.get_errors()) == [E.ERR_POOL_FULL] * B
    counts0, root_n0 = eng.get_counts()
    with pytest.raises(E.EngineError, match="node pool exhausted"):  # still pending, trees skipped
        eng.simulate_fused(5, E.EVAL_HASH, hash_seed=3)
    counts1, root_n1 = eng.get_counts()
    assert np.array_equal(counts0, counts1) and np.array_equal(root_n0, root_n1)
    eng.clear(1)
    assert list(eng.get_errors()) == [E.ERR_POOL_FULL, 0, E.ERR_POOL_FULL]
    with pytest.raises(E.EngineError, match=r"node pool exhausted \(tree 0; 2 trees in error\)"):
        eng.simulate_fused(5, E.EVAL_HASH, hash_seed=3)
    _, root_n = eng.get_counts()
    assert root_n[1] == 4 and root_n[0] == root_n0[0]  # the cleared tree searches again, the others do not
    ot = O.OracleTree(S, cpuct=4.0)
    ot.set_eps(0.0)
    ot.simulate(5, O.hash_evaluator(3))
    assert np.array_equal(eng.get_counts()[0][1].astype(np.int64), ot.root_counts())
    eng.clear()
    assert not eng.get_errors().any()
    eng.simulate_fused(5, E.EVAL_HASH, hash_seed=3)
    assert list(eng.get_counts()[1]) == [4] * B
    # an illegal move on the device path: reported once, tree untouched, the next check is clean
    dev = torch.device("cuda", 0)
    best = int(eng.get_counts()[0][0].argmax())
    eng.make_move(np.array([best, -1, -1], np.int32))
    eng.make_move_device(torch.tensor([best, -1, -1], dtype=torch.int32, device=dev).data_ptr())
    with pytest.raises(E.EngineError, match="Invalid move") as ei:
        eng.sync_check()
    assert ei.value.code == E.ERR_INVALID_MOVE
    eng.sync_check()
    assert eng.get_state(0)[1] == 1
    eng.close()
