"""GPU parity of the stepped path (az_select -> external evaluator -> az_expand_backup),
the path a batched policy/value network drives, against the CPU oracle.
"""
import numpy as np
import pytest

import az_oracle_py as O

pytestmark = pytest.mark.gpu


def _step_sims(eng, torch, n, evaluator, log_priors=False):
    """n searches on every active tree with a host evaluator(raw int8 board) -> (v, p)."""
    B, S, size = eng.n_trees, eng.shape, eng.size
    dev = torch.device("cuda", eng.device)
    boards = torch.zeros((B, 2, S, S), dtype=torch.float32, device=dev)
    kinds = torch.zeros(B, dtype=torch.int32, device=dev)
    values = torch.zeros(B, dtype=torch.float32, device=dev)
    priors = torch.zeros((B, size), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    eng.begin_simulate(stream)
    n_expand = 0
    for _ in range(n):
        eng.select(boards.data_ptr(), kinds.data_ptr(), stream)
        k = kinds.cpu().numpy()
        b = boards.cpu().numpy()
        v = np.zeros(B, np.float32)
        p = np.zeros((B, size), np.float32)
        for t in range(B):
            if k[t] == 1:
                v[t], p[t] = evaluator(b[t].astype(np.int8))
                n_expand += 1
        values.copy_(torch.from_numpy(v))
        priors.copy_(torch.from_numpy(p))
        eng.expand_backup(values.data_ptr(), priors.data_ptr(), log_priors, stream)
    eng.sync_check(stream)
    return n_expand


@pytest.mark.parametrize("S,eps", [(15, 0.0), (15, 0.25), (9, 0.25), (20, 0.0)])
def test_step_matches_oracle(S, eps):
    import torch
    from alphazero_amd import engine as E

    size = S * S
    B, sims, plies = 4, 120, 5
    eng = E.Engine(B, S, nodes_per_tree=sims * 4, cpuct=4.0)
    eng.set_alpha_default()
    eng.set_eps(eps)
    noise = None
    if eps > 0:
        noise = np.random.default_rng(5).dirichlet(np.full(size, eng.get_alpha()), size=sims + 4)
        eng.set_noise_table(noise)
    ev = O.hash_evaluator(77)
    trees = []
    for g in range(B):
        t = O.OracleTree(S, cpuct=4.0)
        t.set_alpha_default()
        t.set_eps(eps)
        trees.append(t)
    act = np.array([O.opening(size, g)[0] for g in range(B)], np.int32)
    eng.make_move(act)
    for t, a in zip(trees, act):
        t.make_move(int(a))
    for ply in range(plies):
        _step_sims(eng, torch, sims, lambda b: O.hash_eval(S, 77, b))
        for ot in trees:
            ot.simulate(sims, ev, noise)
        counts, root_n = eng.get_counts()
        for t, ot in enumerate(trees):
            assert root_n[t] == ot.root_n(), (ply, t)
            assert np.array_equal(counts[t].astype(np.int64), ot.root_counts()), (ply, t)
            node = eng.get_node(t, [])
            assert np.array_equal(node[2], ot.root_w()) and np.array_equal(node[3], ot.root_p())
        act = counts.argmax(axis=1).astype(np.int32)
        eng.make_move(act)
        for ot, a in zip(trees, act):
            ot.make_move(int(a))
    # the fused path continues seamlessly on trees built by the stepped path
    eng.set_noise_table(noise)
    eng.simulate_fused(50, eval_kind=E.EVAL_HASH, hash_seed=77)
    for ot in trees:
        ot.simulate(50, ev, noise)
    counts, root_n = eng.get_counts()
    for t, ot in enumerate(trees):
        assert np.array_equal(counts[t].astype(np.int64), ot.root_counts())
    st = eng.get_stats()
    tot = dict(sims=0, expansions=0, terminals=0, select_calls=0)
    for ot in trees:
        for k, v in ot.stats().items():
            tot[k] += v
    assert st == tot
    eng.close()


def test_step_terminals_small_board():
    import torch
    from alphazero_amd import engine as E

    S, size = 6, 36
    B, sims = 6, 60
    eng = E.Engine(B, S, nodes_per_tree=sims * 3, cpuct=2.0)
    eng.set_eps(0.0)
    ev = O.hash_evaluator(5)
    trees = [O.OracleTree(S, cpuct=2.0) for _ in range(B)]
    for t in trees:
        t.set_eps(0.0)
    finished = np.zeros(B, bool)
    for ply in range(size):
        eng.set_active((~finished).astype(np.uint8))
        _step_sims(eng, torch, sims, lambda b: O.hash_eval(S, 5, b))
        for t, ot in enumerate(trees):
            if not finished[t]:
                ot.simulate(sims, ev)
        counts, root_n = eng.get_counts()
        act = np.full(B, -1, np.int32)
        for t, ot in enumerate(trees):
            assert np.array_equal(counts[t].astype(np.int64), ot.root_counts()), (ply, t)
            if not finished[t]:
                act[t] = int(counts[t].argmax())
        eng.make_move(act)
        for t, ot in enumerate(trees):
            if act[t] >= 0:
                ot.make_move(int(act[t]))
                finished[t] = ot.winner != -1
        pl, mc, wi = eng.get_states()
        assert [float(x) for x in wi] == [ot.winner for ot in trees]
        if finished.all():
            break
    assert finished.all()
    assert eng.get_stats()["terminals"] == sum(ot.stats()["terminals"] for ot in trees) > 0
    eng.close()


def test_step_log_priors_tolerance():
    """Log-probability priors go through the device expf; the reference uses the
    host libm's float exp (node.cpp:42), which is not bit-pinned across machines,
    so P is compared at 1e-6 relative (a few float ulps), not bitwise."""
    import torch
    from alphazero_amd import engine as E

    S, size = 15, 225
    rng = np.random.default_rng(11)
    logits = rng.normal(size=size).astype(np.float32)
    logp = (logits - np.log(np.exp(logits.astype(np.float64)).sum())).astype(np.float32)
    eng = E.Engine(1, S, nodes_per_tree=16, cpuct=4.0)
    eng.set_eps(0.0)
    _step_sims(eng, torch, 1, lambda b: (0.25, logp), log_priors=True)
    node = eng.get_node(0, [])
    ot = O.OracleTree(S, cpuct=4.0)
    ot.set_eps(0.0)
    ot.simulate(1, O.callback_evaluator(lambda b, pl: (0.25, logp), S, log_priors=True))
    np.testing.assert_allclose(node[3], ot.root_p(), rtol=1e-6, atol=0)
    eng.close()


@pytest.mark.timeout(400)
def test_step_8192_trees_sampled_against_oracle():
    """The network-driven configuration at its full width: 8192 concurrent trees on the
    stepped path, one batched evaluator call per search step, noise at every
    expansion; a spread of 48 trees is replayed by the oracle, invariants on all."""
    from alphazero_amd import selfplay as sp
    from helpers import ExactTorchEvaluator, oracle_exact_evaluator

    G, S, sims, plies = 8192, 15, 100, 3
    size = S * S
    batch = sp.TreeBatch(G, S, ExactTorchEvaluator(), cpuct=4.0, eps=0.25, nodes_per_tree=sims * 3 + 64)
    noise = np.random.default_rng(9).dirichlet(np.full(size, batch.eng.get_alpha()), size=sims + 4)
    batch.set_noise_table(noise)
    sample = sorted(set(int(x) for x in np.linspace(0, G - 1, 48)))
    oev = oracle_exact_evaluator(S)
    trees = {}
    for g in sample:
        t = O.OracleTree(S, cpuct=4.0)
        t.set_alpha_default()
        t.set_eps(0.25)
        trees[g] = t
    for ply in range(2):
        act = np.array([O.opening(size, g)[ply] for g in range(G)], np.int32)
        batch.make_move(act)
        for g, t in trees.items():
            t.make_move(int(act[g]))
    carried = np.full(G, -1, np.int64)  # nodeN the root brought along, -1 = not expanded yet
    for ply in range(plies):
        batch.simulate(sims)
        counts, root_n = batch.eng.get_counts()
        # every tree: all searches but the one expanding a fresh root visit a child
        assert np.array_equal(counts.sum(axis=1), root_n)
        fresh = carried < 0
        assert np.array_equal(root_n[fresh], np.full(int(fresh.sum()), sims - 1))
        assert np.array_equal(root_n[~fresh], carried[~fresh] + sims)
        for g, t in trees.items():
            t.simulate(sims, oev, noise)
            assert root_n[g] == t.root_n(), (ply, g)
            assert np.array_equal(counts[g].astype(np.int64), t.root_counts()), (ply, g)
        act = counts.argmax(axis=1).astype(np.int32)
        # the chosen child's nodeN is its visit count minus the visit that created it
        carried = counts[np.arange(G), act].astype(np.int64) - 1
        batch.make_move(act)
        for g, t in trees.items():
            t.make_move(int(act[g]))
        _, _, winner = batch.eng.get_states()
        for g, t in trees.items():
            assert float(winner[g]) == t.winner
    batch.close()


def test_step_paths_deeper_than_64_levels():
    """cpuct = 0 and a constant value: all scores tie, every search extends the line of
    first legal actions by one node (level 80 on 20x20), through the stepped kernels."""
    import torch
    from alphazero_amd import engine as E

    S, size, sims = 20, 400, 450
    p = np.full(size, np.float32(1.0 / size), np.float32)
    eng = E.Engine(2, S, nodes_per_tree=sims + 8, cpuct=0.0)
    eng.set_eps(0.0)
    _step_sims(eng, torch, sims, lambda b: (np.float32(0.0), p))
    ot = O.OracleTree(S, cpuct=0.0)
    ot.set_eps(0.0)
    ot.simulate(sims, O.uniform_evaluator(np.float32(1.0 / size)))
    counts, root_n = eng.get_counts()
    assert root_n[0] == ot.root_n() and np.array_equal(counts[0].astype(np.int64), ot.root_counts())
    assert np.array_equal(counts[1], counts[0])
    st, want = eng.get_stats(), ot.stats()
    assert st["select_calls"] == 2 * want["select_calls"] and st["terminals"] == 2 * want["terminals"]
    assert want["select_calls"] / want["sims"] > 30
    eng.close()


@pytest.mark.timeout(400)
def test_step_random_configurations():
    """Differential sweep of the stepped path (az_select / az_expand_backup) against the
    oracle: board edge, search length, noise share and cpuct at random."""
    import torch
    from alphazero_amd import engine as E

    rng = np.random.default_rng(77001)
    for _ in range(16):
        S = int(rng.integers(5, 21))
        size = S * S
        sims, plies, B = int(rng.integers(10, 90)), int(rng.integers(1, 4)), 2
        eps = float(rng.choice([0.0, 0.25, 1.0]))
        cpuct = float(rng.choice([0.0, 1.0, 4.0]))
        seed = int(rng.integers(1, 1000))
        eng = E.Engine(B, S, nodes_per_tree=sims * (plies + 1) + 16, cpuct=cpuct)
        eng.set_alpha_default()
        eng.set_eps(eps)
        noise = None
        if eps > 0:
            noise = np.random.default_rng(seed).dirichlet(np.full(size, eng.get_alpha()), size=sims + 2)
            eng.set_noise_table(noise)
        ev = O.hash_evaluator(seed)
        trees = []
        for g in range(B):
            t = O.OracleTree(S, cpuct=cpuct)
            t.set_alpha_default()
            t.set_eps(eps)
            trees.append(t)
        act = np.array([O.opening(size, g + seed)[0] for g in range(B)], np.int32)
        eng.make_move(act)
        for t, a in zip(trees, act):
            t.make_move(int(a))
        for ply in range(plies):
            _step_sims(eng, torch, sims, lambda b: O.hash_eval(S, seed, b))
            counts, root_n = eng.get_counts()
            for t, ot in enumerate(trees):
                ot.simulate(sims, ev, noise)
                tag = (S, sims, eps, cpuct, seed, ply, t)
                assert root_n[t] == ot.root_n(), tag
                assert np.array_equal(counts[t].astype(np.int64), ot.root_counts()), tag
                node = eng.get_node(t, [])
                assert np.array_equal(node[2], ot.root_w()) and np.array_equal(node[3], ot.root_p()), tag
            act = counts.argmax(axis=1).astype(np.int32)
            eng.make_move(act)
            for ot, a in zip(trees, act):
                ot.make_move(int(a))
            if any(ot.winner != -1 for ot in trees):
                break
        eng.close()
