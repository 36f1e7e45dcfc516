"""az_simulate_net_noalloc (the allocation-free variant of az_simulate_net, step staging
in the library's fixed per-GPU buffer) against az_simulate_net, through the C ABI:
bit-identical statistics on two calls in a row, the second with a smaller input
(more trees than the GPU has CUs, so it too takes the stepped path and lays its
smaller staging layout over what the first call left in the buffer), no change in
the device's free memory across the calls, and a clear error beyond the buffer's size."""
import numpy as np
import pytest

import az_oracle_py as O

pytestmark = pytest.mark.gpu


def _pair(E, B, S, sims, seed):
    size = S * S
    out = []
    for _ in range(2):
        e = E.Engine(B, S, nodes_per_tree=2 * sims + 64, cpuct=4.0)
        e.set_alpha_default()
        e.set_eps(0.25)
        e.set_noise_table(np.random.default_rng(seed).dirichlet(np.full(size, e.get_alpha()), size=sims + 2))
        e.make_move(np.array([O.opening(size, 3 * g + seed)[0] for g in range(B)], np.int32))
        out.append(e)
    return out


def _same(a, b):
    ca, ra = a.get_counts()
    cb, rb = b.get_counts()
    assert np.array_equal(ra, rb) and np.array_equal(ca, cb)
    for t in sorted(set(int(x) for x in np.linspace(0, a.n_trees - 1, 12))):
        na, nb = a.get_node(t, []), b.get_node(t, [])
        assert np.array_equal(na[2], nb[2]) and np.array_equal(na[3], nb[3]), t
    assert a.get_stats() == b.get_stats()
    return ca


def test_noalloc_entry_point_equals_az_simulate_net():
    import torch
    from alphazero_amd import engine as E
    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet

    stream = torch.cuda.current_stream().cuda_stream
    evs = {}
    for S in (15, 9):
        torch.manual_seed(S)
        evs[S] = FusedNetEvaluator(PolicyValueNet(S).eval(), S, 0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    # first call: 600 trees of 15x15; second: 300 trees of 9x9 over the first call's leftovers.  Both batches hold
    # more trees than the GPU has CUs: a batch of at most one tree per CU would run as one launch that stages in LDS.
    # third call: four concurrent workers per tree (the staging then holds four leaves per tree)
    for B, S, sims, seed, threads in ((600, 15, 40, 1, 0), (300, 9, 30, 2, 0), (280, 9, 30, 3, 4)):
        assert B > cus
        ev = evs[S]
        fixed, ref = _pair(E, B, S, sims, seed)
        for e in (fixed, ref):
            e.set_threads(threads)  # (allocates the workers' state: the search call itself allocates nothing)
        for ply in range(2):
            torch.cuda.synchronize()
            held, free_before = fixed.device_bytes(), torch.cuda.mem_get_info(0)[0]
            fixed.simulate_net_noalloc(sims, ev.params.data_ptr(), ev.fused_layers, stream)
            torch.cuda.synchronize()
            assert fixed.device_bytes() == held  # the engine allocated nothing ...
            if ply > 0:  # ... and neither did anything below it (the first ply loads the kernels' code)
                assert torch.cuda.mem_get_info(0)[0] == free_before
            ref.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
            counts = _same(fixed, ref)
            act = counts.argmax(axis=1).astype(np.int32)
            fixed.make_move(act)
            ref.make_move(act)
        fixed.close()
        ref.close()
    # beyond the fixed buffer: refused with the limit in the message, nothing launched
    big = E.Engine(4096, 20, nodes_per_tree=2, cpuct=4.0)
    big.set_eps(0.0)
    torch.manual_seed(20)
    ev20 = FusedNetEvaluator(PolicyValueNet(20).eval(), 20, 0)
    with pytest.raises(E.EngineError, match="fixed buffer holds 5242880") as ei:
        big.simulate_net_noalloc(4, ev20.params.data_ptr(), ev20.fused_layers, stream)
    assert ei.value.code == E.ERR_ARG
    assert not big.get_counts()[1].max() >= 0  # no search ran: every root is still unexpanded
    big.close()
