"""The library's own network kernel (az_net_eval through FusedNetEvaluator)
against PyTorch's float64 evaluation of the same module on the CPU.  Network
outputs are a floating-point result: tolerance 2e-5 absolute on value and
log-probabilities (f32 products and sums in a different order than any BLAS)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(shape, front, pol, val, seed):
    import torch
    from alphazero_amd.model import PolicyValueNet

    torch.manual_seed(seed)
    m = PolicyValueNet(shape, 2, 16, front, pol, val)
    # non-trivial inference statistics
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.3, 0.3)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.2, 0.2)
    return m.eval()


def _boards(n, shape, seed):
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 2, shape * shape), np.float32)
    for k in range(n):
        stones = rng.choice(shape * shape, size=rng.integers(0, shape * shape // 2), replace=False)
        for t, a in enumerate(stones):
            b[k, t % 2, a] = 1
    return b.reshape(n, 2, shape, shape)


@pytest.mark.parametrize("form", ["auto", "direct", "winograd", "team", "wide", "w43"])
@pytest.mark.parametrize("shape,front,pol,val,n", [(15, 4, 2, 2, 300), (13, 4, 2, 2, 50), (20, 3, 1, 2, 40),
                                                     (7, 1, 0, 0, 9), (5, 2, 3, 1, 1100), (14, 2, 0, 1, 70),
                                                     (12, 1, 1, 0, 33), (10, 3, 1, 1, 20), (6, 2, 1, 1, 5),
                                                     (11, 1, 0, 2, 64), (9, 2, 2, 0, 130), (8, 3, 0, 0, 17),
                                                     (16, 4, 2, 2, 90), (17, 2, 1, 3, 30), (18, 1, 0, 1, 21), (19, 3, 3, 0, 40),
                                                     (20, 4, 2, 2, 600),
                                                     (15, 1, 0, 0, 40), (15, 2, 0, 1, 33), (16, 1, 1, 0, 25), (15, 3, 1, 0, 2100),
                                                     (16, 2, 0, 2, 19), (15, 1, 2, 2, 64), (16, 3, 3, 1, 2300),
                                                     (13, 1, 0, 0, 30), (13, 3, 1, 1, 2200), (14, 4, 2, 2, 2100), (14, 1, 2, 0, 45)])
def test_fused_net_matches_torch_float64(shape, front, pol, val, n, form, net_form):
    """Every kernel form (one wavefront per position with direct or Winograd 3x3 layers, team of
    wavefronts, one wavefront per SIMD) against PyTorch float64, over every branch layout (no policy /
    value branch, trunk of one)."""
    import torch
    from alphazero_amd.evaluator import FusedNetEvaluator

    if shape > 15 and form == "direct":
        pytest.skip("above 15x15 the direct form is the team kernel")
    if form == "w43" and shape < 13:
        pytest.skip("the F(4x4,3x3) kernels are built for 13x13 and up")
    m = _model(shape, front, pol, val, 3)
    boards = _boards(n, shape, 5)
    with torch.no_grad():
        v_ref, lp_ref = m.double()(torch.from_numpy(boards).double())
    ev = FusedNetEvaluator(m.float(), shape)
    net_form(form)
    v, lp = ev(torch.from_numpy(boards).cuda())
    v, lp = v.cpu().numpy(), lp.cpu().numpy()
    assert v.shape == (n,) and lp.shape == (n, shape * shape)
    assert np.abs(v - v_ref.reshape(-1).numpy()).max() < 2e-5
    assert np.abs(lp - lp_ref.numpy()).max() < 2e-5
    assert np.allclose(np.exp(lp.astype(np.float64)).sum(axis=1), 1.0, atol=1e-5)


def test_fused_net_drives_a_search():
    import torch
    from alphazero_amd import selfplay as sp
    from alphazero_amd.evaluator import FusedNetEvaluator, TorchEvaluator

    S, G, sims = 9, 6, 40
    m = _model(S, 4, 2, 2, 1)
    fused = sp.TreeBatch(G, S, FusedNetEvaluator(m, S), cpuct=4.0, eps=0.0, nodes_per_tree=sims * 4)
    ref = sp.TreeBatch(G, S, TorchEvaluator(m), cpuct=4.0, eps=0.0, nodes_per_tree=sims * 4)
    for b in (fused, ref):
        b.make_move(np.arange(G, dtype=np.int32) * 3)
        b.simulate(sims)
    # same network, two float32 evaluation orders: the visit distributions agree closely
    assert np.abs(fused.probs() - ref.probs()).max() < 0.08
    assert (fused.probs().argmax(axis=1) == ref.probs().argmax(axis=1)).mean() >= 0.5
    fused.close()
    ref.close()


def test_torchscript_file_of_the_reference_architecture_gets_the_fused_kernel(tmp_path):
    import torch
    from alphazero_amd.evaluator import FusedNetEvaluator, TorchEvaluator, load_evaluator
    from alphazero_amd.model import PolicyValueNet

    S = 13
    m = _model(S, 4, 2, 2, 9)
    path = str(tmp_path / "net.pt")
    torch.jit.trace(m, torch.rand(1, 2, S, S)).save(path)  # what src/self_play.py:62-66 writes
    ev = load_evaluator(path, S)
    assert isinstance(ev, FusedNetEvaluator)
    boards = torch.from_numpy(_boards(17, S, 2)).cuda()
    v, lp = ev(boards)
    v2, lp2 = TorchEvaluator(m)(boards)
    assert (v - v2).abs().max().item() < 5e-5 and (lp - lp2).abs().max().item() < 5e-5
    # another architecture keeps the PyTorch path
    torch.manual_seed(0)
    other = PolicyValueNet(S, conv_channels=8, front_layers=2, policy_layers=1, value_layers=1).eval()
    path2 = str(tmp_path / "other.pt")
    torch.jit.trace(other, torch.rand(1, 2, S, S)).save(path2)
    assert isinstance(load_evaluator(path2, S), TorchEvaluator)


def test_fused_net_skips_leaves_that_need_no_evaluation():
    """az_net_eval with the step's leaf codes: boards coded AZ_LEAF_EXPAND get the
    same outputs as without the codes, all other rows stay untouched."""
    import torch
    from alphazero_amd.evaluator import FusedNetEvaluator

    shape, n = 15, 700
    size = shape * shape
    ev = FusedNetEvaluator(_model(shape, 4, 2, 2, 8).float(), shape)
    boards = torch.from_numpy(_boards(n, shape, 2)).cuda()
    v_all, lp_all = ev(boards)
    kinds = torch.from_numpy(np.random.default_rng(4).integers(0, 3, n).astype(np.int32)).cuda()  # NONE / EXPAND / TERMINAL
    value = torch.full((n,), 7.0, dtype=torch.float32, device="cuda")
    logp = torch.full((n, size), 7.0, dtype=torch.float32, device="cuda")
    rc = ev.L.az_net_eval(shape, n, boards.data_ptr(), kinds.data_ptr(), ev.params.data_ptr(), *ev.layers,
                          value.data_ptr(), logp.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    need = (kinds == 1).cpu().numpy()
    assert 0 < need.sum() < n
    value, logp = value.cpu().numpy(), logp.cpu().numpy()
    assert np.array_equal(value[need], v_all.cpu().numpy()[need])
    assert np.array_equal(logp[need], lp_all.cpu().numpy()[need])
    assert np.all(value[~need] == 7.0) and np.all(logp[~need] == 7.0)


@pytest.mark.parametrize("shape,form", [(13, "auto"), (14, "auto"), (17, "w43"), (18, "w43"), (15, "auto"), (20, "auto")])
def test_outputs_do_not_depend_on_the_batch_slot(shape, form, net_form):
    """A position's value and log-probabilities are a function of the position alone: the same
    board placed in several slots of a batch larger than the launch grid (so that workgroups
    evaluate it after different predecessors, on buffers holding different leftovers) gives
    bit-identical outputs.  Covers the boards whose tile grid overhangs the board (13, 14, 17,
    18), where the input layer's patches reach past the two input planes."""
    import torch
    from alphazero_amd.evaluator import FusedNetEvaluator

    n = 2048 * 3 + 77  # more than eight positions per CU on 256 CUs, several rounds per workgroup
    m = _model(shape, 4, 2, 2, 11)
    boards = _boards(n, shape, 13)
    probe = boards[5].copy()
    slots = [0, 1, 63, 2047, 2048, 2049, 3000, 4095, 4096, 5000, n - 2, n - 1]
    boards[slots] = probe
    ev = FusedNetEvaluator(m.float(), shape)
    net_form(form)
    v, lp = ev(torch.from_numpy(boards).cuda())
    v, lp = v.cpu().numpy(), lp.cpu().numpy()
    for s in slots[1:]:
        assert v[s] == v[slots[0]], s
        assert np.array_equal(lp[s], lp[slots[0]]), s
    # and the same position as every workgroup's first, on freshly zeroed buffers
    net_form("w43" if shape >= 13 and form == "auto" else form)  # (a batch of one would get the latency kernel)
    v1, lp1 = ev(torch.from_numpy(np.repeat(probe[None], 300, axis=0)).cuda())
    assert v1.cpu().numpy()[0] == v[slots[0]] and np.array_equal(lp1.cpu().numpy()[0], lp[slots[0]])
