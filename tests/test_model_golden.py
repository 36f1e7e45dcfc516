"""The network restatement (alphazero_amd/model.py) and the fused GPU evaluator
against outputs of the reference's own `simplerNN` (src/model.py:9-44) for ITS
parameters (tests/golden/model_S13.npz, made by tests/golden/make_model_golden.py)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_S13.npz"))


def _state():
    import torch

    return {k[len("param:"):]: torch.from_numpy(G[k].copy()) for k in G.files if k.startswith("param:")}


def test_restated_network_loads_reference_parameters_and_matches_on_cpu():
    import torch
    from alphazero_amd.model import PolicyValueNet

    S = int(G["shape"])
    front, pol, val = (int(x) for x in G["layers"])
    net = PolicyValueNet(S, 2, 16, front, pol, val)
    net.load_state_dict(_state(), strict=True)  # same sub-module names and shapes
    net.eval()
    with torch.no_grad():
        value, logp = net(torch.from_numpy(G["boards"]))
    # same torch operators on the same machine: allow only last-bit differences of the conv backend
    assert np.abs(value.numpy().reshape(-1) - G["value"].reshape(-1)).max() < 1e-6
    assert np.abs(logp.numpy() - G["logp"]).max() < 1e-5


@pytest.mark.gpu
def test_fused_kernel_matches_reference_network_outputs():
    import torch
    from alphazero_amd.evaluator import FusedNetEvaluator

    S = int(G["shape"])
    ev = FusedNetEvaluator(_state(), S)
    assert ev.layers == tuple(int(x) for x in G["layers"])
    v, lp = ev(torch.from_numpy(G["boards"]).cuda())
    assert np.abs(v.cpu().numpy() - G["value"].reshape(-1)).max() < 2e-5  # f32 sums in another order
    assert np.abs(lp.cpu().numpy() - G["logp"]).max() < 2e-5
