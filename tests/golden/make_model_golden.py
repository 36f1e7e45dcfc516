#!/usr/bin/env python3
"""Generates tests/golden/model_S13.npz from the reference's own network class
(`simplerNN`, /root/reference/src/model.py:9-44, with /root/reference/config.py):
its random-init parameters under torch.manual_seed(0) with perturbed batch-norm
statistics, a set of input boards, and the outputs of ITS forward pass in
inference mode on the CPU (float32).

Run in the build container only (the reference does not travel):
    cd /tmp && python /root/repo/tests/golden/make_model_golden.py
The fixture holds parameters, inputs and outputs -- data, not source.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/src")
import config as ref_config  # noqa: E402  (reference modules)
import model as ref_model  # noqa: E402


def main():
    S = ref_model.SHAPE
    torch.manual_seed(0)
    net = ref_model.simplerNN(ref_config)
    g = torch.Generator().manual_seed(1)
    for mod in net.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.rand(mod.running_mean.shape, generator=g) * 0.6 - 0.3)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
            mod.bias.data.copy_(torch.rand(mod.bias.shape, generator=g) * 0.4 - 0.2)
    net.eval()
    rng = np.random.default_rng(2)
    n = 48
    boards = np.zeros((n, 2, S * S), np.float32)
    for k in range(n):
        stones = rng.choice(S * S, size=rng.integers(0, 80), replace=False)
        for t, a in enumerate(stones):
            boards[k, t % 2, a] = 1
    boards = boards.reshape(n, 2, S, S)
    with torch.no_grad():
        value, logp = net(torch.from_numpy(boards))
    out = {"shape": np.int32(S), "boards": boards, "value": value.numpy(), "logp": logp.numpy(),
           "layers": np.array([ref_config.FRONT_LAYER_CNT, ref_config.POLICY_LAYER_CNT, ref_config.VALUE_LAYER_CNT])}
    for k, v in net.state_dict().items():
        out["param:" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "model_S13.npz"), **out)
    print(S, value.shape, logp.shape, len(net.state_dict()))


if __name__ == "__main__":
    main()
