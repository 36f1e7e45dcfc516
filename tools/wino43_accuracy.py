#!/usr/bin/env python3
"""Rounding error of the 16 -> 16 layers in float32 Winograd arithmetic against the float64 network, per
form: direct, F(2x2,3x3) (what the kernels use) and F(4x4,3x3) (36 products per 16 outputs instead of 64).
Pure numpy on the CPU; random-init reference network with randomised batch-norm statistics, random boards.

    python tools/wino43_accuracy.py [--shapes 15 20] [--seeds 6]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alphazero_amd.model import PolicyValueNet  # noqa: E402

FORMS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                  [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)),
}


def wino_layer(x, w, shift, m):
    """relu(conv3x3(x, w) + shift) with output tiles of m x m, every step rounded to float32."""
    BT, G, AT = FORMS[m]
    C, S, _ = x.shape
    a, nt = m + 2, (S + m - 1) // m
    pad = np.zeros((C, nt * m + 2, nt * m + 2), np.float32)
    pad[:, 1:S + 1, 1:S + 1] = x
    U = np.einsum("ij,ocjk,lk->ocil", G, w, G).astype(np.float32)  # transformed on the host in float64
    BTf, ATf = BT.astype(np.float32), AT.astype(np.float32)
    out = np.zeros((w.shape[0], nt * m, nt * m), np.float32)
    for ty in range(nt):
        for tx in range(nt):
            d = pad[:, ty * m:ty * m + a, tx * m:tx * m + a]
            V = np.einsum("ij,cjk,lk->cil", BTf, d, BTf).astype(np.float32)
            M = np.einsum("ocil,cil->oil", U, V).astype(np.float32)
            out[:, ty * m:ty * m + m, tx * m:tx * m + m] = np.einsum("ij,ojk,lk->oil", ATf, M, ATf)
    return np.maximum(out[:, :S, :S] + shift[:, None, None].astype(np.float32), 0).astype(np.float32)


def direct_layer(x, w, shift, dt):
    C, S, _ = x.shape
    pad = np.zeros((C, S + 2, S + 2), dt)
    pad[:, 1:S + 1, 1:S + 1] = x
    out = np.zeros((w.shape[0], S, S), dt)
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("oc,cyx->oyx", w[:, :, dy, dx].astype(dt), pad[:, dy:dy + S, dx:dx + S])
    return np.maximum(out + shift[:, None, None].astype(dt), 0)


def folded(seq):
    """(weights with the batch-norm scale folded in, shift) per conv block, float64."""
    layers = []
    for i in range(0, len(seq), 3):
        conv, bn = seq[i], seq[i + 1]
        inv = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        layers.append((conv.weight.detach().double().numpy() * inv.double().numpy()[:, None, None, None],
                       (bn.bias.detach() - bn.running_mean * inv).double().numpy()))
    return layers


def errors(S, seed, form):
    torch.manual_seed(seed)
    net = PolicyValueNet(S).eval()
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 2.0)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.3)
    rng = np.random.default_rng(seed)
    occ = rng.random((S, S))
    board = np.stack([occ < 0.25, (occ >= 0.25) & (occ < 0.5)]).astype(np.float32)
    front, pol, val = folded(net.front), folded(net.policy_conv), folded(net.value_conv)

    def stack(x, layers, dt, first=False):
        for i, (w, sh) in enumerate(layers):
            if dt == np.float64 or form == 0 or (first and i == 0):
                x = direct_layer(x.astype(dt), w, sh, dt)
            else:
                x = wino_layer(x, w, sh, form)
        return x

    outs = []
    for dt in (np.float64, np.float32):
        trunk = stack(board.astype(dt), front, dt, True)
        p, v = stack(trunk, pol, dt), stack(trunk, val, dt)
        pw = net.policy[0].weight.detach().double().numpy().astype(dt)
        pad = np.zeros((p.shape[0], S + 2, S + 2), dt)
        pad[:, 1:S + 1, 1:S + 1] = p
        lg = np.zeros((S, S), dt)
        for dy in range(3):
            for dx in range(3):
                lg += np.einsum("c,cyx->yx", pw[0, :, dy, dx], pad[:, dy:dy + S, dx:dx + S])
        lg = (lg + dt(net.policy[0].bias.item())).ravel().astype(np.float64)
        logp = lg - lg.max() - np.log(np.exp(lg - lg.max()).sum())
        vw = net.value[0].weight.detach().double().numpy().astype(dt)
        value = np.tanh(float(np.einsum("c,cyx->", vw[0, :, 0, 0], v)) / (S * S) + net.value[0].bias.item())
        outs.append((logp, value))
    return np.abs(outs[0][0] - outs[1][0]).max(), abs(outs[0][1] - outs[1][1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", type=int, nargs="+", default=[15, 20])
    ap.add_argument("--seeds", type=int, default=6)
    args = ap.parse_args()
    for S in args.shapes:
        for form, name in ((0, "direct"), (2, "F(2x2,3x3)"), (4, "F(4x4,3x3)")):
            e = [errors(S, s, form) for s in range(args.seeds)]
            print(f"{S}x{S} {name:11s} log-prob error max {max(x[0] for x in e):.2e}   value error max {max(x[1] for x in e):.2e}")


if __name__ == "__main__":
    main()
