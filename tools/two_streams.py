#!/usr/bin/env python3
"""Experiment: the network search of 8192 trees as ONE batch on one stream against TWO batches of 4096 on two
streams (the tree kernels of one batch can then fill the gaps of the other's evaluator launches).

    python tools/two_streams.py [shape] [sims]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from alphazero_amd import selfplay as sp
from alphazero_amd.evaluator import FusedNetEvaluator
from alphazero_amd.model import PolicyValueNet

S = int(sys.argv[1]) if len(sys.argv) > 1 else 15
SIMS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
torch.manual_seed(0)
ev = FusedNetEvaluator(PolicyValueNet(S).eval(), S)
size = S * S


def batch(n, seed):
    tb = sp.TreeBatch(n, S, ev, cpuct=4.0, eps=0.25, nodes_per_tree=SIMS + 64, noise_seed=seed)
    tb.make_move((np.arange(n) % size).astype(np.int32))
    tb._ensure_noise(SIMS)
    tb.eng.set_active(None)
    return tb


for parts in (1, 2, 4, 1, 2):
    n = 8192 // parts
    tbs = [batch(n, k) for k in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for tb, st in zip(tbs, streams):
        with torch.cuda.stream(st):
            tb.queue_simulate(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # queued in slices so that the streams' launches interleave in submission order too
    for _ in range(SIMS // 20):
        for tb, st in zip(tbs, streams):
            with torch.cuda.stream(st):
                tb.queue_simulate(20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sims = sum(tb.eng.get_stats()["sims"] for tb in tbs) - 8192 * 8
    print(f"S={S} {parts} batch(es) of {n}: {dt*1e3:.1f} ms, {sims/dt/1e6:.2f} M sims/s, {dt/ (SIMS//20*20)*1e3:.4f} ms per search step", flush=True)
    for tb in tbs:
        tb.close()
