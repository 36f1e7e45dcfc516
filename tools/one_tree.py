#!/usr/bin/env python3
"""Latency of the network search for very few trees (the one-tree cmcts facade, the Gomocup bot):
time per search step of simulate(800) at 1, 8 and 64 trees.

    python tools/one_tree.py [threads [shape [trees]]]     threads > 1: concurrent workers in lock step
"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from alphazero_amd import selfplay as sp
from alphazero_amd.evaluator import FusedNetEvaluator
from alphazero_amd.model import PolicyValueNet
torch.manual_seed(0)
THREADS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
SHAPES = (int(sys.argv[2]),) if len(sys.argv) > 2 else (15, 20)
TREES = (int(sys.argv[3]),) if len(sys.argv) > 3 else (1, 8, 64)
for S in SHAPES:
    ev = FusedNetEvaluator(PolicyValueNet(S).eval(), S)
    for G in TREES:
        tb = sp.TreeBatch(G, S, ev, cpuct=4.0, eps=0.25, threads=THREADS, nodes_per_tree=4096)
        tb.make_move(np.arange(G, dtype=np.int32))
        tb.simulate(50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tb.simulate(800)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"S={S} trees={G}: simulate(800) {dt*1e3:.2f} ms = {dt/800*1e6:.1f} us per search step", flush=True)
        tb.close()
