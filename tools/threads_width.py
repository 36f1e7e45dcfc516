#!/usr/bin/env python3
"""Network-search throughput by batch width and worker count (one simulate(SIMS) per line).

    python tools/threads_width.py [shape [sims [trees,trees,... [threads,threads,...]]]]

AZ_NET_GROUPS=<n> in the environment forces the number of tree groups (A/B runs).
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from alphazero_amd import selfplay as sp  # noqa: E402
from alphazero_amd.evaluator import FusedNetEvaluator  # noqa: E402
from alphazero_amd.model import PolicyValueNet  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 15
SIMS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
TREES = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8192, 2048, 512]
THREADS = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 4]
torch.manual_seed(0)
ev = FusedNetEvaluator(PolicyValueNet(S).eval(), S)
for trees in TREES:
    for threads in THREADS:
        tb = sp.TreeBatch(trees, S, ev, cpuct=4.0, eps=0.25, threads=threads, nodes_per_tree=SIMS + 64)
        tb.make_move((np.arange(trees) % (S * S)).astype(np.int32))
        tb.simulate(16)
        tb.eng.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tb.simulate(SIMS)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"S={S} trees={trees} threads={threads}: {tb.eng.get_stats()['sims'] / dt / 1e6:.2f} M searches/s, {dt * 1e3:.1f} ms "
              f"(path {tb.eng.get_net_path()})", flush=True)
        tb.close()
