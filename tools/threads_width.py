import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from alphazero_amd import selfplay as sp
from alphazero_amd.evaluator import FusedNetEvaluator
from alphazero_amd.model import PolicyValueNet
S, SIMS = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
ev = FusedNetEvaluator(PolicyValueNet(S).eval(), S)
for trees, threads in ((8192, 0), (8192, 4), (2048, 0), (2048, 4), (512, 0), (512, 4), (512, 8)):
    tb = sp.TreeBatch(trees, S, ev, cpuct=4.0, eps=0.25, threads=threads, nodes_per_tree=SIMS + 64)
    tb.make_move((np.arange(trees) % (S * S)).astype(np.int32))
    tb.simulate(16)
    tb.eng.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tb.simulate(SIMS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"S={S} trees={trees} threads={threads}: {tb.eng.get_stats()['sims']/dt/1e6:.2f} M searches/s, {dt*1e3:.1f} ms", flush=True)
    tb.close()
