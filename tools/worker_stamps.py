#!/usr/bin/env python3
"""Where a worker kernel (k_workers_backup_select) spends its time, from in-kernel cycle stamps of the master
wavefront.  Needs the diagnostic library: `make -C alphazero_amd/csrc stamps` (15x15 only).

    python tools/worker_stamps.py [threads [trees]]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from alphazero_amd import engine as E  # noqa: E402

E.LIB_PATH = os.path.join(ROOT, "alphazero_amd", os.environ.get("AZ_STAMP_LIB", "libazmcts_stamps.so"))
import torch  # noqa: E402
from alphazero_amd import selfplay as sp  # noqa: E402
from alphazero_amd.evaluator import FusedNetEvaluator  # noqa: E402
from alphazero_amd.model import PolicyValueNet  # noqa: E402

NAMES = ["header", "backup phase", "tail stores", "descent: wait for the record", "descent: pick", "descent: stores + child request",
         "descent: rules", "descent: leaf (new child, board out)"]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
ev = FusedNetEvaluator(PolicyValueNet(15).eval(), 15)
tb = sp.TreeBatch(G, 15, ev, cpuct=4.0, eps=0.25, threads=T, nodes_per_tree=4096)
tb.eng.L.az_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
tb.make_move(np.arange(G, dtype=np.int32))
tb.simulate(50)


def stamps():
    out = np.zeros((G, 8), np.uint64)
    tb.eng._check(tb.eng.L.az_debug_stamps(tb.eng.h, out.ctypes.data))
    return out.astype(np.int64)


before = stamps()
torch.cuda.synchronize()
t0 = time.perf_counter()
tb.simulate(800)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
d = (stamps() - before).mean(axis=0)
steps = 800 // T
print(f"threads={T} trees={G}: simulate(800) {dt * 1e3:.2f} ms, {steps} steps; stamp units per step (master wavefront):")
for name, v in zip(NAMES, d):
    print(f"  {name:24s} {v / steps:9.1f}")
print(f"  {'sum':24s} {d.sum() / steps:9.1f}")
st = tb.eng.get_stats()
print("select calls per search:", st["select_calls"] / st["sims"])
