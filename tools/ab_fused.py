#!/usr/bin/env python3
"""A/B timing of the fused search kernel between builds of the library.

    python tools/ab_fused.py libazmcts.so libazmcts_base.so [...]

Each library (file name inside alphazero_amd/) is measured in a child process
of its own, round-robin over `--rounds` rounds, on the bench workload
(1024 trees, 15x15, 800 searches per launch, uniform evaluator); prints the
median launch time per library.  Box-to-box spread is a few percent, so
variants are only comparable inside one run of this tool.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def child(lib, launches):
    import numpy as np
    sys.path.insert(0, ROOT)
    from alphazero_amd import engine as E
    E.LIB_PATH = os.path.join(ROOT, "alphazero_amd", lib)
    import torch
    trees, size = 1024, 225
    eng = E.Engine(trees, 15, nodes_per_tree=4096, cpuct=4.0)
    eng.set_eps(0.0)
    for ply in range(2):
        act = np.array([(g % size, (g % size + 1 + (g // size) % (size - 1)) % size)[ply] for g in range(trees)], np.int32)
        eng.make_move(act)
    p = float(np.float32(1 / 225))
    da = torch.zeros(trees, dtype=torch.int32, device="cuda")
    ms = []
    for it in range(launches):
        eng.timing_begin()
        eng.simulate_fused(800, E.EVAL_UNIFORM, p, 0.0, 0, None, sync=False)
        ms.append(eng.timing_end())
        eng.select_greedy(da.data_ptr())
        eng.make_move_device(da.data_ptr())
        eng.sync_check()
    eng.close()
    print(" ".join(f"{m:.4f}" for m in ms[2:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--launches", type=int, default=12)
    ap.add_argument("--child")
    a = ap.parse_args()
    if a.child:
        child(a.child, a.launches)
        return
    import statistics
    res = {l: [] for l in a.libs}
    for r in range(a.rounds):
        for l in a.libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l, "--launches", str(a.launches)],
                                 capture_output=True, text=True, timeout=120)
            if out.returncode != 0:
                print(l, "FAILED", out.stderr[-400:])
                return 1
            res[l] += [float(x) for x in out.stdout.strip().splitlines()[-1].split()]
    for l in a.libs:
        v = res[l]
        print(f"{l:32s} median {statistics.median(v):.4f} ms  min {min(v):.4f}  ({819200 / statistics.median(v) / 1e3:.1f} Msims/s)")


if __name__ == "__main__":
    sys.exit(main())
