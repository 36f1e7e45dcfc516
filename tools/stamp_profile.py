#!/usr/bin/env python3
"""Phase breakdown of the fused search kernel from in-kernel cycle stamps.

Needs the diagnostic library: `make -C alphazero_amd/csrc stamps`
(libazmcts_stamps.so, 15x15 only).  Never used for throughput numbers: the
stamps serialise the instruction stream.  Phase boundaries are approximate: the
compiler moves vector instructions across a stamp (the next root scan, for
one, is scheduled above the loop-top stamp), so read the split together with
the ISA.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from alphazero_amd import engine as E  # noqa: E402

E.LIB_PATH = os.path.join(ROOT, "alphazero_amd", os.environ.get("AZ_STAMP_LIB", "libazmcts_stamps.so"))
NAMES = ["root_select", "root_move", "node_load_wait", "node_select", "expand", "node_move", "backup", "loop_top"]


def main():
    trees = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    eng = E.Engine(trees, 15, nodes_per_tree=4096, cpuct=4.0)
    eng.L.az_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    eng.set_eps(0.0)
    size = 225
    for ply in range(2):
        act = np.array([(g % size, (g % size + 1 + (g // size) % (size - 1)) % size)[ply] for g in range(trees)], np.int32)
        eng.make_move(act)
    p = float(np.float32(1 / 225))
    for it in range(3):
        eng.timing_begin()
        eng.simulate_fused(800, E.EVAL_UNIFORM, p, 0.0, 0, None, sync=False)
        ms = eng.timing_end()
        out = np.zeros((trees, 8), np.uint64)
        eng._check(eng.L.az_debug_stamps(eng.h, out.ctypes.data))
        xcc = (out[:, 6] >> np.uint64(40)).astype(np.int64)
        hw = (out[:, 7] >> np.uint64(40)).astype(np.int64)
        out &= np.uint64((1 << 40) - 1)
        if os.environ.get("AZ_PF_STATS"):
            hit = (out[:, 5] >> np.uint64(20)).sum()
            tot = (out[:, 5] & np.uint64((1 << 20) - 1)).sum()
            print(f"prefetch hint: {hit} of {tot} root picks with a child were the hinted record ({100.0 * hit / max(1, tot):.1f} %)")
            out[:, 5] = 0
        per_tree = out.sum(axis=1)
        # placement: HW_ID = wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
        simd_key = xcc * 65536 + (hw & 0xff30)
        cu_key = xcc * 65536 + (hw & 0xff00)
        _, per_simd = np.unique(simd_key, return_counts=True)
        _, per_cu = np.unique(cu_key, return_counts=True)
        print(f"placement: {len(per_simd)} SIMDs used (max {per_simd.max()} trees on one), "
              f"{len(per_cu)} CUs used (max {per_cu.max()}), XCCs {np.bincount(xcc).tolist()}")
        pt = np.sort(per_tree)
        print(f"cycles per tree: min {pt[0]} p10 {pt[len(pt) // 10]} median {pt[len(pt) // 2]} "
              f"p90 {pt[len(pt) * 9 // 10]} max {pt[-1]}")
        for cnt in np.unique(per_simd):
            keys = np.unique(simd_key)[per_simd == cnt]
            sel = np.isin(simd_key, keys)
            print(f"   trees sharing a SIMD with {cnt - 1} others: {sel.sum()}, mean cycles {per_tree[sel].mean():.0f}")
        tot = per_tree.mean()
        print(f"launch {it}: {ms:.3f} ms, mean cycles/tree {tot:.0f} ({tot / 800:.0f} per search)")
        for n, v in zip(NAMES, out.mean(axis=0)):
            print(f"   {n:16s} {v / 800:8.0f} cycles/search  {100 * v / tot:5.1f} %")
        d = np.zeros(trees, np.int32)
        import torch
        da = torch.zeros(trees, dtype=torch.int32, device="cuda")
        eng.select_greedy(da.data_ptr())
        eng.make_move_device(da.data_ptr())
        eng.sync_check()
    eng.close()


if __name__ == "__main__":
    main()
