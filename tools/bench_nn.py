#!/usr/bin/env python3
"""Stepped search with a batched policy/value network (not the headline bench).

    python tools/bench_nn.py [--trees 8192] [--shape 15] [--sims 64] [--moves 2] [--evaluator fused|torch]

Many concurrent trees, one leaf per tree per search step packed into a single
network evaluation (random-init PolicyValueNet, seed 0; the library's fused
kernel or a PyTorch-ROCm forward); reports
simulations/s and how the step time splits between the tree kernels
(az_select + az_expand_backup) and the evaluator.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=8192)
    ap.add_argument("--shape", type=int, default=15)
    ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--moves", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--dtype", default="float32", choices=["float32", "bfloat16"])
    ap.add_argument("--evaluator", default="fused", choices=["fused", "torch"],
                    help="fused = the library's own f32 matrix-core kernel, torch = PyTorch-ROCm (MIOpen)")
    args = ap.parse_args()
    import torch

    from alphazero_amd import selfplay as sp
    from alphazero_amd.evaluator import FusedNetEvaluator, TorchEvaluator
    from alphazero_amd.model import PolicyValueNet

    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    net = PolicyValueNet(args.shape).eval()
    if args.evaluator == "fused":
        ev = FusedNetEvaluator(net, args.shape, dev)
        args.dtype = "float32"
    elif args.dtype == "bfloat16":
        net = net.to(torch.bfloat16)

        class Bf16Eval(TorchEvaluator):
            @torch.no_grad()
            def __call__(self, boards):
                return super().__call__(boards.to(torch.bfloat16))
        ev = Bf16Eval(net, dev)
    else:
        ev = TorchEvaluator(net, dev)
    tb = sp.TreeBatch(args.trees, args.shape, ev, cpuct=4.0, eps=0.25, nodes_per_tree=args.nodes)
    size = args.shape * args.shape
    act = np.array([g % size for g in range(args.trees)], np.int32)
    tb.make_move(act)
    tb.simulate(8)  # warm-up (MIOpen kernel selection, allocator)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev).cuda_stream
    eng = tb.eng
    t_tree = t_eval = 0.0
    t0 = time.perf_counter()
    n_steps = 0
    for mv in range(args.moves):
        eng.set_active(None)
        tb._ensure_noise(args.sims)
        eng.begin_simulate(stream)
        with torch.no_grad():
            for _ in range(args.sims):
                e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
                e0.record()
                eng.select(tb.boards.data_ptr(), tb.kinds.data_ptr(), stream)
                e1.record()
                value, prior = ev(tb.boards)
                e2.record()
                eng.expand_backup(value.data_ptr(), prior.data_ptr(), ev.log_priors, stream)
                e3.record()
                e3.synchronize()
                t_tree += e0.elapsed_time(e1) + e2.elapsed_time(e3)
                t_eval += e1.elapsed_time(e2)
                n_steps += 1
        eng.sync_check(stream)
        counts, _ = eng.get_counts()
        tb.make_move(counts.argmax(axis=1).astype(np.int32))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st = eng.get_stats()
    print(json.dumps({
        "workload": f"{args.shape}x{args.shape}, {args.trees} concurrent trees, random-init conv policy/value net "
                    f"({args.evaluator}, {args.dtype}), batched leaf evaluation, eps=0.25",
        "sims_per_s": args.trees * n_steps / wall,
        "ms_per_step": wall / n_steps * 1e3,
        "tree_kernels_ms_per_step": t_tree / n_steps,
        "evaluator_ms_per_step": t_eval / n_steps,
        "tree_kernel_sims_per_s": args.trees / (t_tree / n_steps / 1e3),
        "mean_select_depth": st["select_calls"] / max(1, st["sims"]),
        "device_bytes": eng.device_bytes(),
        # 3x3 convolutions of the reference configuration (4 + 2 + 2 layers of 16 channels, 2 input
        # planes) + heads, 2 flops per multiply-add; priced against the dense f32 matrix-core peak
        "evaluator_roofline": {
            "bound": "mfma", "unit": "TFLOP/s", "peak": 157.3,
            "achieved": args.trees * 2.0 * args.shape ** 2 * (9 * 2 * 16 + 7 * 9 * 16 * 16 + 9 * 16 + 16)
                        / (t_eval / n_steps / 1e3) / 1e12,
        },
    }))
    tb.close()


if __name__ == "__main__":
    main()
