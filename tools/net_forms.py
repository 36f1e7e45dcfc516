#!/usr/bin/env python3
"""Time az_net_eval per board size and kernel form (auto / direct / Winograd / team).

    python tools/net_forms.py [--shapes 15,16,20] [--n 8192] [--reps 20] [--lib libazmcts.so]

Prints one line per (shape, form): milliseconds per launch of n positions and the
direct-form flop rate.  Events on torch's current stream, the stream the kernel runs on.
"""
import argparse
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="15,16,17,18,19,20")
    ap.add_argument("--forms", default="0,1,2")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--kinds", action="store_true", help="pass leaf codes (all AZ_LEAF_EXPAND), as the search does")
    ap.add_argument("--interleave", action="store_true", help="a small elementwise kernel between the launches")
    args = ap.parse_args()
    import torch
    from alphazero_amd import engine as E
    if args.lib:
        E.LIB_PATH = os.path.join(ROOT, "alphazero_amd", args.lib)
    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet

    names = {0: "auto", 1: "direct", 2: "winograd", 3: "team", 4: "wide", 5: "w43"}
    for s in [int(x) for x in args.shapes.split(",")]:
        torch.manual_seed(0)
        ev = FusedNetEvaluator(PolicyValueNet(s).eval(), s)
        b = (torch.rand(args.n, 2, s, s, device="cuda") < 0.2).float()
        kinds = torch.ones(args.n, dtype=torch.int32, device="cuda") if args.kinds else None
        scratch = torch.zeros(1 << 20, device="cuda")
        flops = 2.0 * s * s * (9 * 2 * 16 + 7 * 9 * 16 * 16 + 9 * 16 + 16)
        for f in [int(x) for x in args.forms.split(",")]:
            assert ev.L.az_net_set_form(f) == 0
            for _ in range(3):
                ev(b, kinds)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                ev(b, kinds)
                if args.interleave:
                    scratch.add_(1.0)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            print(f"S={s:2d} form={names[f]:8s} {ms:7.4f} ms  {args.n * flops / ms / 1e9:7.1f} TF (direct-form count)", flush=True)
        ev.L.az_net_set_form(0)


if __name__ == "__main__":
    main()
