#!/usr/bin/env python3
"""Long runs of the one-launch cluster search (k_net_search_workers) against the stepped launches: many steps
per call, grids up to one workgroup per CU, every statistic of every tree compared bit for bit.  The exchange
between the workgroups of a cluster rests on memory-ordering properties no unit test can pin in a few steps;
this is the run that would show a torn or stale word.

    python tools/cluster_stress.py [--seconds 300] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    from alphazero_amd import engine as E
    from test_gpu_configs import _fused
    import az_oracle_py as O

    rng = np.random.default_rng(args.seed)
    stream = torch.cuda.current_stream().cuda_stream
    t0, runs, steps = time.time(), 0, 0
    while time.time() - t0 < args.seconds:
        S = int(rng.choice([9, 13, 15, 19, 20]))
        T = int(rng.integers(2, 9))
        B = int(rng.integers(max(1, 256 // (T + 1) - 12), 256 // (T + 1) + 1)) if rng.random() < 0.7 else int(rng.integers(1, 5))
        sims = int(rng.integers(200, 1600))
        size = S * S
        _, ev = _fused(S, seed=int(rng.integers(0, 100)))
        engs = []
        noise_seed = int(rng.integers(0, 1 << 30))
        for _ in range(2):
            e = E.Engine(B, S, nodes_per_tree=sims + 2 * T + 64, cpuct=4.0)
            e.set_alpha_default()
            e.set_eps(0.25)
            e.set_threads(T)
            e.set_noise_table(np.random.default_rng(noise_seed).dirichlet(np.full(size, e.get_alpha()), size=sims + 2))
            e.make_move(np.array([O.opening(size, 3 * g + 2)[0] for g in range(B)], np.int32))
            engs.append(e)
        a, b = engs
        os.environ.pop("AZ_NET_SEARCH", None)
        a.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        assert a.get_net_path() == E.NET_PATH_CLUSTER, (a.get_net_path(), B, S, T)
        os.environ["AZ_NET_SEARCH"] = "0"
        b.simulate_net(sims, ev.params.data_ptr(), ev.fused_layers, stream)
        os.environ.pop("AZ_NET_SEARCH", None)
        ca, ra = a.get_counts()
        cb, rb = b.get_counts()
        assert np.array_equal(ra, rb) and np.array_equal(ca, cb), (B, S, T, sims)
        assert a.get_stats() == b.get_stats(), (B, S, T, sims)
        for t in range(0, B, max(1, B // 8)):
            na, nb = a.get_node(t, []), b.get_node(t, [])
            assert np.array_equal(na[2], nb[2]) and np.array_equal(na[3], nb[3]), (B, S, T, sims, t)
        a.close()
        b.close()
        runs += 1
        steps += -(-sims // T)
        print(f"\r{runs} calls, {steps} cluster steps, last: {B} trees of {S}x{S}, {T} workers, {sims} searches: ok", end="", flush=True)
    print()


if __name__ == "__main__":
    main()
