#!/usr/bin/env python3
"""Longer randomized differential run than the test suite's fixed-seed sweeps: the fused search
(every kernel form) and the network search's mask path against the oracle / the three-step path,
concurrent workers against the oracle's lock-step mode, the one-launch cluster search against the
stepped launches, the resident game loop against the host-stepped one.

    python tools/soak.py [--seconds 120] [--seed 1]
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


class _Env:
    """What the tests take as pytest's monkeypatch."""

    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=True):
        os.environ.pop(k, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import test_gpu_search as TS
    import test_gpu_configs as TC
    import test_gpu_workers as TW
    import test_gpu_resident as TR

    rng = np.random.default_rng(args.seed)
    t0, n_fused, n_net, n_workers, n_resident, n_cluster = time.time(), 0, 0, 0, 0, 0
    while time.time() - t0 < args.seconds:
        S = int(rng.integers(5, 21))
        sims = int(rng.integers(20, 700))
        os.environ["AZ_FUSED_LEAN"] = str(int(rng.integers(0, 3)))
        TS._play_fused(S, B=int(rng.integers(1, 6)), sims=sims, plies=int(rng.integers(2, 6)),
                       ev_kind=str(rng.choice(["uniform", "hash"])), eps=float(rng.choice([0.0, 0.1, 0.25, 1.0])),
                       cpuct=float(rng.choice([0.0, 0.5, 4.0, 10.0])), nodes=sims * 8 + 64, seed=int(rng.integers(1, 1000)))
        n_fused += 1
        if n_fused % 8 == 0:
            B, S2 = int(rng.integers(2, 200)), int(rng.choice([5, 6, 7, 9, 13, 14, 15, 16, 17, 19, 20]))
            from alphazero_amd import engine as E
            from conftest import NET_FORMS
            lib = E.load_library()
            try:
                TC.test_simulate_net_position_mask_with_idle_and_terminal_leaves(
                    B, S2, int(rng.integers(2, 6)) if S2 != 5 else 24, bool(rng.integers(0, 2)),
                    lambda form: lib.az_net_set_form(NET_FORMS.get(form, form)))
            finally:
                lib.az_net_set_form(0)
            n_net += 1
        if n_fused % 12 == 0:
            T = int(rng.choice([int(rng.integers(2, 9)), int(rng.integers(9, 65))], p=[0.7, 0.3]))
            TW.test_concurrent_workers_match_the_oracle_lockstep(int(rng.choice([5, 7, 9, 13, 15, 20])), T,
                                                                 int(rng.integers(3, 90 if T <= 8 else 260)), float(rng.choice([0.0, 0.25])))
            n_workers += 1
        if n_fused % 16 == 0:  # the one-launch cluster search against the stepped launches
            T = int(rng.integers(2, 9))
            TW.test_one_launch_cluster_search_equals_the_stepped_launches(
                int(rng.integers(1, 256 // (T + 1) + 1)), int(rng.choice([5, 6, 7, 9, 13, 15, 19, 20])), T, int(rng.integers(2, 120)), _Env())  # (one search is one worker: not the cluster search)
            os.environ.pop("AZ_NET_SEARCH", None)
            n_cluster += 1
        if n_fused % 48 == 0:
            TR.test_resident_self_play_equals_stream_driver(float(rng.choice([0.0, 0.25])), int(rng.integers(-1, 6)),
                                                            int(rng.integers(1, 6)), tuple(int(x) for x in rng.choice(49, size=int(rng.integers(0, 3)), replace=False)))
            n_resident += 1
        print(f"\r{n_fused} fused, {n_net} network, {n_workers} worker, {n_cluster} cluster, {n_resident} resident configurations ok", end="", flush=True)
    print()


if __name__ == "__main__":
    main()
