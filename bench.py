#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations/sec, 15x15 gomoku @ 800 sims/move.

Workload (tree-kernel isolation, the configuration the metric is quoted on): 1024 concurrent
trees per GPU, 800 simulations per move, uniform-prior evaluator (p = 1/225,
v = 0) computed inside the kernel, cpuct = 4, eps = 0.  One "step" = one move
of every tree: simulate(800) -> greedy move choice + make_move (re-rooting +
node-pool reclamation) -> finished games cleared and restarted from their
two-ply opening, all inside the timed region and all on the GPU stream.
All inputs live in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

For N > 1 launch under torch.distributed.run (one rank per GPU): every rank
runs its own 1024 trees (weak scaling, no collective in the search), RCCL is
used only for the barrier and to gather the counters.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPE = 15
SIZE = SHAPE * SHAPE
TREES = 1024
SIMS = 800
CPUCT = 4.0
NODES_PER_TREE = 4096
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def algorithmic_bytes_per_sim(depth):
    """Dense SoA minimum per simulation: int32 N + fp64 W + fp64 P = 20 B per child;
    `depth` select calls read SIZE children each, the expansion writes one node,
    takes SIZE fp32 priors in and two bit planes out, the backup touches 16 B per level."""
    return depth * SIZE * 20 + SIZE * 20 + SIZE * 4 + 2 * SIZE / 8 + depth * 16


def cpu_baseline():
    """The oracle (this repo's CPU port of the reference cmcts) on the host cores:
    independent single-thread games, one per core, same 15x15 / 800-sim workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import az_oracle_py as O

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    games, plies = 96 * cores, 50  # about 15 s on 16 host threads
    ev = O.uniform_evaluator(np.float32(1.0 / SIZE))
    total, secs = O.baseline_games(SHAPE, games, cores, plies, SIMS, CPUCT, ev)
    one_total, one_secs = O.baseline_games(SHAPE, 12, 1, plies, SIMS, CPUCT, ev)  # about 2 s
    return {
        "value": total / secs,
        "unit": "sims/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{games} independent games x {plies} plies x {SIMS} sims, one game per thread, "
                  f"{total} sims in {secs:.2f} s",
        "single_thread_value": one_total / one_secs,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--trees", type=int, default=TREES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # rehearsal knobs for a one-GPU box: AZ_BENCH_DEVICE pins every rank to one
    # card, AZ_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks per GPU)
    backend = os.environ.get("AZ_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("AZ_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == "nccl" else torch.device("cpu")

    from alphazero_amd import engine as E

    B = args.trees
    eng = E.Engine(B, SHAPE, nodes_per_tree=NODES_PER_TREE, cpuct=CPUCT, device=dev_index)
    eng.set_alpha_default()
    eng.set_eps(0.0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    prior = float(np.float32(1.0 / SIZE))
    # distinct two-ply opening per tree (global game index keeps ranks distinct)
    op = np.zeros((2, B), np.int32)
    for t in range(B):
        g = rank * B + t
        a0 = g % SIZE
        op[0, t] = a0
        op[1, t] = (a0 + 1 + (g // SIZE) % (SIZE - 1)) % SIZE

    def open_games(mask=None):
        for ply in range(2):
            act = op[ply].copy()
            if mask is not None:
                act[~mask] = -1
            eng.make_move(act, stream)

    open_games()
    d_openings = torch.from_numpy(op).to(dev)                      # [2][B]
    d_restarts = torch.zeros(1, dtype=torch.int32, device=dev)
    timed_slots = []

    def step(timed):
        # everything stays on the stream: no host round trip inside a step (finished
        # games are cleared and re-opened by az_restart_finished, the kernel's HIP
        # events are read after the timed region)
        if timed:
            eng.timing_mark(len(timed_slots), 0, stream)
        eng.simulate_fused(SIMS, E.EVAL_UNIFORM, prior, 0.0, 0, stream, sync=False)
        if timed:
            eng.timing_mark(len(timed_slots), 1, stream)
            timed_slots.append(len(timed_slots))
        eng.make_move_greedy(stream)
        eng.restart_finished(d_openings.data_ptr(), 2, d_restarts.data_ptr(), stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    eng.sync_check(stream)
    eng.reset_stats()
    d_restarts.zero_()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    eng.sync_check(stream)  # a tree in error would have stopped searching: fail loudly
    kernel_ms = [eng.timing_elapsed(i) for i in timed_slots]
    restarts = int(d_restarts.item())

    st = eng.get_stats()
    used, max_used = eng.pool_usage(0)
    sims_local = st["sims"]
    depth = st["select_calls"] / max(1, st["sims"])
    vals = torch.tensor([elapsed, float(sims_local), float(st["select_calls"])], dtype=torch.float64, device=cdev)
    if dist is not None:
        tmax = vals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_sims = float(vals[1])
    depth_all = float(vals[2]) / max(1.0, total_sims)

    if rank == 0:
        value = total_sims / elapsed
        bytes_per_sim = algorithmic_bytes_per_sim(depth)
        avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
        sims_per_launch = B * SIMS
        achieved = sims_per_launch * bytes_per_sim / avg_kernel_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "MCTS simulations/sec (whole node), 15x15 gomoku @800 sims/move",
            "value": value,
            "unit": "sims/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"config2: {SHAPE}x{SHAPE} gomoku, {B} concurrent trees per GPU x {SIMS} sims/move, "
                            "uniform-prior in-kernel evaluator, cpuct=4, eps=0, greedy moves, "
                            "step = one move of every tree",
                "trees_per_gpu": B,
                "sims_per_move": SIMS,
                "nodes_per_tree": NODES_PER_TREE,
                "mean_select_depth": depth_all,
                "restarted_games": restarts,
                "max_live_nodes_per_tree_after_move": max_used,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_simulate_fused",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "bytes_per_sim": bytes_per_sim,
                "sims_per_launch": sims_per_launch,
                "avg_launch_ms": avg_kernel_s * 1e3,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["config"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
