#!/usr/bin/env python3
"""End-to-end self-play / arena throughput with the batched network (not the
headline bench): whole games through `self_play_games_device` or
`eval_games_device`, the library's fused evaluator, Dirichlet noise drawn on the
GPU, one process per GPU with the sample gather over RCCL.

    python tools/bench_selfplay.py [--games 2048] [--shape 15] [--sims 100] [--tau 10] [--mode selfplay|arena]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_selfplay.py --games 65536 --shape 20

`--games` is the total over all ranks (each rank plays its contiguous share).
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
    ap.add_argument("--games", type=int, default=2048)
    ap.add_argument("--shape", type=int, default=15)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--tau", type=int, default=10)
    ap.add_argument("--eps", type=float, default=0.25)
    ap.add_argument("--mode", default="selfplay", choices=["selfplay", "arena"])
    ap.add_argument("--slots", type=int, default=0,
                    help="self-play only: tree slots per GPU; finished games hand their slot to the next game "
                         "(self_play_stream).  0 = one slot per game, all games in lock step")
    args = ap.parse_args()
    import torch

    from alphazero_amd import selfplay as sp
    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    lo, hi = sp.shard_range(args.games, rank, world)
    n_mine = hi - lo
    stream = args.mode == "selfplay" and 0 < args.slots < n_mine
    G = args.slots if stream else n_mine
    nodes = args.sims * 3 + 64  # a move keeps at most the chosen child's subtree
    nets = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        nets.append(FusedNetEvaluator(PolicyValueNet(args.shape).eval(), args.shape, dev, name=f"seed{seed}"))
    for ev in nets:
        sp.broadcast_parameters(ev)  # every rank plays with rank 0's weights
    b0 = sp.TreeBatch(G, args.shape, nets[0], cpuct=4.0, eps=args.eps, nodes_per_tree=nodes, device=local_rank,
                      noise_seed=rank)
    b1 = sp.TreeBatch(G, args.shape, nets[0 if args.mode == "selfplay" else 1], cpuct=4.0, eps=args.eps,
                      nodes_per_tree=nodes, device=local_rank, noise_seed=1000 + rank)
    size = args.shape * args.shape
    if not stream:
        opening = [np.array([(lo + g) % size for g in range(G)], np.int32)]  # distinct first stones
        for act in opening:
            b0.make_move(act)
            b1.make_move(act)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if args.mode == "selfplay":
        if stream:
            games = sp.self_play_stream(b0, b1, args.sims, args.tau, n_mine, rng=np.random.default_rng(rank))
        else:
            games = sp.self_play_games_device(b0, b1, args.sims, args.tau, rng=np.random.default_rng(rank))
        local = sp.all_records(games) if len(games) else np.zeros(0, sp.sample_dtype(args.shape))
        n_local = len(local)
        samples = sp.gather_samples(local)
        tallies = None
    else:
        tallies = sp.eval_games_device(b0, b1, args.sims, args.sims)
        samples, n_local = None, 0
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    st0, st1 = b0.eng.get_stats(), b1.eng.get_stats()
    counters = sp.reduce_counters([st0["sims"] + st1["sims"], n_local, wall] + list(tallies or (0, 0, 0)))
    if rank == 0:
        out = {
            "workload": f"{args.mode}: {args.games} games of {args.shape}x{args.shape}, {args.sims} sims/move, "
                        f"random-init reference net (fused f32 evaluator), eps={args.eps}, {world} GPU(s)"
                        + (f", {G} slots per GPU refilled" if stream else ""),
            "wall_s": wall,
            "sims_per_s": counters[0] / wall,
            "games_per_s": args.games / wall,
        }
        if args.mode == "selfplay":
            out["samples"] = int(len(samples))
            out["samples_per_s"] = len(samples) / wall
            out["gathered_bytes"] = int(samples.nbytes)
        else:
            out["wins_first_net"], out["wins_second_net"], out["draws"] = (int(x) for x in counters[3:6])
        print(json.dumps(out), flush=True)
    b0.close()
    b1.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
