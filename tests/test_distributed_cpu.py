"""The N > 1 host path on CPU: two gloo ranks shard the games, gather their
sample records and sum their counters (the only exchanges of the multi-GPU
design; on GPUs the same calls run over RCCL)."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from alphazero_amd import selfplay as sp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_games = 7
    lo, hi = sp.shard_range(n_games, rank, world)
    dt = sp.sample_dtype(5)
    # each game contributes (game index + 1) records tagged with the game index
    recs = []
    for g in range(lo, hi):
        a = np.zeros(g + 1, dt)
        a["r"] = g
        a["pi"][:, g % 25] = 1
        a["board"][:, 0, g % 5, 0] = 1
        recs.append(a)
    mine = np.concatenate(recs) if recs else np.zeros(0, dt)
    # a piece size far below one rank's payload (a record is 304 bytes), and not a divisor of it
    everything = sp.gather_samples(mine, chunk_bytes=1000)
    np.save(os.path.join(out_dir, f"all_everywhere_{rank}.npy"), sp.gather_samples(mine, all_ranks=True, chunk_bytes=777))
    # uneven and empty shards: rank 1 has nothing to send, rank 0 is not the destination
    lop = mine[:5] if rank == 0 else mine[:0]
    np.save(os.path.join(out_dir, f"lop_{rank}.npy"), sp.gather_samples(lop, dst=1, chunk_bytes=100))
    counters = sp.reduce_counters([len(mine), hi - lo, rank])
    # evaluator replicas start from rank 0's weights
    import torch

    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3), torch.nn.BatchNorm2d(4))
    sp.broadcast_parameters(net)
    flat = torch.cat([t.detach().reshape(-1).double() for t in list(net.parameters()) + list(net.buffers())])
    np.save(os.path.join(out_dir, f"net_{rank}.npy"), flat.numpy())
    np.save(os.path.join(out_dir, f"all_{rank}.npy"), everything)
    np.save(os.path.join(out_dir, f"cnt_{rank}.npy"), counters)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_reduce(tmp_path):
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a0 = np.load(tmp_path / "all_0.npy")
    a1 = np.load(tmp_path / "all_1.npy")
    assert a0.dtype.names == ("board", "pi", "r")
    assert len(a1) == 0 and a1.dtype == a0.dtype  # gathered on rank 0 only
    for r in (0, 1):  # all_ranks=True: every rank holds the same, complete set
        e = np.load(tmp_path / f"all_everywhere_{r}.npy")
        assert np.array_equal(e["r"], a0["r"]) and np.array_equal(e["pi"], a0["pi"]) and np.array_equal(e["board"], a0["board"])
    l0, l1 = np.load(tmp_path / "lop_0.npy"), np.load(tmp_path / "lop_1.npy")
    assert len(l0) == 0 and len(l1) == 5 and np.array_equal(l1["r"], a0["r"][:5]) and np.array_equal(l1["pi"], a0["pi"][:5])
    assert [int(b[0, g % 5, 0]) for b, g in zip(a0["board"], a0["r"].astype(int))] == [1] * len(a0)
    # 7 games, game g has g+1 records, rank order preserved
    assert len(a0) == sum(range(1, 8))
    assert a0["r"].tolist() == [float(g) for g in range(7) for _ in range(g + 1)]
    n0, n1 = np.load(tmp_path / "net_0.npy"), np.load(tmp_path / "net_1.npy")
    assert n0.size > 0 and np.array_equal(n0, n1)
    c0, c1 = np.load(tmp_path / "cnt_0.npy"), np.load(tmp_path / "cnt_1.npy")
    assert c0.tolist() == c1.tolist() == [28.0, 7.0, 1.0]


def test_single_process_helpers_are_identity():
    sys.path.insert(0, ROOT)
    from alphazero_amd import selfplay as sp

    dt = sp.sample_dtype(5)
    a = np.zeros(3, dt)
    assert sp.gather_samples(a) is a
    assert sp.reduce_counters([1, 2]).tolist() == [1.0, 2.0]
