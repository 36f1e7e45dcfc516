"""The multi-GPU exchanges over RCCL itself (backend "nccl"), as far as one GPU can take them: a process
group of ONE rank on the box's GPU drives gather_samples (staging in device memory, pieces smaller than
the payload, the re-broadcast of all_ranks=True), reduce_counters and broadcast_parameters through their
device-tensor branches, and the node-level blocks of bench.py (rank census, config-4 gather, config-5
tallies) run on the real backend with small shards.  The two-rank logic is covered on CPU with gloo
(tests/test_distributed_cpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def one_rank_rccl():
    import torch
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_exchanges_over_rccl_with_one_rank(one_rank_rccl):
    import torch
    from alphazero_amd import selfplay as sp
    from alphazero_amd.evaluator import FusedNetEvaluator
    from alphazero_amd.model import PolicyValueNet

    dist = one_rank_rccl
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dt = sp.sample_dtype(9)
    rng = np.random.default_rng(0)
    samples = np.zeros(37, dt)
    samples["board"] = rng.integers(0, 2, samples["board"].shape)
    samples["pi"] = rng.random(samples["pi"].shape)
    samples["r"] = rng.choice([1.0, -1.0, 1e-3], len(samples))
    assert samples.nbytes > 3 * 5000
    got = sp.gather_samples(samples, chunk_bytes=5000)  # the local copy of the destination rank, in pieces
    assert got.dtype == dt and got.tobytes() == samples.tobytes()
    got = sp.gather_samples(samples, all_ranks=True, chunk_bytes=5000)  # + the broadcast through the device staging buffer
    assert got.tobytes() == samples.tobytes()
    assert len(sp.gather_samples(samples[:0], all_ranks=True)) == 0
    assert sp.reduce_counters([3, 4.5, 1e12]).tolist() == [3.0, 4.5, 1e12]  # all_reduce on a device tensor
    torch.manual_seed(3)
    ev = FusedNetEvaluator(PolicyValueNet(9).eval(), 9, 0)
    before = ev.params.clone()
    out = sp.broadcast_parameters(ev)
    assert out[0] is ev.params and out[0].is_cuda and torch.equal(ev.params, before)


def test_bench_node_blocks_on_the_real_backend(one_rank_rccl, monkeypatch):
    """rank_census / node_config4 / node_config5 of bench.py with RCCL carrying every collective."""
    import torch

    sys.path.insert(0, ROOT)
    import bench

    dist = one_rank_rccl
    dev = torch.device("cuda", 0)
    census = bench.rank_census(torch, dist, 0, 1, 0, dev)
    assert census["rccl_ranks"] == 1 and census["backend"] == "nccl" and census["distinct_devices"] == 1
    assert census["devices"][0]["rank"] == 0 and census["devices"][0]["pci"] is not None
    c4 = bench.node_config4(torch, dist, 0, 1, 0, dev, games_per_rank=96, sims=12)
    assert "error" not in c4, c4
    assert c4["gather_complete"] and c4["records"] == c4["records_expected"] > 96
    assert c4["gathered_bytes"] == c4["records"] * (3 * 400 + 1) * 4 and c4["value"] > 0
    c5 = bench.node_config5(torch, dist, 0, 1, 0, dev, games_per_rank=64, sims=12)
    assert "error" not in c5, c5
    assert c5["games_tallied"] == 64 and c5["value"] > 0
