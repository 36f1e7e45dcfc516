"""bench.py launches its own ranks: `python bench.py --gpus N` with no outer launcher
starts N processes (python -m torch.distributed.run) before touching a GPU and
passes rank 0's single JSON line on.  Rehearsed here without a GPU: AZ_BENCH_STUB=1
swaps the search step for a sleep and RCCL for gloo; the launcher, the rendezvous,
the barrier-bracketed timing and the max/sum reductions are the real ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_plain_invocation_with_two_gpus_launches_two_ranks():
    rec = _run({"AZ_BENCH_STUB": "1"}, "--gpus", "2", "--steps", "7", "--warmup", "2", "--trees", "16")
    assert rec["n_gpus"] == 2 and rec["steps"] == 7 and rec["warmup"] == 2
    assert rec["scaling"] == "weak" and rec["higher_is_better"] is True
    # whole-job value: both ranks' work over the slowest rank's time
    assert abs(rec["value"] * rec["ms_per_step"] * 1e-3 * rec["steps"] - 2 * 16 * 800 * 7) < 1e-3 * 2 * 16 * 800 * 7
    # what an N > 1 line says about the node itself: ranks counted by a collective, one device record per rank,
    # the sample gather (config 4) and the tally reduction (config 5) as secondary blocks
    assert rec["rccl_ranks"] == 2 and rec["backend"] == "gloo"
    assert [d["rank"] for d in rec["devices"]] == [0, 1] and rec["distinct_devices"] == 2
    c4, c5 = rec["secondary"]["config4"], rec["secondary"]["config5"]
    assert c4["records"] == 2 + 3 and c4["gather_complete"] is True and c4["gathered_bytes"] == 5 * 304 and "gather_s" in c4
    assert (c5["wins_first_net"], c5["wins_second_net"], c5["draws"]) == (3, 2, 0)


def test_single_rank_stub_needs_no_launcher():
    rec = _run({"AZ_BENCH_STUB": "1"}, "--steps", "3", "--warmup", "1")
    assert rec["n_gpus"] == 1 and rec["steps"] == 3
