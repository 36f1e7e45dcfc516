"""The root scan of the search kernel divides by the cached N+1 through a
reciprocal and two fused residual corrections instead of a hardware division.
The sequence must be correctly rounded for every input the search can produce
(bit-exact PUCT, mcts/src/node.cpp:105-106); this runs the same IEEE operations
on the host over random and adversarial inputs."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_cached_reciprocal_division_is_correctly_rounded(tmp_path):
    exe = str(tmp_path / "div_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "csrc", "div_check.c"), "-lm"])
    out = subprocess.check_output([exe, "3000000", "4096"], text=True).split()
    tested, bad = int(out[0]), int(out[1])
    assert tested > 4_000_000
    assert bad == 0
