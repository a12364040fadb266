"""bench.py itself: the one-rank run of the multi-GPU path (torch.distributed over RCCL, the resolved band
all-gathered inside the timed region) and the JSON line's contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", "bunny", "--width", "512", "--height", "256", "--steps", "16",
           "--warmup", "4", "--settle", "24", "--no-extras", "--cpu-seconds", "0", *flags]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_line_and_the_rccl_gather_with_one_rank():
    plain = run_bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in plain, key
    assert plain["value"] > 0 and plain["n_gpus"] == 1 and plain["config"]["tiling"] == "single GPU"
    assert set(plain["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    gathered = run_bench("--collective", "--arith", "mad")
    assert gathered["config"]["tiling"].startswith("1 row band(s), RGBA8 all-gather per frame (nccl)")
    assert gathered["config"]["arithmetic"] == "mad"
    assert gathered["value"] > 0 and gathered["frame_resolve_gather_ms"] > 0
