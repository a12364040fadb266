"""bench.py's N > 1 path, launched the way the driver launches it (python -m torch.distributed.run, one process per
rank), on the CPU: two gloo ranks with a stand-in for the GPU context (tests/bench_rank_stub.py).  What is checked is
the benchmark's plumbing, not a rate: every rank gets its band, the barriers and the all-gather complete, rank 0
prints exactly one JSON line, and `value` is the rays of all ranks over the slowest rank's time."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_print_one_line_with_whole_job_throughput():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    steps = 6
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "bench_rank_stub.py"), "--gpus", "2", "--steps", str(steps),
           "--warmup", "2", "--settle", "8", "--backend", "gloo", "--scene", "bunny", "--width", "256", "--height", "144"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "weak" and d["unit"] == "Mrays/s"
    assert d["metric"].startswith("Mrays/s") and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "2 row band(s)" in d["config"]["tiling"] and "gloo" in d["config"]["tiling"]
    # the stub starts 4,000 rays per step on rank 0 and 8,000 on rank 1; rank 1 is the slower one
    rays = steps * (4000 + 8000)
    seconds = d["ms_per_step"] * 1e-3 * steps
    assert abs(d["value"] - rays / seconds / 1e6) < 1e-6 * d["value"]
    assert seconds >= steps * 0.004                      # at least rank 1's sleeps: the maximum over ranks, not rank 0's time
    assert abs(d["mrays_primary_only"] / d["value"] - 0.75) < 1e-9
    assert "extra" not in d and "cpu_baseline" not in d   # single-GPU legs only
    assert d["roofline"]["kernel"] == "k_process_aabb" and d["roofline"]["frac"] > 0


def test_cpu_baseline_leg_on_a_small_frame():
    """The bench line's `cpu_baseline` object (the oracle on the host cores, a middle band of the frame) with a budget of
    a second: the keys the line promises, a positive rate, the sample named, the arithmetic mode passed on."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    from raytracer_amd import scene as S
    for arith in ("mad", "strict"):
        args = argparse.Namespace(width=256, height=128, arith=arith)
        out = bench.cpu_baseline(args, [S.BUNNY, S.BUNNY], 1.0)
        assert set(out) == {"value", "unit", "cores", "kind", "sample"}
        assert out["kind"] == "port" and out["unit"] == "Mrays/s" and 1 <= out["cores"] <= 16
        assert out["value"] > 0 and "rows 56..71 of the 256x128 frame, pool 4096 slots" in out["sample"]
