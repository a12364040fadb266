"""Helper of test_bench_ranks_cpu.py, not a test: runs bench.py as one rank of a multi-process launch with a stand-in
for the GPU context, so that the benchmark's own multi-rank plumbing -- launcher environment, band per rank, barriers,
the frame all-gather, the reductions over ranks, the single JSON line -- executes on a machine without GPUs.  The
stand-in renders nothing: it counts steps, reports made-up ray counts that differ per rank, and fills its band with
the rank number.  bench.py itself is run unmodified."""
import os
import runpy
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracer_amd import api, layout as L  # noqa: E402

RANK = int(os.environ.get("RANK", "0"))


class StubContext:
    RAYS_PER_STEP = 1000

    def __init__(self, scene, device=0, warp_width=0, sample_limit=0, hit_capacity=0, free_schedule=False, shading=0, mad=False):
        self.scene = scene
        self.pool = L.align_up(scene.pool_size, L.SORT_TILE)
        self.steps = self.profiled = self.mode = 0

    def init_frame(self):
        self.steps = 0

    def make_step(self, n=1):
        self.steps += n
        if self.mode:
            self.profiled += n
        time.sleep(0.002 * n * (1 + RANK))  # later ranks are slower: the job's time is the slowest rank's

    def counters(self):
        per = self.RAYS_PER_STEP * (1 + RANK)
        return {"primary_spawned": 3 * per * self.steps, "shadow_spawned": per * self.steps,
                "box_tests": 400 * per * self.steps, "tri_tests": 100 * per * self.steps}

    def draw_frame(self, out=None):
        if out is None:
            out = np.zeros((self.scene.height, self.scene.width, 4), np.uint8)
        out[...] = 10 + RANK
        return out

    def draw_frame_device(self, device_ptr=None):
        raise AssertionError("the gloo rehearsal gathers through host memory")

    def sync(self):
        pass

    def set_profiling(self, mode):
        self.mode = int(mode)
        if mode:
            self.profiled = 0

    def stage_times(self):
        n = max(self.profiled, 1)
        return {"traverse": 0.15 * n, "shade": 0.0, "sort": 0.03 * n, "compact": 0.0, "leaves": 0.1 * n}, n

    def close(self):
        pass


api.RayTracer = StubContext
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
