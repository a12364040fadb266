"""Developer tool: closest-hit queries for random rays in a k*k*k grid of stand-in dragons (the shape of the
incoherent workload): rays per second through rt_trace_closest (host pointers in and out)."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scene as S, layout as L
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon):
    S.write_standin_ply(dragon)
pool = L.align_up(n, L.SORT_TILE)
sc = S.build_scene([dragon, dragon], 64, 64, pool, grid=k)
root = int(sc.data["cam"]["root_group"][0]) & L.GROUP_ID_MASK
first, count = int(sc.groups["w"][root][0]), int(sc.groups["w"][root][1])
boxes = sc.aabbs[first:first + count]
lo, hi = boxes["min"].min(axis=0), boxes["max"].max(axis=0)
rng = np.random.default_rng(0x5EED)
rays = np.zeros(n, L.ray_dtype)
rays["start_min"][:, :3] = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
rays["start_min"][:, 3] = 0.001
d = rng.normal(size=(n, 3))
rays["dir_max"][:, :3] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
rays["dir_max"][:, 3] = np.inf
rt = api.RayTracer(sc, warp_width=64)
rt.trace_closest(rays[:100000])
for rep in range(3):
    t0 = time.perf_counter()
    hits = rt.trace_closest(rays)
    dt = time.perf_counter() - t0
    c = rt.counters()
    print(f"{count} instances, {sc.resident_triangles} instanced triangles, {n} rays: {dt * 1e3:.1f} ms, {n / dt / 1e6:.1f} Mrays/s, "
          f"hit fraction {np.isfinite(hits['t']).mean():.3f}, steps {int(c['steps'])}", flush=True)
