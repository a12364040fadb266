"""Developer experiment: N contexts on one GPU, each a row band with 1/N of the pool, stepped
concurrently on their own streams, against one context with the whole frame."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scene as S, tiling
W, H = 1920, 1080
dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon):
    S.write_standin_ply(dragon)
files = [S.BUNNY, dragon]
for parts in (1, 2, 3):
    rts = []
    for r in range(parts):
        first, rows = tiling.band_of(r, parts, H)
        pool = S.L.align_up(W * H // parts, S.L.SORT_TILE)
        sc = S.build_scene(files, W, H, pool, band_first=first, band_rows=rows)
        rt = api.RayTracer(sc, warp_width=64, free_schedule=True)
        rt.init_frame()
        rts.append(rt)
    for k in range(96):
        for rt in rts: rt.make_step(1)
    for rt in rts: rt.sync()
    c0 = [rt.counters() for rt in rts]
    t0 = time.perf_counter()
    K = 64
    for k in range(K):
        for rt in rts: rt.make_step(1)
    for rt in rts: rt.sync()
    dt = time.perf_counter() - t0
    c1 = [rt.counters() for rt in rts]
    rays = sum(int(b["primary_spawned"]) - int(a["primary_spawned"]) + int(b["shadow_spawned"]) - int(a["shadow_spawned"]) for a, b in zip(c0, c1))
    print(f"{parts} context(s): {rays / dt / 1e6:8.1f} Mrays/s, {dt / K * 1e3:.4f} ms per round of steps", flush=True)
    for rt in rts: rt.close()
