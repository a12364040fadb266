"""Developer tool: per-band step time and ray rate when the 1080p frame is split into N row bands
(each band rendered alone on this GPU with a full-size pool, as one rank of an N-GPU run would)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scene as S, tiling
W, H = 1920, 1080
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon):
    S.write_standin_ply(dragon)
pool = S.L.align_up(W * H, S.L.SORT_TILE)
rates = []
for r in range(N):
    first, rows = tiling.band_of(r, N, H)
    sc = S.build_scene([S.BUNNY, dragon], W, H, pool, band_first=first, band_rows=rows)
    rt = api.RayTracer(sc, warp_width=64)
    rt.init_frame(); rt.make_step(96); rt.sync()
    c0 = rt.counters(); t0 = time.perf_counter()
    rt.make_step(64); rt.sync()
    dt = time.perf_counter() - t0; c1 = rt.counters()
    rays = int(c1["primary_spawned"]) - int(c0["primary_spawned"]) + int(c1["shadow_spawned"]) - int(c0["shadow_spawned"])
    rates.append((rays, dt))
    print(f"band {r}: rows {first}..{first + rows - 1}  {dt / 64 * 1e3:.4f} ms/step  {rays / dt / 1e6:7.1f} Mrays/s  live {int(rt.global_data()['ray_count'][0])}", flush=True)
    rt.close()
tmax = max(dt for _, dt in rates)
print(f"sum of rays / slowest band: {sum(r for r, _ in rates) / tmax / 1e6:.1f} Mrays/s; sum of rates {sum(r / dt for r, dt in rates) / 1e6:.1f}")
