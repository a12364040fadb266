import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from raytracer_amd import api, scene as S
dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon): S.write_standin_ply(dragon, 934, 466)
sc = S.build_scene([S.BUNNY, dragon], 1920, 1080, 1920 * 1080)
for free in (True, False):
    rt = api.RayTracer(sc, warp_width=64, free_schedule=free)
    rt.init_frame(); rt.make_step(128)
    c0 = rt.counters(); g0 = rt.global_data()[0]
    rt.make_step(32)
    c1 = rt.counters(); g1 = rt.global_data()[0]
    d = {k: (int(c1[k]) - int(c0[k])) / 32 for k in ("primary_spawned", "shadow_spawned", "box_tests", "tri_tests")}
    _, index, groups = rt.dump_state(rays=False)
    n = int(g1["ray_count"])
    sh = index["key"][:n] >> 28
    print("free" if free else "reference", d, "live", n, "box rays", int((sh == 4).sum()), "leaf rays", int((sh == 5).sum()), "spawn", int((index["key"][:n] == 0).sum()), "parked", rt.pool - n)
    rt.close()
