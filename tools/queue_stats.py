import os, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np
from raytracer_amd import api, scene as S, layout as L
dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon): S.write_standin_ply(dragon)
sc = S.build_scene([S.BUNNY, dragon], 1920, 1080, 1920*1080)
rt = api.RayTracer(sc, warp_width=64, free_schedule="--reference" not in sys.argv); rt.init_frame(); rt.make_step(120); rt.sync()
rays, index, groups = rt.dump_state()
n = int(rt.global_data()["ray_count"][0])
live = index[:n]
sh = (live["key"] >> 28) & 0xF
ql = rays["queue_len"][live["slot"]]
for name, code in (("aabb", L.SH_AABB if hasattr(L, "SH_AABB") else None), ("mesh", L.SH_MESH if hasattr(L, "SH_MESH") else None)):
    pass
for code in np.unique(sh):
    m = sh == code
    print("shader", int(code), "rays", int(m.sum()), "queue_len hist", np.bincount(ql[m], minlength=9)[:9].tolist(), "mean %.2f" % ql[m].mean())
