"""Developer tool: steady-state population statistics of the ray pool on the headline scene."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, layout as L, scene as S

dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon):
    S.write_standin_ply(dragon)
W, H = 1920, 1080
warp = int(os.environ.get("RT_WARP", "64"))
sc = S.build_scene([S.BUNNY, dragon], W, H, W * H)
rt = api.RayTracer(sc, warp_width=warp)
rt.init_frame()
rt.make_step(96)
c0 = rt.counters()
for rep in range(2):
    rays, index, groups = rt.dump_state(rays=(rep == 0))
    g = rt.global_data()[0]
    n = int(g["ray_count"])
    keys = index["key"][:n]
    sh = L.group_shader(keys)
    print("live", n, "of", rt.pool, "by shader", dict(zip(["spawn", "sky", "light", "mat", "aabb", "mesh"], np.bincount(sh, minlength=6).tolist())))
    ids = keys & L.GROUP_ID_MASK
    u, cnt = np.unique(ids, return_counts=True)
    print("distinct groups", len(u), "median rays/group", np.median(cnt), "mean", cnt.mean(), "groups<64:", (cnt < 64).sum())
    aab = sh == L.SH_AABB
    boxes = sc.groups["w"][ids[aab]][:, 1]
    print("aabb rays", aab.sum(), "boxes per visit: mean", boxes.mean(), "hist", np.bincount(boxes)[[8, 64, 128, 256]] if boxes.max() >= 256 else np.unique(boxes, return_counts=True))
    mesh = sh == L.SH_MESH
    tris = sc.groups["w"][ids[mesh]][:, 2]
    print("mesh rays", mesh.sum(), "tris per visit mean", tris.mean())
    if rep == 0:
        ql = rays["queue_len"][index["slot"][:n]]
        print("queue_len hist (live)", np.bincount(ql, minlength=9).tolist())
    rt.make_step(1)
c1 = rt.counters()
print("per step: primaries", int(c1["primary_spawned"] - c0["primary_spawned"]) / 2, "shadows", int(c1["shadow_spawned"] - c0["shadow_spawned"]) / 2,
      "box tests", int(c1["box_tests"] - c0["box_tests"]) / 2, "tri tests", int(c1["tri_tests"] - c0["tri_tests"]) / 2)
