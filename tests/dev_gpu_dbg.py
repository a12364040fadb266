import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scene as S, layout as L, api
from oracle import pyoracle as O
W = H = 128
sc = S.build_scene([S.BUNNY, S.BUNNY], W, H, W * H)
rt = api.RayTracer(sc, warp_width=32, hit_capacity=W * H)
o = O.Oracle(sc, warp_width=32, hit_capacity=W * H)
o.init_frame(); rt.init_frame()
rl, ri, gd = rt.dump_state()
print("init equal idx", np.array_equal(ri, o.ray_index), "rays", rl.tobytes() == o.ray_list.tobytes())
o.step(1); rt.make_step(1)
rl, ri, gd = rt.dump_state()
orl = o.ray_list
# compare per slot
ql = orl["queue_len"]; gl = rl["queue_len"]
print("queue_len equal", np.array_equal(ql, gl), "mismatch count", (ql != gl).sum())
bad = np.nonzero(ql != gl)[0]
print("first bad slots", bad[:10])
for s in bad[:3]:
    print("slot", s, "oracle len", ql[s], "gpu len", gl[s])
    print(" oracle q", orl["queue"][s][:4])
    print(" gpu    q", rl["queue"][s][:4])
    print(" o min/max", orl["start_min"][s], orl["dir_max"][s], " g", rl["start_min"][s], rl["dir_max"][s])
keys_o = np.sort(o.ray_index["key"]); keys_g = np.sort(ri["key"])
print("key multiset equal", np.array_equal(keys_o, keys_g))
u, c = np.unique(ri["key"], return_counts=True); print("gpu keys", [(hex(a), b) for a, b in zip(u[:8], c[:8])])
u, c = np.unique(o.ray_index["key"], return_counts=True); print("orc keys", [(hex(a), b) for a, b in zip(u[:8], c[:8])])
