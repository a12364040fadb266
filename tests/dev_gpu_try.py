"""Scratch driver: HIP path vs oracle on a small scene (developer tool, not a test)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scene as S, layout as L, api
from oracle import pyoracle as O

W = H = int(os.environ.get("RT_SIZE", "128"))
steps = int(os.environ.get("RT_STEPS", "12"))
sc = S.build_scene([S.BUNNY, S.BUNNY], W, H, W * H)
print("scene", len(sc.groups), sc.group_count, "pool", sc.pool_size, flush=True)

# sort stage
rt = api.RayTracer(sc, warp_width=32, hit_capacity=W * H)
rng = np.random.default_rng(1)
cnt = rt.pool
for n in (cnt, cnt - 32, 8192 + 100, 700, 0):
    pairs = np.zeros(cnt, L.index_pair_dtype)
    pairs["key"] = rng.integers(0, sc.group_count, cnt, dtype=np.uint32) | (rng.integers(0, 256, cnt, dtype=np.uint32) << 24)
    pairs["slot"] = rng.permutation(cnt).astype(np.uint32)
    a = O.sort_pairs(pairs, n, sc.group_count)
    b = rt.sort_pairs(pairs, n, sc.group_count)
    print("sort n=%d equal=%s" % (n, a.tobytes() == b.tobytes()), flush=True)

o = O.Oracle(sc, warp_width=32, hit_capacity=W * H)
o.init_frame(); rt.init_frame()
for s in range(steps):
    o.step(1); rt.make_step(1)
    g0 = o.global_data[0]; g1 = rt.global_data()[0]
    rl, ri, gd = rt.dump_state()
    same_idx = np.array_equal(ri, o.ray_index)
    same_grp = np.array_equal(gd, o.grp_data)
    orl = o.ray_list
    fields = {}
    for name in L.ray_queue_dtype.names:
        a = np.ascontiguousarray(orl[name]); b = np.ascontiguousarray(rl[name])
        fields[name] = a.tobytes() == b.tobytes()
    bad = [k for k, v in fields.items() if not v]
    print(s, "glob", [int(g0[k]) for k in ("pixel_offset", "pixel_count", "old_count", "ray_count")],
          [int(g1[k]) for k in ("pixel_offset", "pixel_count", "old_count", "ray_count")],
          "idx", same_idx, "grp", same_grp, "bad fields", bad, flush=True)
    if not same_idx or bad:
        break
a0 = o.area; a1 = rt.area()
print("area max abs diff", np.abs(a0 - a1).max(), "exact", np.array_equal(a0, a1))
h0 = o.hits; h1 = rt.hits()
print("hits equal", h0.tobytes() == h1.tobytes())
print("counters", o.counters(), rt.counters())
img0 = o.resolve_rgba8(); img1 = rt.draw_frame()
print("img max diff", np.abs(img0.astype(int) - img1.astype(int)).max())
