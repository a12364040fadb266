"""Developer tool: per-wave start/end times of one launch of the box kernel (needs the RT_STAMPS=3 build)."""
import ctypes, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "csrc", "librt_hip_stamps3.so")
os.environ["RT_NO_OVERLAP"] = "1"
from raytracer_amd import api, scene as S
dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon):
    S.write_standin_ply(dragon)
sc = S.build_scene([S.BUNNY, dragon], 1920, 1080, 1920 * 1080)
rt = api.RayTracer(sc, warp_width=64, free_schedule=os.environ.get("RT_SCHEDULE", "free") == "free")
rt.init_frame(); rt.make_step(120); rt.sync()
ptr = rt.debug_stamps()[7]
n = rt.pool // 64
tr = np.zeros((n, 4), np.uint64)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
assert hip.hipMemcpy(tr.ctypes.data, ctypes.c_void_p(ptr), tr.nbytes, 2) == 0
t0 = tr[:, 0].astype(np.int64); t1 = tr[:, 1].astype(np.int64)
work = t1 > 0
base = t0.min()
s = (t0 - base) / 100.0; e = np.where(work, (t1 - base) / 100.0, s)   # us (100 MHz)
print("waves", n, "working", work.sum(), "kernel span us", e.max())
print("working wave duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (
    (e - s)[work].mean(), *np.percentile((e - s)[work], [10, 50, 90]), (e - s)[work].max()))
# resident working waves over time
T = np.arange(0, e.max(), 5.0)
for t in T:
    live = ((s <= t) & (e > t) & work).sum()
    started = (s <= t).sum()
    print("t=%6.1f us  working waves resident %5d   waves started %6d" % (t, live, started))
hw = tr[:, 2]
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
print("working waves per XCC", [int(((xcc == x) & work).sum()) for x in range(8)])
key = tr[:, 3][work] & np.uint64(0xFFFFFF)
dur = (e - s)[work]
for g in np.unique(key)[:6]:
    print("group", int(g), "waves", int((key == g).sum()), "mean us", dur[key == g].mean())
order = np.argsort(-np.bincount(key.astype(np.int64)))[:5]
for g in order:
    print("big group", int(g), "waves", int((key == g).sum()), "mean us %.1f" % dur[key == g].mean(), "first start %.1f" % s[work][key == g].min(), "last end %.1f" % e[work][key == g].max())
