"""Developer tool: per-phase cycle shares of the box-group kernel from a -DRT_STAMPS build."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WHICH = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["RT_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "csrc", f"librt_hip_stamps{WHICH}.so")
from raytracer_amd import api, scene as S
dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon):
    S.write_standin_ply(dragon)
sc = S.build_scene([S.BUNNY, dragon], 1920, 1080, 1920 * 1080)
FREE = os.environ.get("RT_SCHEDULE", "free") == "free"
rt = api.RayTracer(sc, warp_width=64, free_schedule=FREE)
print("schedule:", "free" if FREE else "reference")
rt.init_frame(); rt.make_step(96); rt.sync()
rt.debug_stamps(reset=True)
rt.make_step(32); rt.sync()
st = rt.debug_stamps()
names = {"1": ["load+transform", "box scan", "hit network", "merge+store"],
         "3": ["box visits (waves, M)", "in the one-octant scan", "scene-root visits", "scene root in the one-octant scan"],
         "4": ["merge: ids + pending landed (after issue)", "merge: ranks + stores issued", "merge: stores acknowledged", "whole box visit"],
         "2": ["leaf: load", "leaf: scan", "leaf: whole visit", "all cases (incl. leaf)"]}[WHICH[-1]]
for n, v in zip(names, st):
    print(f"{n:40s} {v/32/1e6:10.2f} M per step ({v/32:.0f})")
