"""Developer tool: how many cluster/box tests the box kernel executes against what a full scan would
(needs `make variant NAME=scan DEFS=-DRT_COUNT_SCANNED`; run with RT_HIP_LIB pointing at that build)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scene as S

dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon):
    S.write_standin_ply(dragon, 934, 466)
sc = S.build_scene([S.BUNNY, dragon], 1920, 1080, 1920 * 1080)
for free in (True, False):
    rt = api.RayTracer(sc, warp_width=64, free_schedule=free)
    rt.init_frame()
    rt.make_step(96)
    c0, s0 = rt.counters(), rt.debug_stamps()
    rt.make_step(32)
    c1, s1 = rt.counters(), rt.debug_stamps()
    full = int(c1["box_tests"]) - int(c0["box_tests"])
    done = s1[0] - s0[0]
    print(f"free={free}: full scan {full / 32e6:.1f} M box tests per step, executed {done / 32e6:.1f} M ({done / full:.3f})")
    rt.close()
