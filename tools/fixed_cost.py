"""Developer tool: what a timed region costs beside its steps (launch, frame resolve, synchronisation)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scene as S
dragon = os.path.join(tempfile.gettempdir(), "rt_dragon_standin_tool.ply")
if not os.path.exists(dragon):
    S.write_standin_ply(dragon, 934, 466)
sc = S.build_scene([S.BUNNY, dragon], 1920, 1080, 1920 * 1080)
rt = api.RayTracer(sc, warp_width=64, free_schedule=True)
rt.init_frame(); rt.make_step(300); rt.sync()
def timed(f, n=5):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for k in (1, 2, 5, 20, 64, 200):
    ms = timed(lambda: (rt.make_step(k), rt.sync()))
    print(f"{k:4d} steps + sync: {ms:8.3f} ms  = {ms / k:.4f} per step")
print("sync of an idle context: %.4f ms" % timed(rt.sync, 20))
print("resolve on the device (kernel + sync): %.4f ms" % timed(rt.draw_frame_device, 10))
print("20 steps + resolve + sync: %.3f ms" % timed(lambda: (rt.make_step(20), rt.draw_frame_device(), rt.sync())))
print("enqueue of 20 steps alone (no sync): %.4f ms" % timed(lambda: rt.make_step(20), 1)); rt.sync()
