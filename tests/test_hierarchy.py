"""A narrower hierarchy over the same leaves (an option of the host scene build: inner groups of 8-15 boxes instead
of the reference's 128-255, model.cpp:42-46 with aabb_threshold = 8; the leaves and their triangles are the
reference's).  It is just another scene to the device path and to the oracle, so the two stay bit-identical; against
the reference's hierarchy every camera ray hits the same triangle at the same t -- the leaf's group id differs, the
triangle's corners and t do not -- while the image may differ where the reference's shadow quirk (a shadowed sample
darkens once per occluding leaf visited, ray-tracer.cl:174-177) sees another order of leaves."""
import numpy as np
import pytest

from raytracer_amd import api, layout as L, scene as S

import helpers as H
from test_oracle_pipeline import camera_rays, run_to_completion as oracle_to_completion


def corners_of(sc, hits):
    """World-independent identity of what was hit: instance and the three model-space corners of the triangle."""
    hit = np.isfinite(hits["t"])
    out = np.full((len(hits), 9), np.nan, np.float32)
    grp = sc.groups["w"][hits["leaf_group"][hit] & L.GROUP_ID_MASK]          # vtx_offs, tri_offs, tri_count, material
    word = sc.triangles[grp[:, 1] + hits["tri"][hit]]
    for k in range(3):
        out[hit, 3 * k:3 * k + 3] = sc.vertices["pos"][grp[:, 0] + ((word >> (10 * k)) & 0x3FF), :3]
    return out


def same_surface(sc_a, a, sc_b, b):
    return (H.same_bytes(a["t"], b["t"]) and np.array_equal(a["instance"], b["instance"])
            and np.array_equal(corners_of(sc_a, a), corners_of(sc_b, b), equal_nan=True))


def test_oracle_hits_do_not_depend_on_the_hierarchy():
    from oracle import pyoracle as O
    wide = S.build_scene([S.BUNNY, S.BUNNY], 64, 64, 8192)
    narrow = S.build_scene([S.BUNNY, S.BUNNY], 64, 64, 8192, aabb_threshold=8)
    assert len(narrow.groups) > len(wide.groups) and len(narrow.triangles) == len(wide.triangles)
    rays = camera_rays(wide, 3000, np.random.default_rng(8))
    a = O.Oracle(wide, warp_width=32)
    b = O.Oracle(narrow, warp_width=32)
    ha, hb = a.trace_closest(rays), b.trace_closest(rays)
    assert same_surface(wide, ha, narrow, hb)
    assert np.isfinite(ha["t"]).sum() > 300
    wa, wb = a.work(), b.work()
    root = 256 * len(rays)                                   # every ray scans the scene root's 256 instance boxes once
    assert int(wb["box_tests"]) - root < 0.3 * (int(wa["box_tests"]) - root) and int(wb["tri_tests"]) <= int(wa["tri_tests"])
    a.close(); b.close()


@pytest.mark.gpu
def test_gpu_narrow_hierarchy_state_matches_oracle():
    from oracle import pyoracle as O
    from test_gpu_pipeline import assert_same_state
    sc = S.build_scene([S.BUNNY, S.BUNNY], 128, 128, 16384, aabb_threshold=8)
    rt = api.RayTracer(sc, warp_width=4, hit_capacity=16384 * 4)
    o = O.Oracle(sc, warp_width=4, hit_capacity=16384 * 4)
    rt.init_frame(); o.init_frame()
    for step in range(64):
        rt.make_step(1); o.step(1)
        assert_same_state(rt, o, step)
    assert H.same_bytes(rt.hits(), o.hits)
    c, oc = rt.counters(), o.counters()
    for k in ("primary_spawned", "shadow_spawned", "box_tests", "tri_tests"):
        assert int(c[k]) == int(oc[k]), k
    assert int(c["shadow_spawned"]) > 1000
    rt.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mad", [False, True])
def test_gpu_narrow_hierarchy_hits_the_same_triangles(standin_full_ply, mad):
    """Full-size scene, 1920x1080, free schedule, one sample per pixel under the reference's hierarchy and under the
    narrow one: same t, same instance, same triangle for every pixel; inside the models the narrow one needs under a
    third of the box tests (as the reference's loops count them; the scene root's list of 256 instances is the same in
    both); the images agree to a PSNR above 40 dB."""
    from test_gpu_fullsize import run_until_quiet
    W, Hh = 1920, 1080
    files = [S.BUNNY, standin_full_ply]
    out = {}
    for name, threshold in (("wide", 128), ("narrow", 8)):
        sc = S.build_scene(files, W, Hh, W * Hh, aabb_threshold=threshold)
        rt = api.RayTracer(sc, warp_width=64, sample_limit=W * Hh, hit_capacity=W * Hh, free_schedule=True, mad=mad)
        stuck = run_until_quiet(rt, steps_parked=16)
        out[name] = (sc, rt.hits(), rt.draw_frame(), stuck, rt.counters())
        rt.close()
    done = np.ones(W * Hh, bool)
    done[out["wide"][3]] = False
    done[out["narrow"][3]] = False
    assert (~done).sum() <= 8
    assert same_surface(out["wide"][0], out["wide"][1][done], out["narrow"][0], out["narrow"][1][done])
    inside = lambda c: int(c["box_tests"]) - 256 * (int(c["primary_spawned"]) + int(c["shadow_spawned"]))   # without the scene root's scans
    ratio = inside(out["narrow"][4]) / inside(out["wide"][4])
    print("box tests inside the models, narrow / reference hierarchy:", ratio,
          "| all box tests:", int(out["narrow"][4]["box_tests"]) / int(out["wide"][4]["box_tests"]))
    assert ratio < 0.34
    diff = (out["wide"][2].astype(np.float64) - out["narrow"][2].astype(np.float64))[..., :3].reshape(-1, 3)[done]
    psnr = 10 * np.log10(255.0 ** 2 / max((diff ** 2).mean(), 1e-12))
    print("PSNR between the hierarchies:", psnr)
    assert psnr > 40
