"""The two arithmetic modes (rt_config.flags, RT_FLAG_ARITH_MAD): the reference builds its kernels with
-cl-mad-enable (main.cpp:253), which lets its compiler fuse multiply-adds; which ones it fuses is the compiler's
business, so no output of the reference pins them.  Here the fused places are stated (include/rt_hip.h), the
oracle and the HIP path agree bit for bit within either mode, and across the modes a frame differs only where
a ray grazes an edge: hit ids equal on at least 99.99 % of camera rays, t within 1e-5 relative, image within
1 LSB on all but a handful of pixels.  Parity unpinned, like the kernels themselves."""
import os

import numpy as np
import pytest

from raytracer_amd import api, layout as L, scene as S

import helpers as H
from test_oracle_pipeline import camera_rays, run_to_completion as oracle_to_completion

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def same_hits_fraction(a, b):
    same = (a["leaf_group"] == b["leaf_group"]) & (a["instance"] == b["instance"]) & (a["tri"] == b["tri"])
    return same.mean(), same


def test_oracle_modes_agree_on_what_is_hit(bunny_scene_64):
    from oracle import pyoracle as O
    rays = camera_rays(bunny_scene_64, 4000, np.random.default_rng(12))
    plain = O.Oracle(bunny_scene_64, warp_width=32)
    fused = O.Oracle(bunny_scene_64, warp_width=32, mad=True)
    a, b = plain.trace_closest(rays), fused.trace_closest(rays)
    plain.close(); fused.close()
    frac, same = same_hits_fraction(a, b)
    assert frac >= 0.999
    hit = same & np.isfinite(a["t"])
    assert hit.sum() > 500
    assert np.allclose(a["t"][hit], b["t"][hit], rtol=1e-5, atol=0)
    assert not H.same_bytes(a["t"][hit], b["t"][hit])          # the modes do round differently


def test_oracle_mad_frame_is_schedule_independent(bunny_scene_64):
    from oracle import pyoracle as O
    limit = 64 * 64
    results = []
    for warp, parked in ((32, 24), (1, 3)):
        o = O.Oracle(bunny_scene_64, warp_width=warp, sample_limit=limit, hit_capacity=limit, mad=True)
        oracle_to_completion(o, steps_parked=parked)
        results.append((o.hits.copy(), o.area.copy()))
        o.close()
    assert H.same_bytes(results[0][0], results[1][0])
    assert np.allclose(results[0][1], results[1][1], rtol=1e-5, atol=1e-6)


def test_oracle_mad_golden_vectors(bunny_scene_128, bunny_scene_64):
    """The fused mode frozen in committed vectors: state after 48 steps, a finished frame with the reference's shading
    and one with two mirror bounces (tests/golden/make_golden.py)."""
    import hashlib
    from oracle import pyoracle as O
    g = np.load(os.path.join(GOLDEN, "pipeline_bunny128_mad.npz"))
    o = O.Oracle(bunny_scene_128, warp_width=int(g["warp_width"]), hit_capacity=len(g["hits"]), mad=True)
    o.init_frame()
    o.step(int(g["steps"]))
    assert H.same_bytes(o.global_data, g["global_data"]) and H.same_bytes(o.ray_index, g["ray_index"])
    assert H.same_bytes(o.grp_data, g["grp_data"]) and H.same_bytes(o.area, g["area"]) and H.same_bytes(o.hits, g["hits"])
    assert H.same_bytes(np.array([o.counters()], L.counters_dtype), g["counters"])
    rl = o.ray_list
    for name, digest in g["ray_field_sha"]:
        assert hashlib.sha256(np.ascontiguousarray(rl[str(name)]).tobytes()).hexdigest() == str(digest), name
    o.close()
    strict = np.load(os.path.join(GOLDEN, "pipeline_bunny128.npz"))
    assert not H.same_bytes(g["hits"][:len(strict["hits"])], strict["hits"])      # the modes do round differently
    for name in ("drain_bunny64_mad.npz", "drain_bunny64_mirror2_mad.npz"):
        g = np.load(os.path.join(GOLDEN, name))
        limit = int(g["sample_limit"])
        o = O.Oracle(bunny_scene_64, warp_width=32, sample_limit=limit, hit_capacity=limit, shading=int(g["shading"]), mad=True)
        assert oracle_to_completion(o, max_steps=1500) == int(g["steps"]), name
        assert H.same_bytes(o.area, g["area"]) and H.same_bytes(o.hits, g["hits"]), name
        assert H.same_bytes(o.resolve_rgba8(), g["image"]), name
        assert H.same_bytes(np.array([o.counters()], L.counters_dtype), g["counters"]), name
        o.close()


@pytest.mark.gpu
def test_gpu_mad_golden_vectors(bunny_scene_128, bunny_scene_64):
    """The HIP path in the fused mode against the committed vectors alone (no oracle in the loop): state after 48
    steps of the reference schedule; finished frames on both schedules."""
    import hashlib
    from test_gpu_pipeline import run_to_completion
    g = np.load(os.path.join(GOLDEN, "pipeline_bunny128_mad.npz"))
    rt = api.RayTracer(bunny_scene_128, warp_width=int(g["warp_width"]), hit_capacity=len(g["hits"]), mad=True)
    rt.init_frame()
    rt.make_step(int(g["steps"]))
    rays, index, groups = rt.dump_state()
    assert H.same_bytes(rt.global_data(), g["global_data"])
    assert H.same_bytes(index, g["ray_index"]) and H.same_bytes(groups, g["grp_data"])
    assert H.same_bytes(rt.hits(), g["hits"])
    assert np.allclose(rt.area(), g["area"], rtol=1e-5, atol=1e-6)
    for name, digest in g["ray_field_sha"]:
        assert hashlib.sha256(np.ascontiguousarray(rays[str(name)]).tobytes()).hexdigest() == str(digest), name
    rt.close()
    for name in ("drain_bunny64_mad.npz", "drain_bunny64_mirror2_mad.npz"):
        g = np.load(os.path.join(GOLDEN, name))
        limit = int(g["sample_limit"])
        for free in (True, False):
            rt = api.RayTracer(bunny_scene_64, warp_width=64, sample_limit=limit, hit_capacity=limit, free_schedule=free,
                               shading=int(g["shading"]), mad=True)
            run_to_completion(rt, steps_parked=12, max_steps=1500)
            assert H.same_bytes(rt.hits(), g["hits"]), (name, free)
            assert np.allclose(rt.area(), g["area"], rtol=1e-5, atol=1e-6), (name, free)
            assert np.abs(rt.draw_frame().astype(int) - g["image"].astype(int)).max() <= 1, (name, free)
            assert int(rt.counters()["shadow_spawned"]) == int(g["counters"][0]["shadow_spawned"]), (name, free)
            rt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("warp", [32, 4])
def test_gpu_mad_state_matches_oracle_step_by_step(bunny_scene_128, warp):
    from oracle import pyoracle as O
    from test_gpu_pipeline import assert_same_state
    cap = 16384 * 8
    rt = api.RayTracer(bunny_scene_128, warp_width=warp, hit_capacity=cap, mad=True)
    o = O.Oracle(bunny_scene_128, warp_width=warp, hit_capacity=cap, mad=True)
    rt.init_frame(); o.init_frame()
    assert_same_state(rt, o, -1)
    for step in range(48):
        rt.make_step(1); o.step(1)
        assert_same_state(rt, o, step)
    assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6)
    assert H.same_bytes(rt.hits(), o.hits)
    c, oc = rt.counters(), o.counters()
    for k in ("primary_spawned", "shadow_spawned", "steps", "box_tests", "tri_tests"):
        assert int(c[k]) == int(oc[k]), k
    rt.close(); o.close()


@pytest.mark.gpu
def test_gpu_mad_full_model_state_and_bounces(standin_full_ply):
    """The full-size scene (9,728 groups) with two diffuse bounces, fused arithmetic: state step by step."""
    from oracle import pyoracle as O
    from test_gpu_pipeline import assert_same_state
    sc = S.build_scene([S.BUNNY, standin_full_ply], 256, 144, 40960)
    flags = api.shading_flags(api.SHADE_DIFFUSE, 2)
    cap = 256 * 144 * 2
    rt = api.RayTracer(sc, warp_width=8, hit_capacity=cap, shading=flags, mad=True)
    o = O.Oracle(sc, warp_width=8, hit_capacity=cap, shading=flags, mad=True)
    rt.init_frame(); o.init_frame()
    for step in range(56):
        rt.make_step(1); o.step(1)
        assert_same_state(rt, o, step)
    assert (o.ray_list["type"] >> 8).max() == 2
    assert H.same_bytes(rt.hits(), o.hits)
    assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6)
    c, oc = rt.counters(), o.counters()
    for k in ("primary_spawned", "shadow_spawned", "box_tests", "tri_tests"):
        assert int(c[k]) == int(oc[k]), k
    rt.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shading", [0, api.shading_flags(api.SHADE_MIRROR, 2)])
def test_gpu_mad_finished_frames_on_both_schedules(bunny_scene_64, shading):
    from oracle import pyoracle as O
    from test_gpu_pipeline import run_to_completion
    limit = 64 * 64
    o = O.Oracle(bunny_scene_64, warp_width=32, sample_limit=limit, hit_capacity=limit, shading=shading, mad=True)
    oracle_to_completion(o, max_steps=1500)
    for free in (True, False):
        rt = api.RayTracer(bunny_scene_64, warp_width=64, sample_limit=limit, hit_capacity=limit, free_schedule=free,
                           shading=shading, mad=True)
        run_to_completion(rt, steps_parked=12, max_steps=1500)
        assert H.same_bytes(rt.hits(), o.hits), free
        assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6), free
        assert np.abs(rt.draw_frame().astype(int) - o.resolve_rgba8().astype(int)).max() <= 1, free
        assert int(rt.counters()["shadow_spawned"]) == int(o.counters()["shadow_spawned"]), free
        rt.close()
    o.close()


@pytest.mark.gpu
def test_gpu_modes_at_the_headline_size(standin_full_ply):
    """1920x1080, full pool, free schedule, one sample per pixel in either mode.  Within the fused mode 60,000 hit
    records equal the oracle's bit for bit; across the modes at least 99.99 % of the pixels hit the same surface point
    (same ids, or -- on an edge two triangles share -- ids that differ at a t within 1e-5 relative), t within 1e-5
    relative on 99.99 % of the pixels whose ids agree (grazing hits excepted), and the images differ by more than 1 LSB on fewer than 1 pixel in 10,000
    (PSNR above 50 dB)."""
    from oracle import pyoracle as O
    from test_gpu_fullsize import run_until_quiet
    W, Hh = 1920, 1080
    sc = S.build_scene([S.BUNNY, standin_full_ply], W, Hh, W * Hh)
    frames = {}
    for mad in (False, True):
        rt = api.RayTracer(sc, warp_width=64, sample_limit=W * Hh, hit_capacity=W * Hh, free_schedule=True, mad=mad)
        stuck = run_until_quiet(rt, steps_parked=16)
        frames[mad] = (rt.hits(), rt.draw_frame(), stuck)
        rt.close()
    hits, image, stuck = frames[True]
    done = np.ones(W * Hh, bool)
    done[stuck] = False
    done[frames[False][2]] = False
    pixels = np.random.default_rng(355).choice(W * Hh, 60000, replace=False)
    pixels = np.sort(pixels[done[pixels]]).astype(np.uint32)
    o = O.Oracle(sc, warp_width=32, threads=8, mad=True)
    want = o.trace_closest(o.camera_rays(pixels))
    o.close()
    assert H.same_bytes(hits[pixels], want)
    a, b = hits[done], frames[False][0][done]
    frac, same = same_hits_fraction(a, b)
    both_hit = np.isfinite(a["t"]) & np.isfinite(b["t"])
    close = both_hit & (np.abs(a["t"] - b["t"]) <= 1e-5 * np.abs(b["t"]))
    # a ray through an edge two triangles share, or through the seam of two overlapping instances, may take either
    # of them in either mode; what counts as a different result is another surface point
    elsewhere = ~same & ~close
    print("hit ids equal across the modes:", frac, "| different ids:", int((~same).sum()), "of them at the same t:",
          int((~same & close).sum()), "| hit in one mode only:", int((~same & ~both_hit).sum()))
    assert frac >= 0.9995 and elsewhere.mean() <= 1e-4
    # t itself: within 1e-5 relative except where the ray meets its triangle at a grazing angle (t is a quotient whose
    # divisor, dot(dir, n), then cancels); those are few and stay within 1e-3
    rel = np.abs(a["t"] - b["t"])[same & both_hit] / np.abs(b["t"][same & both_hit])
    print("same triangle, t differs by more than 1e-5 relative:", int((rel > 1e-5).sum()), "of", len(rel), "| largest:", float(rel.max()))
    assert (rel > 1e-5).mean() <= 1e-4 and rel.max() < 1e-3
    diff = np.abs(image.astype(int) - frames[False][1].astype(int)).reshape(-1, 4)[done]
    assert (diff.max(axis=1) > 1).mean() < 1e-4
    mse = (diff[:, :3].astype(np.float64) ** 2).mean()
    assert 10 * np.log10(255.0 ** 2 / max(mse, 1e-12)) > 50


def test_fused_mode_does_not_handle_axis_parallel_rays(bunny_scene_64):
    """A limit of the fused slab test (include/rt_hip.h, tests/test_culling_property.py): a ray with a direction
    component that is exactly zero.  Forty rays along +y through the centres of instance boxes (a camera with dx = dy = 0
    and top_left = (0, 1, 0) makes every sample such a ray): in the strict mode all finish and hit a model; in the fused
    mode the test stops seeing the two axes with an infinite reciprocal, boxes beside the ray are entered, queues
    overflow and most of the rays are still in flight after 600 steps."""
    import copy
    from oracle import pyoracle as O
    boxes = bunny_scene_64.aabbs[:256]
    finished = {False: 0, True: 0}
    hit = {False: 0, True: 0}
    for k in range(40):
        centre = 0.5 * (boxes["min"][6 * k] + boxes["max"][6 * k])
        sc = copy.deepcopy(bunny_scene_64)
        cam = sc.data["cam"][0]
        cam["eye"][:3] = (centre[0], -1.0, centre[2])
        cam["top_left"][:3] = (0, 1, 0)
        cam["dx"][:] = 0
        cam["dy"][:] = 0
        for mad in (False, True):
            o = O.Oracle(sc, warp_width=1, sample_limit=64 * 64, hit_capacity=1, mad=mad)
            o.init_frame_samples(np.array([0], np.uint32))
            assert tuple(o.ray_list[0]["dir_max"][:3]) == (0.0, 1.0, 0.0)
            o.step(600)
            finished[mad] += int(o.global_data["ray_count"][0]) == 0
            hit[mad] += bool(np.isfinite(o.hits["t"][0]))
            o.close()
    assert finished[False] == 40 and hit[False] == 40
    assert finished[True] < 20
