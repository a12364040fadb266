"""The headline scene at its full model size (bunny + the 870,488-triangle dragon stand-in, 9,728
groups) under checking tests: step-by-step state and a finished frame against the oracle at a small
image, closest hits of camera rays at 1920x1080 with the full pool, and the 3840x2160 configuration
with two of its eight row bands against the full frame."""
import numpy as np
import pytest

from raytracer_amd import api, layout as L, scene as S, tiling

import helpers as H
from test_gpu_pipeline import assert_same_state, run_to_completion
from test_oracle_pipeline import camera_rays, run_to_completion as oracle_to_completion

pytestmark = pytest.mark.gpu


def test_full_model_state_matches_oracle(standin_full_ply):
    """256x144 image, 40,960 slots: the oracle's cost follows the pool, not the model."""
    from oracle import pyoracle as O
    sc = S.build_scene([S.BUNNY, standin_full_ply], 256, 144, 40960)
    assert sc.group_count == 9728 and len(sc.triangles) > 900000
    cap = 256 * 144 * 2
    rt = api.RayTracer(sc, warp_width=32, hit_capacity=cap)
    o = O.Oracle(sc, warp_width=32, hit_capacity=cap)   # one thread: with more, the oracle's accumulator has the reference's race
    rt.init_frame(); o.init_frame()
    for step in range(48):
        rt.make_step(1); o.step(1)
        assert_same_state(rt, o, step)
    assert H.same_bytes(rt.hits(), o.hits)
    assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6)
    c, oc = rt.counters(), o.counters()
    for k in ("primary_spawned", "shadow_spawned", "box_tests", "tri_tests"):
        assert int(c[k]) == int(oc[k]), k
    rt.close(); o.close()


def test_full_model_finished_frame_matches_oracle(standin_full_ply):
    """One sample per pixel run to completion at the production granularity: hit records exact
    (t, leaf, instance, triangle), image within 1 LSB of the oracle's (which runs WARP_WIDTH 32)."""
    from oracle import pyoracle as O
    sc = S.build_scene([S.BUNNY, standin_full_ply], 256, 144, 40960)
    limit = 256 * 144
    rt = api.RayTracer(sc, warp_width=64, sample_limit=limit, hit_capacity=limit)
    o = O.Oracle(sc, warp_width=32, sample_limit=limit, hit_capacity=limit)
    run_to_completion(rt, steps_parked=10)
    oracle_to_completion(o, steps_parked=10)
    assert H.same_bytes(rt.hits(), o.hits)
    assert np.isfinite(rt.hits()["t"]).mean() > 0.2
    assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6)
    assert np.abs(rt.draw_frame().astype(int) - o.resolve_rgba8().astype(int)).max() <= 1
    rt.close()
    # the free schedule (what the benchmark runs) gives every sample the same result
    rt = api.RayTracer(sc, warp_width=64, sample_limit=limit, hit_capacity=limit, free_schedule=True)
    run_to_completion(rt, steps_parked=10)
    assert H.same_bytes(rt.hits(), o.hits)
    assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6)
    assert np.abs(rt.draw_frame().astype(int) - o.resolve_rgba8().astype(int)).max() <= 1
    c, oc = rt.counters(), o.counters()
    for k in ("primary_spawned", "shadow_spawned"):
        assert int(c[k]) == int(oc[k]), k
    rt.close(); o.close()


def check_step_invariants(rt, pool, warp, chunks=3, steps=16):
    prev = pool
    for chunk in range(chunks):
        rt.make_step(steps)
        g = rt.global_data()[0]
        n = int(g["ray_count"])
        assert n <= pool and n % warp == 0
        _, index, groups = rt.dump_state(rays=False)
        ids = index["key"][:n] & L.GROUP_ID_MASK
        assert np.all(np.diff(ids.astype(np.int64)) >= 0)
        assert np.all(ids.reshape(-1, warp).min(axis=1) == ids.reshape(-1, warp).max(axis=1))
        assert np.array_equal(np.sort(index["slot"]), np.arange(pool, dtype=np.uint32))
        assert int(groups["count"][:, 0].sum()) == n and n + int(groups["count"][:, 1].sum()) == pool
        assert int(g["pixel_offset"]) >= prev
        prev = int(g["pixel_offset"])
    return prev


def run_until_quiet(rt, steps_parked, patience=160, max_steps=3000):
    """Runs one frame until no ray is in flight, or until the same few rays have been in flight for
    `patience` steps.  The reference's automaton has a livelock: when eight or more pending nodes share
    one entry parameter, the overflow restart (ray-tracer.cl:228-231) re-creates the same queue
    forever; such a slot never finishes in the reference either.  Returns the pixels of the rays
    left in flight (2 of 8.3 million samples at 3840x2160 on this scene)."""
    rt.init_frame()
    rt.make_step(steps_parked)
    rt.set_warp_width(1)
    steps, quiet, last = steps_parked, 0, -1
    while steps < max_steps:
        rt.make_step(8)
        steps += 8
        n = int(rt.global_data()["ray_count"][0])
        if n == 0:
            return np.zeros(0, np.int64)
        quiet = quiet + 8 if n == last else 0
        last = n
        if n <= 16 and quiet >= patience:
            rays, index, _ = rt.dump_state(groups=False)
            return np.unique(rays["pixel"][index["slot"][:n]].astype(np.int64))
    raise AssertionError(f"{last} rays still in flight after {steps} steps")


def test_headline_config_full_model(standin_full_ply):
    """1920x1080, pool 2,080,768, G = 9,728: the invariants of the reference's debug hooks
    (main.cpp:150, 153-168) over 48 steps, then 60,000 camera rays traced to their closest hit,
    bit-exact against the oracle."""
    from oracle import pyoracle as O
    sc = S.build_scene([S.BUNNY, standin_full_ply], 1920, 1080, 1920 * 1080)
    assert sc.group_count == 9728
    for free in (True, False):
        rt = api.RayTracer(sc, warp_width=64, free_schedule=free)
        rt.init_frame()
        assert rt.pool == 2080768
        check_step_invariants(rt, rt.pool, 64)
        if free:
            rt.close()
    area = rt.area()
    assert np.isfinite(area).all() and area[:, 3].max() > 0
    rays = camera_rays(sc, 60000, np.random.default_rng(1080))
    got = rt.trace_closest(rays)
    o = O.Oracle(sc, warp_width=32, threads=8)
    want = o.trace_closest(rays)
    assert H.same_bytes(got, want)
    assert 0.2 < np.isfinite(got["t"]).mean() < 1.0
    rt.close(); o.close()


def test_4k_config_and_two_of_eight_bands(standin_full_ply):
    """3840x2160, pool 8,298,496: step invariants on the full frame; then one sample per pixel of the
    full frame and of bands 0 and 5 of 8 (270 rows each, as ranks 0 and 5 of an 8-GPU run render
    them): every band sample's hit record equals the full frame's for the same pixel, and the band
    images equal the full image's rows."""
    W, Hh = 3840, 2160
    pool = L.align_up(W * Hh, L.SORT_TILE)
    assert pool == 8298496
    full_sc = S.build_scene([S.BUNNY, standin_full_ply], W, Hh, W * Hh)
    full = api.RayTracer(full_sc, warp_width=64, sample_limit=W * Hh, hit_capacity=W * Hh)
    full.init_frame()
    check_step_invariants(full, pool, 64, chunks=2, steps=12)
    stuck = run_until_quiet(full, steps_parked=12)
    hits = full.hits()
    image = full.draw_frame()
    area = full.area()
    full.close()
    assert len(stuck) <= 16
    done = np.ones(W * Hh, bool)
    done[stuck] = False                             # pixels of livelocked rays keep changing
    assert np.all(area[done, 3] >= 1 - 1e-5)        # every other pixel got its sample
    for rank in (0, 5):
        first, rows = tiling.band_of(rank, 8, Hh)
        assert rows == 270
        sc = S.build_scene([S.BUNNY, standin_full_ply], W, Hh, W * rows, band_first=first, band_rows=rows)
        rt = api.RayTracer(sc, warp_width=64, sample_limit=W * rows, hit_capacity=W * rows)
        band_stuck = run_until_quiet(rt, steps_parked=12)
        ok = done[first * W:(first + rows) * W]
        assert np.array_equal(band_stuck + first * W, stuck[(stuck >= first * W) & (stuck < (first + rows) * W)]), rank
        assert H.same_bytes(rt.hits(), hits[first * W:(first + rows) * W]), rank   # primary hits precede the livelock
        diff = np.abs(rt.draw_frame().astype(int) - image[first:first + rows].astype(int)).reshape(-1, 4)
        assert diff[ok].max() <= 1, rank
        assert np.allclose(rt.area()[ok], area[first * W:(first + rows) * W][ok], rtol=1e-5, atol=1e-6), rank
        rt.close()
