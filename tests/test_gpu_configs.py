"""The schedule the benchmark runs (RT_FLAG_FREE_SCHEDULE) and the two configurations beyond the reference's own
scene, under checking tests at size: the 1920x1080 frame with the full pool against the oracle's closest hits,
3840x2160 with diffuse bounces (a band of it finished against the oracle, the full pool through the step
invariants), a 5x5x5 grid of the full-size second model with diffuse bounces from the hit points, and the samples
that never finish at 3840x2160 followed one by one through the oracle."""
import numpy as np
import pytest

from raytracer_amd import api, layout as L, scene as S, tiling

import helpers as H
from test_gpu_fullsize import check_step_invariants, run_until_quiet
from test_gpu_pipeline import run_to_completion
from test_oracle_pipeline import run_to_completion as oracle_to_completion

pytestmark = pytest.mark.gpu

DIFFUSE2 = api.shading_flags(api.SHADE_DIFFUSE, 2)


def periodic_from(states, max_period):
    """Smallest p <= max_period with states[i] == states[i + p] for every i of the recorded window, or 0."""
    n = len(states)
    for p in range(1, max_period + 1):
        if n - p >= 2 * p and all(states[i] == states[i + p] for i in range(n - p)):
            return p
    return 0


def oracle_never_finishing(files, W, Hh, suspects, extra=2000, seed=7):
    """Puts the samples `suspects` (first sample of those pixels of the W x Hh frame), their neighbours and `extra`
    random other samples through the oracle, each in a slot of its own, full shading included.  Returns the
    samples still in flight after 1,200 steps and the period with which the state of their slots recurs over the
    next 400 steps (0: no recurrence seen)."""
    from oracle import pyoracle as O
    suspects = np.asarray(suspects, np.int64)
    near = np.concatenate([suspects + d for d in (-W - 1, -W, -W + 1, -2, -1, 1, 2, W - 1, W, W + 1)]) if len(suspects) else suspects
    near = near[(near >= 0) & (near < W * Hh)]
    others = np.random.default_rng(seed).choice(W * Hh, extra, replace=False)
    samples = np.unique(np.concatenate([suspects, near, others])).astype(np.uint32)
    sc = S.build_scene(files, W, Hh, 8192)
    o = O.Oracle(sc, warp_width=1, sample_limit=W * Hh)
    o.init_frame_samples(samples)                                 # slot i runs samples[i]
    o.step(1200)
    n = int(o.global_data["ray_count"][0])
    live_slots = np.sort(o.ray_index["slot"][:n])
    states = []
    for _ in range(400 if n else 0):
        o.step(1)
        assert int(o.global_data["ray_count"][0]) == n
        states.append(o.ray_list[live_slots].tobytes())
    o.close()
    return samples[live_slots].astype(np.int64), periodic_from(states, 100)


def test_headline_frame_on_the_free_schedule_matches_oracle_hits(standin_full_ply):
    """1920x1080, pool 2,080,768, one sample per pixel on the free schedule, run until no ray is in flight but those
    that never finish (the reference's automaton has a livelock, see test_samples_that_never_finish...): the hit
    records of 60,000 pixels equal, bit for bit, what the oracle finds for those pixels' camera rays; so do the same
    rays traced through the free context's rt_trace_closest; and the samples left in flight are exactly those that
    never finish in the oracle either."""
    from oracle import pyoracle as O
    W, Hh = 1920, 1080
    files = [S.BUNNY, standin_full_ply]
    sc = S.build_scene(files, W, Hh, W * Hh)
    rt = api.RayTracer(sc, warp_width=64, sample_limit=W * Hh, hit_capacity=W * Hh, free_schedule=True)
    assert rt.pool == 2080768
    stuck = run_until_quiet(rt, steps_parked=16)
    print("samples left in flight at 1920x1080:", stuck.tolist())
    assert len(stuck) <= 4
    hits = rt.hits()
    area = rt.area()
    done = np.ones(W * Hh, bool)
    done[stuck] = False
    assert np.all(area[done, 3] >= 1 - 1e-5)                  # every other pixel got its sample
    c = rt.counters()
    assert int(c["primary_spawned"]) == W * Hh
    pixels = np.random.default_rng(1080).choice(W * Hh, 60000, replace=False)
    pixels = np.sort(pixels[done[pixels]]).astype(np.uint32)
    o = O.Oracle(sc, warp_width=32, threads=8)
    rays = o.camera_rays(pixels)                              # sample id = pixel for the first sample of every pixel
    want = o.trace_closest(rays)
    o.close()
    assert 0.2 < np.isfinite(want["t"]).mean() < 1.0
    assert H.same_bytes(hits[pixels], want)
    assert H.same_bytes(rt.trace_closest(rays), want)
    rt.close()
    in_oracle, period = oracle_never_finishing(files, W, Hh, stuck)
    assert np.array_equal(in_oracle, stuck)
    assert len(stuck) == 0 or period > 0


def test_4k_free_schedule_frame_and_bands(standin_full_ply):
    """3840x2160, pool 8,298,496, free schedule: step invariants, then one sample per pixel of the full frame and of
    bands 0 and 5 of 8; a band's hit records and image rows are the full frame's, and 40,000 of the frame's hit
    records equal the oracle's closest hits."""
    from oracle import pyoracle as O
    W, Hh = 3840, 2160
    files = [S.BUNNY, standin_full_ply]
    full_sc = S.build_scene(files, W, Hh, W * Hh)
    full = api.RayTracer(full_sc, warp_width=64, sample_limit=W * Hh, hit_capacity=W * Hh, free_schedule=True)
    assert full.pool == 8298496
    full.init_frame()
    check_step_invariants(full, full.pool, 64, chunks=2, steps=12)
    stuck = run_until_quiet(full, steps_parked=12)
    hits, image, area = full.hits(), full.draw_frame(), full.area()
    full.close()
    assert len(stuck) <= 16
    done = np.ones(W * Hh, bool)
    done[stuck] = False
    assert np.all(area[done, 3] >= 1 - 1e-5)
    pixels = np.sort(np.random.default_rng(2160).choice(W * Hh, 40000, replace=False)).astype(np.uint32)
    o = O.Oracle(full_sc, warp_width=32, threads=8)
    want = o.trace_closest(o.camera_rays(pixels))
    o.close()
    assert H.same_bytes(hits[pixels], want)
    for rank in (0, 5):
        first, rows = tiling.band_of(rank, 8, Hh)
        sc = S.build_scene(files, W, Hh, W * rows, band_first=first, band_rows=rows)
        rt = api.RayTracer(sc, warp_width=64, sample_limit=W * rows, hit_capacity=W * rows, free_schedule=True)
        band_stuck = run_until_quiet(rt, steps_parked=12)
        ok = done[first * W:(first + rows) * W]
        assert np.array_equal(band_stuck + first * W, stuck[(stuck >= first * W) & (stuck < (first + rows) * W)]), rank
        assert H.same_bytes(rt.hits(), hits[first * W:(first + rows) * W]), rank
        diff = np.abs(rt.draw_frame().astype(int) - image[first:first + rows].astype(int)).reshape(-1, 4)
        assert diff[ok].max() <= 1, rank
        assert np.allclose(rt.area()[ok], area[first * W:(first + rows) * W][ok], rtol=1e-5, atol=1e-6), rank
        rt.close()


def test_samples_that_never_finish_do_so_in_the_oracle(standin_full_ply):
    """The reference's automaton has a livelock: when eight or more pending nodes share one entry parameter, the
    overflow restart (ray-tracer.cl:228-231) re-creates the same queue for ever.  The samples the HIP path leaves in
    flight at 3840x2160 (free schedule, the frame's eight bands one after the other) are put through the oracle one
    by one, together with their neighbours and 2,000 others: exactly those samples stay in flight there too, and
    the state of their slots recurs."""
    W, Hh = 3840, 2160
    files = [S.BUNNY, standin_full_ply]
    found = []
    for rank in range(8):
        first, rows = tiling.band_of(rank, 8, Hh)
        sc = S.build_scene(files, W, Hh, W * rows, band_first=first, band_rows=rows)
        rt = api.RayTracer(sc, warp_width=64, sample_limit=W * rows, hit_capacity=W * rows, free_schedule=True)
        band_stuck = run_until_quiet(rt, steps_parked=12)
        rt.close()
        found.extend((band_stuck + first * W).tolist())
    stuck = np.array(sorted(found), np.int64)
    print("samples left in flight at 3840x2160:", stuck.tolist())
    assert 1 <= len(stuck) <= 16
    in_oracle, period = oracle_never_finishing(files, W, Hh, stuck)
    print("period of the recurring state:", period)
    assert np.array_equal(in_oracle, stuck)
    assert period > 0


def test_4k_band_with_diffuse_bounces_matches_oracle(standin_full_ply):
    """Configuration "3840x2160, multi-bounce, tile-sharded": eight rows of the frame (as a rank's band, row offset
    1076) with two diffuse bounces per sample, finished on both schedules, against the oracle: hit records,
    accumulator, image, ray counts."""
    from oracle import pyoracle as O
    W, Hh, first, rows = 3840, 2160, 1076, 8
    limit = W * rows
    sc = S.build_scene([S.BUNNY, standin_full_ply], W, Hh, 32768, band_first=first, band_rows=rows)
    assert sc.row_offset == first and sc.height == rows
    o = O.Oracle(sc, warp_width=32, sample_limit=limit, hit_capacity=limit, shading=DIFFUSE2)
    oracle_to_completion(o, steps_parked=24, max_steps=4000)
    oc = o.counters()
    assert int(oc["shadow_spawned"]) > 1.2 * int(np.isfinite(o.hits["t"]).sum()) > 5000   # bounces and shadow rays
    for free in (True, False):
        rt = api.RayTracer(sc, warp_width=64, sample_limit=limit, hit_capacity=limit, free_schedule=free, shading=DIFFUSE2)
        run_to_completion(rt, steps_parked=12, max_steps=4000)
        assert H.same_bytes(rt.hits(), o.hits), free
        assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6), free
        assert np.abs(rt.draw_frame().astype(int) - o.resolve_rgba8().astype(int)).max() <= 1, free
        c = rt.counters()
        for k in ("primary_spawned", "shadow_spawned"):
            assert int(c[k]) == int(oc[k]), (k, free)
        rt.close()
    o.close()


def test_4k_full_pool_with_diffuse_bounces_keeps_the_invariants(standin_full_ply):
    W, Hh = 3840, 2160
    sc = S.build_scene([S.BUNNY, standin_full_ply], W, Hh, W * Hh)
    for free in (True, False):
        rt = api.RayTracer(sc, warp_width=64, free_schedule=free, shading=DIFFUSE2)
        rt.init_frame()
        check_step_invariants(rt, rt.pool, 64, chunks=3, steps=16)
        rays, _, _ = rt.dump_state(index=False, groups=False)
        assert (rays["type"] >> 8).max() == 2 and np.all((rays["type"] & 0xFF) <= 1)
        c = rt.counters()
        assert int(c["shadow_spawned"]) > 0
        area = rt.area()
        assert np.isfinite(area).all() and area[:, 3].max() > 0
        rt.close()


def test_incoherent_grid_config_matches_oracle(standin_full_ply):
    """The incoherent workload: a 5x5x5 grid of the full-size second model (125 instances, 109 million instanced
    triangles), two diffuse bounces from the hit points.  A 192x108 frame finished on both schedules against the
    oracle; then the full pool at 1920x1080 through the step invariants."""
    from oracle import pyoracle as O
    files = [standin_full_ply, standin_full_ply]
    W, Hh = 192, 108
    limit = W * Hh
    sc = S.build_scene(files, W, Hh, 24576, grid=5)
    assert sc.resident_triangles > 100_000_000
    o = O.Oracle(sc, warp_width=32, sample_limit=limit, hit_capacity=limit, shading=DIFFUSE2)
    oracle_to_completion(o, steps_parked=24, max_steps=4000)
    oc = o.counters()
    assert np.isfinite(o.hits["t"]).mean() > 0.2 and int(oc["shadow_spawned"]) > 1.2 * int(np.isfinite(o.hits["t"]).sum())
    for free in (True, False):
        rt = api.RayTracer(sc, warp_width=64, sample_limit=limit, hit_capacity=limit, free_schedule=free, shading=DIFFUSE2)
        run_to_completion(rt, steps_parked=12, max_steps=4000)
        assert H.same_bytes(rt.hits(), o.hits), free
        assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6), free
        assert np.abs(rt.draw_frame().astype(int) - o.resolve_rgba8().astype(int)).max() <= 1, free
        c = rt.counters()
        for k in ("primary_spawned", "shadow_spawned"):
            assert int(c[k]) == int(oc[k]), (k, free)
        rt.close()
    o.close()
    big = S.build_scene(files, 1920, 1080, 1920 * 1080, grid=5)
    for free in (True, False):
        rt = api.RayTracer(big, warp_width=64, free_schedule=free, shading=DIFFUSE2)
        rt.init_frame()
        check_step_invariants(rt, rt.pool, 64, chunks=3, steps=16)
        rays, _, _ = rt.dump_state(index=False, groups=False)
        assert (rays["type"] >> 8).max() == 2
        rt.close()
