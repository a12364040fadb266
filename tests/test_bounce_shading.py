"""Shading beyond the reference's single shadow ray (RT_FLAG_SHADING): the mirror reflection the reference
left commented out (shader.cl:37-39) and a cosine-weighted diffuse bounce.  No reference output exists for
either (the compiled reference never runs them): the oracle's restatement is the definition, checked here
for its invariants on the CPU and matched bit for bit by the HIP path on the GPU.  Parity unpinned."""
import numpy as np
import pytest

from raytracer_amd import api, layout as L, scene as S

import helpers as H
from test_oracle_pipeline import run_to_completion as oracle_to_completion

MODES = [(api.SHADE_MIRROR, 2), (api.SHADE_DIFFUSE, 3)]


@pytest.mark.parametrize("mode,bounces", MODES)
def test_oracle_bounces_are_schedule_independent(bunny_scene_64, mode, bounces):
    from oracle import pyoracle as O
    limit = 64 * 64
    flags = api.shading_flags(mode, bounces)
    results = []
    for warp, parked in ((32, 24), (1, 3)):
        o = O.Oracle(bunny_scene_64, warp_width=warp, sample_limit=limit, hit_capacity=limit, shading=flags)
        oracle_to_completion(o, steps_parked=parked, max_steps=1500)
        results.append((o.hits.copy(), o.area.copy(), o.counters()))
        o.close()
    assert H.same_bytes(results[0][0], results[1][0])
    assert np.allclose(results[0][1], results[1][1], rtol=1e-5, atol=1e-6)
    # the camera rays' hit records are those of the reference shading: bounces do not overwrite them
    o = O.Oracle(bunny_scene_64, warp_width=32, sample_limit=limit, hit_capacity=limit)
    oracle_to_completion(o)
    assert H.same_bytes(o.hits, results[0][0])
    plain = o.counters()
    o.close()
    # every sample that hit something bounced at least once, so more secondary rays were started
    assert int(results[0][2]["shadow_spawned"]) > int(plain["shadow_spawned"])
    area = results[0][1]
    assert np.isfinite(area).all() and np.all(area[:, 3] >= 1 - 1e-5) and np.all(area[:, :3] >= 0)


def test_oracle_bounce_state(bunny_scene_64):
    """The bounce count lives above the low byte of RayQueue.type and never exceeds the configured number;
    directions stay unit vectors; a diffuse bounce leaves on the side of the surface it arrived on."""
    from oracle import pyoracle as O
    # granularity 4: with the reference's 32, a pool this small parks nearly every ray in a leaf's tail
    o = O.Oracle(bunny_scene_64, warp_width=4, shading=api.shading_flags(api.SHADE_DIFFUSE, 3))
    o.init_frame()
    seen = set()
    for _ in range(60):
        o.step(1)
        rl = o.ray_list
        depth = rl["type"] >> 8
        seen.update(np.unique(depth).tolist())
        assert depth.max() <= 3 and np.all((rl["type"] & 0xFF) <= 1)
        assert np.all((rl["type"] & 0xFF == 0) | (depth == 0))          # shadow rays carry no count
        d = rl["dir_max"][:, :3]
        assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-5)
        fresh = (depth > 0) & (rl["queue_len"] == 1) & (rl["start_min"][:, 3] == np.float32(0.001))
        if fresh.any():
            n = rl["norm"][fresh, :3]
            n = n / np.linalg.norm(n, axis=1, keepdims=True)
            # the new direction is within 90 degrees of +-normal on the incoming side: |cos| > 0 and the
            # distribution is cosine-weighted, so its mean is well above that of a uniform hemisphere (0.5)
            cos = np.abs(np.einsum("ij,ij->i", n, d[fresh]))
            assert cos.min() >= 0
    assert seen >= {0, 1, 2, 3}
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,bounces", MODES)
def test_gpu_bounce_state_matches_oracle(bunny_scene_128, mode, bounces):
    from oracle import pyoracle as O
    from test_gpu_pipeline import assert_same_state
    flags = api.shading_flags(mode, bounces)
    cap = 128 * 128 * 2
    rt = api.RayTracer(bunny_scene_128, warp_width=4, hit_capacity=cap, shading=flags)   # (see test_oracle_bounce_state)
    o = O.Oracle(bunny_scene_128, warp_width=4, hit_capacity=cap, shading=flags)
    rt.init_frame(); o.init_frame()
    for step in range(60):
        rt.make_step(1); o.step(1)
        assert_same_state(rt, o, step)
    assert (o.ray_list["type"] >> 8).max() == bounces
    assert H.same_bytes(rt.hits(), o.hits)
    assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6)
    c, oc = rt.counters(), o.counters()
    for k in ("primary_spawned", "shadow_spawned", "box_tests", "tri_tests"):
        assert int(c[k]) == int(oc[k]), k
    rt.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,bounces", MODES)
def test_gpu_bounce_frame_on_the_free_schedule(bunny_scene_64, mode, bounces):
    """A finished frame with bounces on the free schedule against the oracle's: same samples."""
    from oracle import pyoracle as O
    from test_gpu_pipeline import run_to_completion
    limit = 64 * 64
    flags = api.shading_flags(mode, bounces)
    o = O.Oracle(bunny_scene_64, warp_width=32, sample_limit=limit, hit_capacity=limit, shading=flags)
    oracle_to_completion(o, max_steps=1500)
    for free in (True, False):
        rt = api.RayTracer(bunny_scene_64, warp_width=64, sample_limit=limit, hit_capacity=limit, free_schedule=free, shading=flags)
        run_to_completion(rt, steps_parked=12, max_steps=1500)
        assert H.same_bytes(rt.hits(), o.hits), free
        assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6), free
        assert np.abs(rt.draw_frame().astype(int) - o.resolve_rgba8().astype(int)).max() <= 1, free
        assert int(rt.counters()["shadow_spawned"]) == int(o.counters()["shadow_spawned"]), free
        rt.close()
    o.close()
