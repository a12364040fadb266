"""Behaviour of the reference's automaton that a plain closest-hit tracer would not have, pinned in the oracle on
scenes small enough to reason about by hand: a node that accepts more than 64 boxes drops them all (shader.cl:93),
a queue that reaches eight entries turns its eighth into a restart from the root (ray-tracer.cl:228-231) -- which, with
eight nodes at one entry parameter, never ends --, and a shadow ray's hit does not shorten its reach, so a shadowed
sample can add its weight to the pixel more than once (ray-tracer.cl:174-177).  No GPU."""
import numpy as np

from oracle import pyoracle as O
from raytracer_amd import layout as L, scene as S

from test_oracle_pipeline import run_to_completion


_most_seen = {}


def most_seen_instance(W, Hh):
    """The instance of the reference's field that most camera rays of a W x Hh frame hit first."""
    if (W, Hh) not in _most_seen:
        sc = S.build_scene([S.BUNNY], W, Hh, 8192)
        o = O.Oracle(sc, warp_width=32, sample_limit=W * Hh, hit_capacity=W * Hh)
        run_to_completion(o)
        inst = o.hits["instance"][np.isfinite(o.hits["t"])]
        o.close()
        _most_seen[(W, Hh)] = int(np.bincount(inst).argmax())
    return _most_seen[(W, Hh)]


def stacked_scene(copies, W=48, Hh=48):
    """The reference's scene with the root's list cut down to `copies` identical boxes of ONE instance (the one most
    camera rays see): same bounds, same model root, same instance matrix."""
    sc = S.build_scene([S.BUNNY], W, Hh, 8192)
    root = int(sc.data["cam"]["root_group"][0]) & L.GROUP_ID_MASK
    first, count = int(sc.groups["w"][root][0]), int(sc.groups["w"][root][1])
    assert copies <= count == 256
    sc.aabbs[first:first + copies] = sc.aabbs[first + most_seen_instance(W, Hh)]
    sc.groups["w"][root][1] = copies
    return sc, root


def entering(sc, root):
    """Pixels whose camera ray (first sample) enters the instance box, from a one-box scene's closest hits or box test."""
    o = O.Oracle(sc, warp_width=32, sample_limit=sc.width * sc.height)
    rays = o.camera_rays(np.arange(sc.width * sc.height, dtype=np.uint32))
    o.close()
    box = sc.aabbs[int(sc.groups["w"][root][0])]
    start, d = rays["start_min"][:, :3].astype(np.float64), rays["dir_max"][:, :3].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1, t2 = (box["min"] - start) / d, (box["max"] - start) / d
    near, far = np.minimum(t1, t2).max(axis=1), np.maximum(t1, t2).min(axis=1)
    return (far > near) & (far > 0.001)


def test_more_than_64_accepted_boxes_drop_the_whole_node():
    """65 identical instance boxes: a ray that enters one enters all, the 65th makes aabb_shader return 0
    (shader.cl:93, "artifacts"), and the sample ends in the sky although the model is in front of it."""
    limit = 48 * 48
    one, root1 = stacked_scene(1)
    ref = O.Oracle(one, warp_width=32, sample_limit=limit, hit_capacity=limit)
    run_to_completion(ref)
    seen = np.isfinite(ref.hits["t"])
    ref.close()
    assert seen.sum() >= 40                                         # the model is in view
    sc, root = stacked_scene(65)
    o = O.Oracle(sc, warp_width=32, sample_limit=limit, hit_capacity=limit)
    run_to_completion(o)
    assert not np.isfinite(o.hits["t"]).any()                       # nothing is ever hit
    assert np.allclose(o.area, np.tile(np.float32([0.5, 1, 1, 1]), (limit, 1)))   # every sample: sky_shader, weight one
    hist = o.hit_histogram()
    inside = entering(sc, root)
    assert inside.sum() >= seen.sum()                               # (every pixel that saw the model enters its box)
    assert hist[0] == limit and hist[1:].sum() == 0                 # one root visit per sample, each with nothing to show
    assert int(o.counters()["shadow_spawned"]) == 0
    o.close()


def test_a_few_stacked_boxes_change_no_closest_hit():
    """Two or three identical boxes: the model is visited once per box, and every sample that finishes has the closest
    hit of the one-box scene.  Not every sample does finish: the copies share one entry parameter, and a ray that also
    enters six of the model root's eight boxes fills its queue, restarts from the root and finds the same queue again
    (see the next test) -- one sample with two copies, a dozen with three."""
    limit = 48 * 48
    results = []
    for copies in (1, 2, 3):
        sc, _ = stacked_scene(copies)
        o = O.Oracle(sc, warp_width=32, sample_limit=limit, hit_capacity=limit)
        o.init_frame()
        o.step(24)
        o.set_warp_width(1)
        o.step(600)
        results.append((o.hits.copy(), o.area[:, 3] >= 1 - 1e-6, int(o.global_data["ray_count"][0]), int(o.work()["node_visits"])))
        o.close()
    hits1, finished1, left1, visits1 = results[0]
    assert left1 == 0 and finished1.all() and np.isfinite(hits1["t"]).sum() >= 40
    for hits, finished, left, visits in results[1:]:
        assert left == (~finished).sum() <= 40 and visits > visits1
        assert O.L.hit_record_dtype == hits.dtype and hits[finished].tobytes() == hits1[finished].tobytes()


def test_eight_nodes_at_one_entry_parameter_restart_for_ever():
    """Nine identical boxes: after the root visit the queue holds seven copies of the model's root and, as eighth
    entry, the restart marker -- the camera's root with the position of the entry it displaced (ray-tracer.cl:228-231).
    Working the queue off re-creates it: the sample never finishes, its slot's state recurs."""
    W = Hh = 48
    limit = W * Hh
    sc, root = stacked_scene(9, W, Hh)
    inside = entering(sc, root)
    pixel = int(np.flatnonzero(inside)[len(np.flatnonzero(inside)) // 2])
    o = O.Oracle(sc, warp_width=1, sample_limit=limit)
    o.init_frame_samples(np.array([pixel], np.uint32))
    o.step(1)                                                       # the root visit
    slot = int(o.ray_index["slot"][0])
    ray = o.ray_list[slot]
    model_root = int(sc.aabbs["group_id"][int(sc.groups["w"][root][0])])
    assert int(ray["queue_len"]) == L.MAX_QUEUE_LEN
    assert [int(g) for g in ray["queue"]["group_id"][:7]] == [model_root] * 7
    assert int(ray["queue"]["group_id"][7]) == int(sc.data["cam"]["root_group"][0])
    assert len(set(ray["queue"]["pos"].tolist())) == 1              # all nine boxes are entered at the same parameter
    o.step(600)
    assert int(o.global_data["ray_count"][0]) == 1                  # still in flight
    states = []
    for _ in range(120):
        o.step(1)
        states.append(o.ray_list[slot].tobytes())
    period = next((p for p in range(1, 60) if all(states[i] == states[i + p] for i in range(len(states) - p))), 0)
    assert period > 0
    o.close()


def test_a_shadowed_sample_can_darken_its_pixel_more_than_once(bunny_scene_64):
    """A shadow ray's hit adds the sample's weight to the pixel's denominator and leaves the ray's reach alone
    (ray-tracer.cl:174-177), so boxes behind the occluder are still entered and a second occluding leaf adds it
    again: a few pixels of a one-sample-per-pixel frame end with twice the weight."""
    limit = 64 * 64
    o = O.Oracle(bunny_scene_64, warp_width=32, sample_limit=limit, hit_capacity=limit)
    run_to_completion(o)
    w = o.area[:, 3]
    hit = np.isfinite(o.hits["t"])
    assert np.allclose(w[~hit], 1)                                  # sky: one sample, weight one
    assert np.all(w[hit] >= 1 - 1e-6) and np.allclose(w, np.round(w))
    assert 0 < (w > 1.5).sum() < hit.sum() // 10                    # rare, but there
    lit = hit & (w < 1.5) & (o.area[:, :3].sum(axis=1) > 0)
    assert lit.sum() > 100                                          # most hit samples see the light
    o.close()


def test_resolve_against_numpy(bunny_scene_64):
    """update_image (ray-tracer.cl:305-309): rgba = pow(area / (area.w + 1e-6), 1 / 2.2), stored as UNORM8 the way
    write_imagef converts (clamp, times 255, round to nearest even).  numpy's float32 power may differ from libm's powf
    in the last place, hence one step of tolerance; pixels nothing has been added to resolve to zero."""
    limit = 64 * 64
    o = O.Oracle(bunny_scene_64, warp_width=32, sample_limit=limit)
    run_to_completion(o)
    area = o.area.copy()
    image = o.resolve_rgba8().reshape(-1, 4)
    o.close()
    with np.errstate(invalid="ignore", divide="ignore"):
        v = np.power((area / (area[:, 3:4] + np.float32(1e-6))).astype(np.float32), np.float32(1 / 2.2))
    want = np.rint(np.clip(np.nan_to_num(v, nan=0.0), 0, 1) * np.float32(255)).astype(int)
    assert np.abs(image.astype(int) - want).max() <= 1
    assert (image[:, 3] >= 254).all()                                # alpha: w / (w + 1e-6) is one to within rounding
    empty = np.zeros((4, 4), np.float32)
    assert np.array_equal(np.rint(np.power(empty / (empty[:, 3:4] + np.float32(1e-6)), np.float32(1 / 2.2)) * 255), np.zeros((4, 4)))
