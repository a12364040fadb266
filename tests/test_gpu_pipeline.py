"""The whole hot path on the GPU through the C ABI against the oracle: state after every step,
finished frames, per-sample hit records, closest-hit queries, and properties at the headline size."""
import os

import numpy as np
import pytest

from raytracer_amd import api, layout as L, scene as S

import helpers as H

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def assert_same_state(rt, o, step):
    """Integer state bit-exact; float state bit-exact too (same operations in the same order) except
    the accumulator, whose atomic adds may arrive in a different order within a step."""
    rays, index, groups = rt.dump_state()
    assert H.same_bytes(rt.global_data(), o.global_data), step
    assert H.same_bytes(index, o.ray_index), step
    assert H.same_bytes(groups, o.grp_data), step
    assert H.ray_fields_equal(rays, o.ray_list) == [], step


@pytest.mark.parametrize("warp", [32, 4])
def test_state_matches_oracle_step_by_step(bunny_scene_128, warp):
    """The reference's granularity (32) parks most rays of a pool this small in the leaves' tails, so few
    samples get as far as the shaders in 48 steps; with granularity 4 tens of thousands do."""
    from oracle import pyoracle as O
    cap = 16384 * 8
    rt = api.RayTracer(bunny_scene_128, warp_width=warp, hit_capacity=cap)
    o = O.Oracle(bunny_scene_128, warp_width=warp, hit_capacity=cap)
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
    assert int(c["shadow_spawned"]) > (10000 if warp == 4 else 0)
    assert rt.current_ray() == int(o.global_data["pixel_offset"][0])
    assert np.abs(rt.draw_frame().astype(int) - o.resolve_rgba8().astype(int)).max() <= 1
    rt.close(); o.close()


def test_state_matches_oracle_with_second_model(standin_ply):
    from oracle import pyoracle as O
    sc = S.build_scene([S.BUNNY, standin_ply], 128, 128, 16384)
    rt = api.RayTracer(sc, warp_width=32)
    o = O.Oracle(sc, warp_width=32)
    rt.init_frame(); o.init_frame()
    for chunk in range(6):
        rt.make_step(8); o.step(8)
        assert_same_state(rt, o, chunk)
    rt.close(); o.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_stage_timers_do_not_change_the_state(bunny_scene_128, mode):
    """With stage timing on, rt_step launches its kernels one by one (mode 2: on one stream) instead of
    replaying the captured step; the state must be the oracle's either way, and every stage gets a time."""
    from oracle import pyoracle as O
    rt = api.RayTracer(bunny_scene_128, warp_width=32)
    o = O.Oracle(bunny_scene_128, warp_width=32)
    rt.init_frame(); o.init_frame()
    rt.make_step(4); o.step(4)          # captured step
    rt.set_profiling(mode)
    rt.make_step(12); o.step(12)        # separate launches with events
    times, steps = rt.stage_times()
    assert steps == 12 and set(times) == set(api.STAGE_NAMES)
    assert all(ms > 0 for ms in times.values())
    rt.set_profiling(0)
    rt.make_step(4); o.step(4)          # captured step again
    assert_same_state(rt, o, mode)
    rt.close(); o.close()


def test_more_than_65536_groups():
    """70 separately built models: 72,704 groups, so the sort takes a third 8-bit pass and the
    compaction more than one round of block sums."""
    from oracle import pyoracle as O
    sc = S.build_scene([S.BUNNY] * 70, 128, 128, 16384, n_obj=140)
    assert sc.group_count > 65536
    rt = api.RayTracer(sc, warp_width=32)
    o = O.Oracle(sc, warp_width=32)
    rt.init_frame(); o.init_frame()
    for chunk in range(5):
        rt.make_step(8); o.step(8)
        assert_same_state(rt, o, chunk)
    rt.close(); o.close()


def test_pipeline_golden_state(bunny_scene_128):
    g = np.load(os.path.join(GOLDEN, "pipeline_bunny128.npz"))
    rt = api.RayTracer(bunny_scene_128, warp_width=int(g["warp_width"]), hit_capacity=len(g["hits"]))
    rt.init_frame()
    rt.make_step(int(g["steps"]))
    rays, index, groups = rt.dump_state()
    assert H.same_bytes(rt.global_data(), g["global_data"])
    assert H.same_bytes(index, g["ray_index"]) and H.same_bytes(groups, g["grp_data"])
    assert H.same_bytes(rt.hits(), g["hits"])
    assert np.allclose(rt.area(), g["area"], rtol=1e-5, atol=1e-6)
    import hashlib
    for name, digest in g["ray_field_sha"]:
        assert hashlib.sha256(np.ascontiguousarray(rays[str(name)]).tobytes()).hexdigest() == str(digest), name
    rt.close()


def run_to_completion(r, steps_parked=24, max_steps=400):
    r.init_frame()
    r.make_step(steps_parked)
    r.set_warp_width(1)
    r.make_step(1)
    steps = steps_parked + 1
    while int(r.global_data()["ray_count"][0]):
        r.make_step(1)
        steps += 1
        assert steps < max_steps
    return steps


def test_finished_frame_golden(bunny_scene_64):
    g = np.load(os.path.join(GOLDEN, "drain_bunny64.npz"))
    limit = int(g["sample_limit"])
    rt = api.RayTracer(bunny_scene_64, warp_width=32, sample_limit=limit, hit_capacity=limit)
    assert run_to_completion(rt) == int(g["steps"])
    assert H.same_bytes(rt.hits(), g["hits"])                                    # ids exact, t exact
    assert np.allclose(rt.area(), g["area"], rtol=1e-5, atol=1e-6)              # colour sums
    assert np.abs(rt.draw_frame().astype(int) - g["image"].astype(int)).max() <= 1  # 1 LSB at 8 bit
    # the same image resolved into the context's own device buffer
    import ctypes
    host = rt.draw_frame()
    rt.draw_frame_device()
    back = np.zeros_like(host)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert rt.image_device()
    assert hip.hipMemcpy(back.ctypes.data, rt.image_device(), back.nbytes, 2) == 0  # hipMemcpyDeviceToHost
    assert np.array_equal(back, host)
    c = rt.counters()
    gc = g["counters"][0]
    assert int(c["primary_spawned"]) == limit == int(gc["primary_spawned"])
    assert int(c["shadow_spawned"]) == int(gc["shadow_spawned"])
    rt.close()


def test_wave64_schedule_gives_the_same_samples(bunny_scene_64):
    """Production granularity (64) and odd granularities change the schedule, never a sample's hit."""
    g = np.load(os.path.join(GOLDEN, "drain_bunny64.npz"))
    limit = int(g["sample_limit"])
    for warp, parked in ((64, 12), (1, 2), (48, 9)):
        rt = api.RayTracer(bunny_scene_64, warp_width=warp, sample_limit=limit, hit_capacity=limit)
        run_to_completion(rt, steps_parked=parked)
        assert H.same_bytes(rt.hits(), g["hits"]), warp
        assert np.allclose(rt.area(), g["area"], rtol=1e-5, atol=1e-6), warp
        assert np.abs(rt.draw_frame().astype(int) - g["image"].astype(int)).max() <= 1, warp
        rt.close()


@pytest.mark.parametrize("warp,parked", [(64, 12), (32, 24), (1, 2)])
def test_free_schedule_gives_the_same_samples(bunny_scene_64, warp, parked):
    """RT_FLAG_FREE_SCHEDULE (plain-order sort, shading and spawn inside the visiting kernels) changes
    which slot holds which sample at a given step, never a sample: hit records exact, colour sums and
    image as for any other schedule."""
    g = np.load(os.path.join(GOLDEN, "drain_bunny64.npz"))
    limit = int(g["sample_limit"])
    rt = api.RayTracer(bunny_scene_64, warp_width=warp, sample_limit=limit, hit_capacity=limit, free_schedule=True)
    run_to_completion(rt, steps_parked=parked)
    assert H.same_bytes(rt.hits(), g["hits"])
    assert np.allclose(rt.area(), g["area"], rtol=1e-5, atol=1e-6)
    assert np.abs(rt.draw_frame().astype(int) - g["image"].astype(int)).max() <= 1
    c = rt.counters()
    gc = g["counters"][0]
    assert int(c["primary_spawned"]) == limit and int(c["shadow_spawned"]) == int(gc["shadow_spawned"])
    rt.close()


def test_free_schedule_keeps_the_step_invariants(bunny_scene_128):
    """Sorted by group, whole waves per group, every slot exactly once, no shader key left in the index."""
    rt = api.RayTracer(bunny_scene_128, warp_width=64, free_schedule=True)
    rt.init_frame()
    pool, prev, checked = rt.pool, rt.pool, 0
    for chunk in range(6):
        rt.make_step(8)
        g = rt.global_data()[0]
        n = int(g["ray_count"])
        rays, index, groups = rt.dump_state()
        ids = index["key"][:n] & L.GROUP_ID_MASK
        assert n % 64 == 0 and np.all(np.diff(ids.astype(np.int64)) >= 0)
        assert np.all(ids.reshape(-1, 64).min(axis=1) == ids.reshape(-1, 64).max(axis=1))
        assert np.array_equal(np.sort(index["slot"]), np.arange(pool, dtype=np.uint32))
        shader = index["key"][:n] >> 28
        assert np.all((shader == L.SH_AABB) | (shader == L.SH_MESH) | (index["key"][:n] == 0))
        # a travelling ray's head is its key; a finished slot (key 0) keeps whatever its last sample left
        live = index["key"][:n] != 0
        assert np.array_equal(rays["queue"]["group_id"][index["slot"][:n][live], 0], index["key"][:n][live])
        # its queue (exported from the stack the free schedule keeps) is in order of entry position, nearest first,
        # and holds groups of the scene only
        q = rays["queue"][index["slot"][:n][live]]
        qlen = rays["queue_len"][index["slot"][:n][live]]
        checked += int(live.sum())
        assert not live.any() or (qlen.min() >= 1 and qlen.max() <= 8)
        for k in range(1, 8):
            has = qlen > k
            assert np.all(q["pos"][has, k] >= q["pos"][has, k - 1])
            assert np.all((q["group_id"][has, k] & L.GROUP_ID_MASK) < bunny_scene_128.group_count)
        assert int(g["pixel_offset"]) >= prev
        prev = int(g["pixel_offset"])
    assert checked > 1000
    c = rt.counters()
    assert int(c["primary_spawned"]) == prev
    rt.close()


def test_trace_closest_matches_oracle_and_brute_force(bunny_scene_64):
    from oracle import pyoracle as O
    from test_oracle_pipeline import camera_rays
    sc = bunny_scene_64
    rng = np.random.default_rng(21)
    rays = camera_rays(sc, 3000, rng)
    # add incoherent rays: random origins inside the field, random directions
    extra = np.zeros(2000, L.ray_dtype)
    extra["start_min"][:, 0] = rng.uniform(-2, 2, 2000)
    extra["start_min"][:, 1] = rng.uniform(0, 4, 2000)
    extra["start_min"][:, 2] = rng.uniform(-1, 1, 2000)
    extra["start_min"][:, 3] = 0.001
    d = rng.normal(size=(2000, 3))
    extra["dir_max"][:, :3] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    extra["dir_max"][:, 3] = np.inf
    rays = np.concatenate([rays, extra])
    rt = api.RayTracer(sc, warp_width=64)
    got = rt.trace_closest(rays)
    o = O.Oracle(sc, warp_width=32)
    want = o.trace_closest(rays)
    assert H.same_bytes(got, want)            # t, leaf, instance, triangle: all bit-exact
    free = api.RayTracer(sc, warp_width=64, free_schedule=True)
    assert H.same_bytes(free.trace_closest(rays), want)
    free.close()
    sub = np.arange(0, len(rays), 25)
    brute = H.brute_force_closest(sc, rays[sub])
    hit = np.isfinite(brute)
    assert np.array_equal(np.isfinite(got["t"][sub]), hit)
    assert np.allclose(got["t"][sub][hit], brute[hit], rtol=1e-4)
    # the context is reusable for a frame afterwards
    rt.init_frame(); rt.make_step(4)
    assert int(rt.global_data()["ray_count"][0]) > 0
    assert len(rt.trace_closest(np.zeros(0, L.ray_dtype))) == 0
    rt.close(); o.close()


def test_incoherent_rays_in_a_grid_of_instances(standin_ply):
    """The shape of the incoherent workload: a regular 4x4x4 grid of instances of two models, rays with
    random origins inside the grid and random directions; hits bit-exact against the oracle."""
    from oracle import pyoracle as O
    sc = S.build_scene([S.BUNNY, standin_ply], 64, 64, 65536, grid=4)
    root = int(sc.data["cam"]["root_group"][0]) & L.GROUP_ID_MASK
    first, count = int(sc.groups["w"][root][0]), int(sc.groups["w"][root][1])
    assert count == 64
    boxes = sc.aabbs[first:first + count]
    lo, hi = boxes["min"].min(axis=0), boxes["max"].max(axis=0)
    rng = np.random.default_rng(5)
    n = 40000
    rays = np.zeros(n, L.ray_dtype)
    rays["start_min"][:, :3] = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    rays["start_min"][:, 3] = 0.001
    d = rng.normal(size=(n, 3))
    rays["dir_max"][:, :3] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["dir_max"][:, 3] = np.inf
    rt = api.RayTracer(sc, warp_width=64)
    o = O.Oracle(sc, warp_width=32, threads=8)
    got, want = rt.trace_closest(rays), o.trace_closest(rays)
    assert H.same_bytes(got, want)
    assert 0.01 < np.isfinite(got["t"]).mean() < 1.0
    rt.close(); o.close()


def test_instance_list_longer_than_256_boxes():
    """A 7x7x7 grid puts 343 boxes under the scene root: more than the box kernel's one-byte places hold, so the scene
    runs the box kernel with two-byte places, which scans the root's list in two parts of at most 256 boxes.  Hits
    bit-exact against the oracle, either schedule."""
    from oracle import pyoracle as O
    sc = S.build_scene([S.BUNNY], 64, 64, 16384, grid=7)
    root = int(sc.data["cam"]["root_group"][0]) & L.GROUP_ID_MASK
    first, count = int(sc.groups["w"][root][0]), int(sc.groups["w"][root][1])
    assert count == 343
    boxes = sc.aabbs[first:first + count]
    lo, hi = boxes["min"].min(axis=0), boxes["max"].max(axis=0)
    rng = np.random.default_rng(343)
    n = 6000
    rays = np.zeros(n, L.ray_dtype)
    rays["start_min"][:, :3] = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    rays["start_min"][:, 3] = 0.001
    d = rng.normal(size=(n, 3))
    rays["dir_max"][:, :3] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["dir_max"][:, 3] = np.inf
    o = O.Oracle(sc, warp_width=32, threads=8)
    want = o.trace_closest(rays)
    o.close()
    assert 0.01 < np.isfinite(want["t"]).mean() < 1.0
    for free in (False, True):
        rt = api.RayTracer(sc, warp_width=64, free_schedule=free)
        assert H.same_bytes(rt.trace_closest(rays), want), free
        rt.close()


def test_api_errors(bunny_scene_64):
    rt = api.RayTracer(bunny_scene_64)
    with pytest.raises(api.RtError, match="RT_ERR_NOT_READY"):
        rt.make_step(1)                       # before init_frame
    with pytest.raises(api.RtError, match="RT_ERR_INVALID_VALUE"):
        rt.trace_closest(np.zeros(rt.pool + 1, L.ray_dtype))
    bad = S.build_scene([S.BUNNY], 64, 64, 8192, n_obj=4)
    bad.aabbs["group_id"][0] = 0x40FFFFF0     # box pointing outside the group table
    with pytest.raises(api.RtError, match="RT_ERR_INVALID_VALUE"):
        rt.upload(bad)
    affine = S.build_scene([S.BUNNY], 64, 64, 8192, n_obj=4)
    affine.aabbs["group_id"][0] = (affine.aabbs["group_id"][0] & ~np.uint32(0xF << 24)) | np.uint32(L.TR_AFFINE << 24)
    with pytest.raises(api.RtError, match="RT_ERR_UNSUPPORTED"):
        rt.upload(affine)                     # tr_affine is an empty case in the reference (shader.cl:69-71)
    other = S.build_scene([S.BUNNY], 32, 32, 8192, n_obj=4)
    with pytest.raises(api.RtError, match="does not match"):
        rt.upload(other)                      # camera size differs from the context
    rt.close()


def test_band_contexts_reproduce_the_full_frame():
    """Two row bands (as two GPUs would render them) give the full frame's samples: same hit records
    per pixel sample, same image."""
    W, Hh = 64, 64
    limit = W * Hh
    full_sc = S.build_scene([S.BUNNY, S.BUNNY], W, Hh, 8192)
    full = api.RayTracer(full_sc, warp_width=64, sample_limit=limit, hit_capacity=limit)
    run_to_completion(full, steps_parked=8)
    image = full.draw_frame()
    area = full.area()
    hits = full.hits()
    full.close()
    parts, part_hits = [], []
    for first in (0, 40):
        rows = 40 if first == 0 else 24
        sc = S.build_scene([S.BUNNY, S.BUNNY], W, Hh, 8192, band_first=first, band_rows=rows)
        rt = api.RayTracer(sc, warp_width=64, sample_limit=W * rows, hit_capacity=W * rows)
        run_to_completion(rt, steps_parked=8)
        parts.append(rt.area())
        part_hits.append(rt.hits())
        assert np.abs(rt.draw_frame().astype(int) - image[first:first + rows].astype(int)).max() <= 1
        rt.close()
    assert np.allclose(np.concatenate(parts), area, rtol=1e-5, atol=1e-6)
    assert H.same_bytes(np.concatenate(part_hits), hits)


def test_headline_size_properties(standin_ply):
    """1920x1080, 2,080,768 slots, bunny + (reduced) second model: invariants that do not need the oracle."""
    sc = S.build_scene([S.BUNNY, standin_ply], 1920, 1080, 1920 * 1080)
    rt = api.RayTracer(sc, warp_width=64)
    rt.init_frame()
    pool = rt.pool
    assert pool == 2080768
    prev = pool
    for chunk in range(4):
        rt.make_step(16)
        g = rt.global_data()[0]
        n = int(g["ray_count"])
        assert n <= pool and n % 64 == 0
        rays, index, groups = rt.dump_state(rays=(chunk == 3))
        ids = index["key"][:n] & L.GROUP_ID_MASK
        assert np.all(np.diff(ids.astype(np.int64)) >= 0)
        assert np.all(ids.reshape(-1, 64).min(axis=1) == ids.reshape(-1, 64).max(axis=1))
        assert np.array_equal(np.sort(index["slot"]), np.arange(pool, dtype=np.uint32))
        assert int(groups["count"][:, 0].sum()) == n and n + int(groups["count"][:, 1].sum()) == pool
        assert int(g["pixel_offset"]) >= prev
        prev = int(g["pixel_offset"])
    assert np.array_equal(rays["queue"]["group_id"][index["slot"], 0], index["key"])
    assert np.all((rays["queue_len"] >= 1) & (rays["queue_len"] <= 8))
    c = rt.counters()
    assert int(c["primary_spawned"]) == prev + int(rt.global_data()["pixel_count"][0]) - 0 or int(c["primary_spawned"]) == prev
    area = rt.area()
    assert np.isfinite(area).all() and area[:, 3].max() > 0
    img = rt.draw_frame()
    assert img.shape == (1080, 1920, 4) and img[..., :3].max() > 100
    rt.close()


def test_cpp_host_renders_the_same_frame(tmp_path, bunny_scene_128):
    """The C++ host (raytracer_amd/host/rt_main: scene build, class RayTracer, frame loop) over the C ABI
    against the Python plumbing on the same scene: one frame of 32 steps."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "raytracer_amd", "host", "rt_main")
    assert os.path.exists(exe), "run `make host` (or __graft_entry__.build())"
    out = str(tmp_path / "frame.ppm")
    run = subprocess.run([exe, "--width", "128", "--height", "128", "--rays", "16384", "--frames", "1", "--steps", "32",
                          "--warp", "32", "--schedule", "reference", "--arith", "strict", "--models", S.BUNNY + ";" + S.BUNNY, "--out", out],
                         cwd=root, capture_output=True, text=True, timeout=240)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "Frame ready in" in run.stdout
    with open(out, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == b"255"
        ppm = np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)
    rt = api.RayTracer(bunny_scene_128, warp_width=32)
    rt.init_frame()
    rt.make_step(32)
    image = rt.draw_frame()
    rt.close()
    assert (w, h) == (128, 128)
    # the PPM is written bottom row first (the reference displays its texture with y up)
    assert np.abs(ppm.astype(int) - image[::-1, :, :3].astype(int)).max() <= 1


def _read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)[::-1]   # stored bottom row first


def test_cpp_host_tiles_the_frame(tmp_path):
    """rt_main's multi-GPU path on the one GPU of the box: one sample per pixel, run until no ray is in flight.
    (a) two bands rendered alone, as two ranks would, assemble to the single-context frame; (b) two bands
    in one process (--devices 0,0: two contexts, gathered by device copies) give the same frame; (c) so does the
    RCCL all-gather with a single rank (--rccl), which is the gather --gpus N uses on N distinct devices."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "raytracer_amd", "host", "rt_main")
    base = [exe, "--width", "128", "--height", "96", "--spp", "1", "--steps", "12", "--models", S.BUNNY + ";" + S.BUNNY]

    def run(extra, name):
        out = str(tmp_path / name)
        r = subprocess.run(base + extra + ["--out", out], cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "still in flight" not in r.stdout
        return _read_ppm(out), r.stdout

    full, _ = run([], "full.ppm")
    assert full.shape == (96, 128, 3) and full.max() > 100
    top, _ = run(["--band-first", "0", "--band-rows", "40"], "top.ppm")
    bottom, _ = run(["--band-first", "40", "--band-rows", "56"], "bottom.ppm")
    assert np.abs(np.concatenate([top, bottom]).astype(int) - full.astype(int)).max() <= 1
    both, log = run(["--devices", "0,0"], "both.ppm")
    assert "2 band(s), gathered by device copies" in log
    assert np.abs(both.astype(int) - full.astype(int)).max() <= 1
    one, log = run(["--rccl"], "rccl.ppm")
    assert "1 band(s), gathered with RCCL" in log
    assert np.abs(one.astype(int) - full.astype(int)).max() <= 1


@pytest.mark.parametrize("models", [13, 33])
def test_free_schedule_with_many_groups(models):
    """13 separately built models give 13,824 groups: the rank kernel's per-group counters then need more than the
    48 KB of LDS a workgroup gets without asking (54 KB, granted on gfx950).  33 models give 34,304 groups, beyond what
    the rank kernel holds at all: such a scene takes the plain-order sort and the two compaction kernels.  Finished
    frames against the oracle on either path."""
    from oracle import pyoracle as O
    from test_oracle_pipeline import run_to_completion as oracle_to_completion
    W, Hh = 96, 64
    limit = W * Hh
    sc = S.build_scene([S.BUNNY] * models, W, Hh, 8192, n_obj=2 * models)
    assert (12288 < sc.group_count <= 32768) if models == 13 else sc.group_count > 32768
    o = O.Oracle(sc, warp_width=32, sample_limit=limit, hit_capacity=limit)
    oracle_to_completion(o, steps_parked=24)
    rt = api.RayTracer(sc, warp_width=64, sample_limit=limit, hit_capacity=limit, free_schedule=True)
    run_to_completion(rt, steps_parked=12)
    assert H.same_bytes(rt.hits(), o.hits)
    assert np.allclose(rt.area(), o.area, rtol=1e-5, atol=1e-6)
    assert np.abs(rt.draw_frame().astype(int) - o.resolve_rgba8().astype(int)).max() <= 1
    c, oc = rt.counters(), o.counters()
    for k in ("primary_spawned", "shadow_spawned"):
        assert int(c[k]) == int(oc[k]), k
    rt.close(); o.close()
