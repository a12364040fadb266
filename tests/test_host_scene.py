"""Host-side scene build (raytracer_amd/host): pinned byte for byte against a real build of the
reference's model.cpp (oracle/_ref/libref_model.so) and against digests of that build committed
under tests/golden/."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as O
from raytracer_amd import layout as L, scene as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def scene_digests(sc):
    kinds = {}
    for gid in list(sc.aabbs["group_id"]) + [int(sc.data["cam"]["root_group"][0])]:
        kinds[int(gid) & L.GROUP_ID_MASK] = (int(gid) >> L.GROUP_SH_SHIFT) & 0xF
    w = sc.groups["w"].copy()
    defined = np.zeros_like(w, dtype=bool)
    for idx, kind in kinds.items():
        defined[idx, :3 if kind == L.SH_AABB else 4] = True
    w[~defined] = 0
    return {"groups": digest(w), "aabbs": digest(sc.aabbs), "vertex_pos": digest(sc.vertices["pos"][:, :3]),
            "vertex_norm": digest(sc.vertices["norm"][:, :3]), "triangles": digest(sc.triangles),
            "matrices": digest(sc.matrices), "counts": [len(sc.groups), len(sc.aabbs), len(sc.vertices), len(sc.triangles)],
            "group_count": sc.group_count}


def test_bunny_scene_shape(bunny_scene_128):
    sc = bunny_scene_128
    # two Model objects of the bunny: 6 + 2 * 1033 groups; one model = root of 8 boxes + 8 x 128 + 1024 leaves
    assert len(sc.groups) == 6 + 2 * 1033
    assert len(sc.aabbs) == 256 + 2 * 1032
    assert len(sc.triangles) == 2 * 69451
    assert sc.group_count == 2560 and sc.group_count % 512 == 0
    assert int(sc.data["cam"]["root_group"][0]) == (5 | L.TR_IDENTITY << 24 | L.SH_AABB << 28)
    root = sc.groups["w"][5]
    assert tuple(root[:3]) == (0, 256, 1)
    leaves = (sc.aabbs["group_id"] >> 28) == L.SH_MESH
    tri_counts = sc.groups["w"][sc.aabbs["group_id"][leaves] & L.GROUP_ID_MASK][:, 2]
    assert tri_counts.min() >= 64 and tri_counts.max() < 128


def test_first_matrices_follow_glibc_random():
    sc = S.build_scene([S.BUNNY], 64, 64, 8192, n_obj=2)
    # first four values of glibc random() with its default seed
    r = [1804289383, 846930886, 1681692777, 1714636915]
    alpha = 2 * 3.14159265359 * r[0] / 2147483647
    m = sc.matrices[0]
    assert m["x"][0] == np.float32(np.cos(alpha)) and m["y"][0] == np.float32(np.sin(alpha))
    assert m["x"][3] == np.float32(4.0 * r[1] / 2147483647 - 2)
    assert m["y"][3] == np.float32(4.0 * r[2] / 2147483647)
    assert m["z"][3] == np.float32(2.0 * r[3] / 2147483647 - 1)


@pytest.mark.skipif(O.ref_model_lib() is None, reason="oracle/_ref/libref_model.so not built (needs /root/reference)")
def test_builder_matches_reference_model_cpp(bunny_scene_128, standin_ply):
    for files in ([S.BUNNY, S.BUNNY], [S.BUNNY, standin_ply]):
        sc = bunny_scene_128 if files[1] == S.BUNNY else S.build_scene(files, 128, 128, 16384)
        groups, aabbs, vertices, triangles = O.ref_model_build(files, sc.matrices)
        assert np.array_equal(triangles, sc.triangles)
        assert aabbs.tobytes() == sc.aabbs.tobytes()
        assert np.array_equal(vertices["pos"][:, :3].view(np.uint32), sc.vertices["pos"][:, :3].view(np.uint32))
        assert np.array_equal(vertices["norm"][:, :3].view(np.uint32), sc.vertices["norm"][:, :3].view(np.uint32))
        # group table: the reference leaves unused words uninitialised, compare the defined ones
        for gid in list(aabbs["group_id"]) + [int(sc.data["cam"]["root_group"][0])]:
            idx, kind = int(gid) & L.GROUP_ID_MASK, (int(gid) >> 28) & 0xF
            words = 3 if kind == L.SH_AABB else 4
            assert np.array_equal(groups["w"][idx][:words], sc.groups["w"][idx][:words]), idx


@pytest.mark.skipif(O.ref_model_lib() is None, reason="oracle/_ref/libref_model.so not built")
def test_set_camera_matches_reference():
    a = S.set_camera(640, 480, 0.7, (0.1, -2.0, 0.3), (0.2, 1.0, -0.1))
    b = O.ref_set_camera(640, 480, 0.7, (0.1, -2.0, 0.3), (0.2, 1.0, -0.1))
    for f in ("eye", "top_left", "dx", "dy"):
        assert np.array_equal(a[f][0][:3].view(np.uint32), b[f][0][:3].view(np.uint32)), f
    assert a["width"][0] == 640 and a["height"][0] == 480


def test_scene_matches_golden_digests_of_the_reference_build(bunny_scene_128):
    """Golden digests were produced from the arrays the reference's model.cpp emitted (make_golden.py)."""
    golden = json.load(open(os.path.join(GOLDEN, "scene_bunny_digests.json")))
    assert scene_digests(bunny_scene_128) == golden["bunny_x2"]


def test_standin_ply_roundtrip(standin_ply):
    sc = S.build_scene([standin_ply], 64, 64, 8192, n_obj=4)
    assert len(sc.triangles) == 2 * 264 * 132
    assert sc.resident_triangles == 4 * 2 * 264 * 132


def test_band_scene_keeps_full_camera():
    full = S.build_scene([S.BUNNY], 64, 64, 8192, n_obj=4)
    band = S.build_scene([S.BUNNY], 64, 64, 8192, n_obj=4, band_first=16, band_rows=8)
    assert band.row_offset == 16 and band.height == 8 and band.width == 64
    for f in ("eye", "top_left", "dx", "dy"):
        assert np.array_equal(full.data["cam"][f], band.data["cam"][f])


def test_bad_model_path_is_reported():
    with pytest.raises(RuntimeError, match="cannot load model"):
        S.build_scene(["/nonexistent.ply"], 64, 64, 8192)


def test_mesh_cache_gives_the_same_scene(tmp_path, monkeypatch):
    """Model::load keeps the parsed arrays of a PLY file in a binary cache keyed by a hash of the file's
    bytes: the scene built from the cache is byte-identical, a damaged cache file is ignored."""
    import glob
    names = ("groups", "aabbs", "vertices", "triangles", "matrices", "data")
    monkeypatch.setenv("RT_MESH_CACHE", "")                      # switched off: parse the text
    plain = S.build_scene([S.BUNNY, S.BUNNY], 64, 64, 8192)
    assert plain.models_from_cache == 0
    monkeypatch.setenv("RT_MESH_CACHE", str(tmp_path))
    first = S.build_scene([S.BUNNY, S.BUNNY], 64, 64, 8192)      # second model file of the list already finds the first's cache
    files = glob.glob(str(tmp_path / "*.rtmesh"))
    assert len(files) == 1 and first.models_from_cache == 1
    again = S.build_scene([S.BUNNY, S.BUNNY], 64, 64, 8192)
    assert again.models_from_cache == 2
    for sc in (first, again):
        for n in names:
            assert np.ascontiguousarray(getattr(sc, n)).tobytes() == np.ascontiguousarray(getattr(plain, n)).tobytes(), n
    raw = bytearray(open(files[0], "rb").read())
    open(files[0], "wb").write(raw[:len(raw) // 2])              # truncated: must be ignored, then rewritten
    damaged = S.build_scene([S.BUNNY], 64, 64, 8192, n_obj=4)
    assert damaged.models_from_cache == 0
    assert os.path.getsize(files[0]) == len(raw)


def test_scene_does_not_depend_on_the_number_of_build_threads(standin_ply, monkeypatch):
    """The hierarchy build sorts the nodes of a level side by side and works out the instance boxes side by side
    (RT_HOST_THREADS, default: the machine's threads, at most 16), and a PLY file's vertex and face lines are parsed side by
    side; every emitted byte is the one-thread build's."""
    names = ("groups", "aabbs", "vertices", "triangles", "matrices", "data")
    built = []
    monkeypatch.setenv("RT_MESH_CACHE", "")              # parse the text: with more than one thread line by line, side by side
    monkeypatch.setenv("RT_PLY_PARALLEL_MIN", "0")       # (files this small would otherwise go through the tokenizer)
    for threads in ("1", "2", "7", "16"):
        monkeypatch.setenv("RT_HOST_THREADS", threads)
        built.append(S.build_scene([S.BUNNY, standin_ply], 96, 64, 8192, aabb_threshold=16))
    for sc in built[1:]:
        for n in names:
            assert np.ascontiguousarray(getattr(sc, n)).tobytes() == np.ascontiguousarray(getattr(built[0], n)).tobytes(), n


@pytest.mark.skipif(O.ref_model_lib() is None, reason="oracle/_ref/libref_model.so not built")
@pytest.mark.parametrize("tri_threshold,aabb_threshold", [(128, 8), (128, 16), (64, 16), (256, 256), (16, 2), (1000, 128)])
def test_other_hierarchy_shapes_match_the_reference_builder(standin_ply, tri_threshold, aabb_threshold):
    """The builder's two thresholds (main.cpp:270 passes 128 and 128) with other values -- the narrow hierarchies the
    benchmark's extras run among them -- against the reference's own subdivide/reserve/fill with the same values."""
    files = [S.BUNNY, standin_ply]
    sc = S.build_scene(files, 64, 64, 8192, n_obj=6, tri_threshold=tri_threshold, aabb_threshold=aabb_threshold)
    groups, aabbs, vertices, triangles = O.ref_model_build(files, sc.matrices, tri_threshold, aabb_threshold)
    assert len(groups) == len(sc.groups) and np.array_equal(triangles, sc.triangles)
    assert aabbs.tobytes() == sc.aabbs.tobytes()
    assert np.array_equal(vertices["pos"][:, :3].view(np.uint32), sc.vertices["pos"][:, :3].view(np.uint32))
    assert np.array_equal(vertices["norm"][:, :3].view(np.uint32), sc.vertices["norm"][:, :3].view(np.uint32))
    for gid in list(aabbs["group_id"]) + [int(sc.data["cam"]["root_group"][0])]:
        idx, kind = int(gid) & L.GROUP_ID_MASK, (int(gid) >> 28) & 0xF
        assert np.array_equal(groups["w"][idx][:3 if kind == L.SH_AABB else 4], sc.groups["w"][idx][:3 if kind == L.SH_AABB else 4]), idx


def glibc_random(seed, n):
    """The first n values of glibc's random() after srandom(seed) (its default generator: additive feedback over 31
    words, x[i] = x[i-3] + x[i-31], seeded by a Lehmer generator, 310 values discarded), written out independently."""
    r = [0] * (344 + n)
    r[0] = seed
    for i in range(1, 31):
        r[i] = (16807 * r[i - 1]) % 2147483647
    for i in range(31, 34):
        r[i] = r[i - 31]
    for i in range(34, 344 + n):
        r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
    return [x >> 1 for x in r[344:]]


def test_the_reference_scene_literally():
    """create_buffers() of the reference (main.cpp:268-343) at its own size: 256 instances from glibc's unseeded
    random(), instance i of model i & 1 with local id i, the two materials, the six leading groups, the camera."""
    sc = S.build_scene([S.BUNNY, S.BUNNY], 1024, 1024, 1 << 20)
    rnd = glibc_random(1, 4 * 256)
    assert rnd[:4] == [1804289383, 846930886, 1681692777, 1714636915]
    f = np.float32
    for i in range(256):
        a, tx, ty, tz = rnd[4 * i:4 * i + 4]
        alpha = 2 * 3.14159265359 * a / 2147483647
        m = sc.matrices[i]
        assert tuple(m["x"]) == (f(np.cos(alpha)), f(0), f(-f(np.sin(alpha))), f(4.0 * tx / 2147483647 - 2)), i
        assert tuple(m["y"]) == (f(np.sin(alpha)), f(0), f(np.cos(alpha)), f(4.0 * ty / 2147483647)), i
        assert tuple(m["z"]) == (f(0), f(1), f(0), f(2.0 * tz / 2147483647 - 1)), i
    cam = sc.data["cam"][0]
    assert tuple(cam["eye"][:3]) == (f(0), f(-0.3), f(0)) and tuple(cam["top_left"][:3]) == (f(-0.5), f(1), f(-0.5))
    assert tuple(cam["dx"][:3]) == (f(1.0 / 1024), f(0), f(0)) and tuple(cam["dy"][:3]) == (f(0), f(0), f(1.0 / 1024))
    root_id = 5 | L.TR_IDENTITY << 24 | L.SH_AABB << 28
    assert (int(cam["width"]), int(cam["height"]), int(cam["root_group"]), int(cam["root_local"])) == (1024, 1024, root_id, 0)
    assert int(sc.data["ray_count"][0]) == 1 << 20 and sc.group_count == 2560         # align(6 + 2 * 1033 + 1, 512)
    colours = sc.groups["w"][3:5].view(np.float32)
    assert colours.tolist() == [[f(0.2), f(0.9), f(0.2), f(0.1)], [f(0.9), f(0.2), f(0.2), f(0.1)]]   # green, red (main.cpp:319-323)
    assert tuple(sc.groups["w"][5][:3]) == (0, 256, 1)                                 # the scene root: boxes 0..255, f_local0
    boxes = sc.aabbs[:256]
    assert np.array_equal(boxes["local_id"], np.arange(256))
    roots = boxes["group_id"]
    assert len(set(roots[0::2])) == 1 and len(set(roots[1::2])) == 1 and roots[0] != roots[1]   # (i & 1 ? dragon : bunny)
    for r in (int(roots[0]), int(roots[1])):
        assert (r >> 24) & 0xF == L.TR_ORTHO and r >> 28 == L.SH_AABB
    leaf_material = {int(sc.groups["w"][int(g) & L.GROUP_ID_MASK][3]) for g in sc.aabbs["group_id"] if int(g) >> 28 == L.SH_MESH}
    assert leaf_material == {3 | L.SH_MATERIAL << 28, 4 | L.SH_MATERIAL << 28}         # tr_none | sh_material


def test_leaf_with_too_many_vertices_is_an_error_not_a_crash():
    """A leaf's triangles index its vertices with ten bits (shader.cl:125, model.cpp:122); the reference asserts that a
    leaf touches fewer than 1,024 (model.cpp:66).  Here a threshold that allows larger leaves fails the build cleanly."""
    with pytest.raises(RuntimeError, match="more than 1023 vertices"):
        S.build_scene([S.BUNNY], 64, 64, 8192, n_obj=2, tri_threshold=8192)
