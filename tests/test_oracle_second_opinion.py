"""The C oracle against a second statement of `process` written independently from the reference's text
(tests/pyprocess.py, numpy float32, one ray at a time): at several points of a run a sample of slots of every kind
(starting samples, sky, light, material, box groups, leaves) is advanced one step by both, and every field of the
slot and its new sort key must agree bit for bit.  Both arithmetic modes; no GPU."""
import numpy as np
import pytest

from oracle import pyoracle as O
from raytracer_amd import layout as L, scene as S

import pyprocess as P


def advance_and_compare(o, sc, per_kind, rng, seen):
    g = o.global_data[0]
    n, pixel_offset = int(g["ray_count"]), int(g["pixel_offset"])
    index = o.ray_index[:n].copy()
    kinds = (index["key"] >> L.GROUP_SH_SHIFT) & 0xF
    chosen = []
    for kind in range(6):
        where = np.flatnonzero(kinds == kind)
        if len(where):
            chosen.extend(rng.choice(where, min(per_kind[kind], len(where)), replace=False).tolist())
    before = o.ray_list[index["slot"][chosen]].copy()
    o.step(1)
    after_rays, after_index = o.ray_list, o.ray_index
    key_of_slot = np.zeros(o.pool, np.uint32)
    key_of_slot[after_index["slot"]] = after_index["key"]
    for ray, position in zip(before, chosen):
        key, slot = int(index["key"][position]), int(index["slot"][position])
        mine = ray.copy()
        new_key, tri = P.process(mine, key, position, sc, pixel_offset)
        kind = (key >> L.GROUP_SH_SHIFT) & 0xF
        seen[kind] = seen.get(kind, 0) + 1
        assert new_key == int(key_of_slot[slot]), (kind, hex(key), position)
        got = after_rays[slot]
        for name in L.ray_queue_dtype.names:
            assert np.ascontiguousarray(mine[name]).tobytes() == np.ascontiguousarray(got[name]).tobytes(), (kind, hex(key), name, mine[name], got[name])
        if tri is not None:
            seen["hits"] = seen.get("hits", 0) + 1
            if int(ray["type"]) == 0:      # a camera ray's hit: the slot's reach is the hit's distance, `orig` names the leaf
                assert got["dir_max"][3] == got["orig"]["pos"] and int(got["orig"]["group_id"]) == key


@pytest.mark.parametrize("mad", [False, True])
@pytest.mark.parametrize("files,warp", [("bunny", 4), ("bunny+standin", 8)])
def test_process_step_by_step_against_the_second_statement(standin_ply, files, warp, mad):
    models = [S.BUNNY, S.BUNNY] if files == "bunny" else [S.BUNNY, standin_ply]
    sc = S.build_scene(models, 64, 64, 8192)
    P.set_mode(mad)
    o = O.Oracle(sc, warp_width=warp, mad=mad)
    o.init_frame()
    # init_rays: every slot against init_ray of the second statement (ray-tracer.cl:56-82)
    rays = o.ray_list
    cam = sc.data["cam"][0]
    for slot in (0, 1, 63, 64, 4095, 4096, 8191):
        mine = np.zeros(1, L.ray_queue_dtype)[0]
        key = P.init_ray(cam, mine, slot)
        assert key == int(o.ray_index["key"][slot]) == int(cam["root_group"])
        for name in L.ray_queue_dtype.names:
            assert np.ascontiguousarray(mine[name]).tobytes() == np.ascontiguousarray(rays[slot][name]).tobytes(), (slot, name)
    rng = np.random.default_rng(64)
    seen = {}
    per_kind = {L.SH_SPAWN: 6, L.SH_SKY: 6, L.SH_LIGHT: 6, L.SH_MATERIAL: 10, L.SH_AABB: 14, L.SH_MESH: 14}
    for step in range(36):
        if step % 3 == 2:
            advance_and_compare(o, sc, per_kind, rng, seen)
        else:
            o.step(1)
    o.close()
    for kind in range(6):
        assert seen.get(kind, 0) >= 6, (kind, seen)     # every branch of the dispatcher was compared
    assert seen[L.SH_AABB] >= 100 and seen[L.SH_MESH] >= 100 and seen.get("hits", 0) >= 10
