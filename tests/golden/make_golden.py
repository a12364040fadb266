#!/usr/bin/env python3
"""Generates the fixtures in this directory.  Run from the repository root after `make`:

    python tests/golden/make_golden.py

scene_bunny_digests.json  digests of the scene arrays emitted by the REFERENCE's model.cpp (built as
                          oracle/_ref/libref_model.so from /root/reference; needs that tree present)
sort_cases.npz            inputs and outputs of the ray-key sort from the oracle
pipeline_bunny128.npz     state after 40 steps of the reference pipeline on the 128x128 bunny scene
drain_bunny64.npz         a finished 2-samples-per-pixel frame (accumulator, image, hit records)
pipeline_bunny128_mad.npz, drain_bunny64_mad.npz, drain_bunny64_mirror2_mad.npz
                          the same kinds of vector in the fused arithmetic (RT_FLAG_ARITH_MAD): state after 48 steps, a
                          finished 1-sample-per-pixel frame with the reference's shading and with two mirror bounces

The oracle is a restatement of the reference kernels (parity unpinned, see oracle/rt_oracle.h); these
files pin the oracle against regressions and give the GPU tests vectors that do not need the oracle.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as O  # noqa: E402
from raytracer_amd import layout as L, scene as S  # noqa: E402
import helpers as H  # noqa: E402
from test_host_scene import scene_digests  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    # 1. scene arrays from the reference's own builder
    if O.ref_model_lib() is None:
        print("oracle/_ref/libref_model.so missing: scene digests not regenerated")
    else:
        mine = S.build_scene([S.BUNNY, S.BUNNY], 128, 128, 16384)
        groups, aabbs, vertices, triangles = O.ref_model_build([S.BUNNY, S.BUNNY], mine.matrices)
        groups["w"][3:5] = mine.groups["w"][3:5]  # the two material colours are set by main.cpp:319-323, not by model.cpp
        ref_scene = S.Scene(mine.data, groups, mine.matrices, aabbs, vertices, triangles, mine.pool_size, 0, 0)
        json.dump({"bunny_x2": scene_digests(ref_scene)}, open(os.path.join(HERE, "scene_bunny_digests.json"), "w"), indent=1)

    # 2. sort cases
    rng = np.random.default_rng(2024)
    cases = {}
    for name, (count, n, G) in {"full": (16384, 16384, 9728), "ragged": (16384, 8192 + 1537, 2560), "small": (8192, 700, 1536)}.items():
        pairs = H.random_pairs(rng, count, G)
        cases[name + "_in"] = pairs
        cases[name + "_out"] = O.sort_pairs(pairs, n, G)
        cases[name + "_args"] = np.array([count, n, G], np.uint32)
    np.savez_compressed(os.path.join(HERE, "sort_cases.npz"), **cases)

    # 3. pipeline state
    sc = S.build_scene([S.BUNNY, S.BUNNY], 128, 128, 16384)
    o = O.Oracle(sc, warp_width=32, hit_capacity=16384 * 4)
    o.init_frame()
    o.step(40)
    rl = o.ray_list
    np.savez_compressed(os.path.join(HERE, "pipeline_bunny128.npz"),
                        steps=np.uint32(40), warp_width=np.uint32(32),
                        global_data=o.global_data.copy(), ray_index=o.ray_index.copy(), grp_data=o.grp_data.copy(),
                        area=o.area.copy(), hits=o.hits.copy(), counters=np.array([o.counters()], L.counters_dtype),
                        ray_field_sha=np.array([[name, sha(rl[name])] for name in L.ray_queue_dtype.names]))
    o.close()

    # 4. a finished frame: 2 samples per pixel, then drain with granularity 1
    sc = S.build_scene([S.BUNNY, S.BUNNY], 64, 64, 8192)
    limit = 64 * 64 * 2
    o = O.Oracle(sc, warp_width=32, sample_limit=limit, hit_capacity=limit)
    o.init_frame()
    o.step(24)
    o.set_warp_width(1)   # the next compaction releases every parked tail
    o.step(1)
    steps = 25
    while int(o.global_data["ray_count"][0]):
        o.step(1)
        steps += 1
    np.savez_compressed(os.path.join(HERE, "drain_bunny64.npz"), steps=np.uint32(steps), sample_limit=np.uint32(limit),
                        area=o.area.copy(), image=o.resolve_rgba8(), hits=o.hits.copy(),
                        counters=np.array([o.counters()], L.counters_dtype))
    print("drained after", steps, "steps;", o.counters())
    o.close()

    # 5. the same two kinds of vector in the fused arithmetic (RT_FLAG_ARITH_MAD), with the parameters of the GPU tests
    #    that compare the HIP path with the oracle of that mode (tests/test_arith_modes.py)
    sc = S.build_scene([S.BUNNY, S.BUNNY], 128, 128, 16384)
    o = O.Oracle(sc, warp_width=32, hit_capacity=16384 * 8, mad=True)
    o.init_frame()
    o.step(48)
    rl = o.ray_list
    np.savez_compressed(os.path.join(HERE, "pipeline_bunny128_mad.npz"),
                        steps=np.uint32(48), warp_width=np.uint32(32),
                        global_data=o.global_data.copy(), ray_index=o.ray_index.copy(), grp_data=o.grp_data.copy(),
                        area=o.area.copy(), hits=o.hits.copy(), counters=np.array([o.counters()], L.counters_dtype),
                        ray_field_sha=np.array([[name, sha(rl[name])] for name in L.ray_queue_dtype.names]))
    o.close()
    sc = S.build_scene([S.BUNNY, S.BUNNY], 64, 64, 8192)
    limit = 64 * 64
    for name, shading in (("drain_bunny64_mad.npz", 0), ("drain_bunny64_mirror2_mad.npz", 1 << 8 | 2 << 12 | 0x5EED << 16)):
        o = O.Oracle(sc, warp_width=32, sample_limit=limit, hit_capacity=limit, shading=shading, mad=True)
        o.init_frame()
        o.step(24)
        o.set_warp_width(1)
        o.step(1)
        steps = 25
        while int(o.global_data["ray_count"][0]):
            o.step(1)
            steps += 1
        np.savez_compressed(os.path.join(HERE, name), steps=np.uint32(steps), sample_limit=np.uint32(limit), shading=np.uint32(shading),
                            area=o.area.copy(), image=o.resolve_rgba8(), hits=o.hits.copy(),
                            counters=np.array([o.counters()], L.counters_dtype))
        print(name, "drained after", steps, "steps;", o.counters())
        o.close()


if __name__ == "__main__":
    main()
