"""Sort and queue-compaction stages on the GPU through the C ABI: bit-exact against the oracle and
against the committed vectors; at full pool size against the definition of the output."""
import os

import numpy as np
import pytest

from raytracer_amd import api, layout as L, scene as S

import helpers as H

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    sc = S.build_scene([S.BUNNY], 256, 128, 32768, n_obj=4)
    rt = api.RayTracer(sc)
    yield rt
    rt.close()


@pytest.fixture(scope="module")
def big_ctx():
    sc = S.build_scene([S.BUNNY], 1920, 1080, 1920 * 1080, n_obj=4)
    rt = api.RayTracer(sc)
    yield rt
    rt.close()


def test_sort_golden_vectors(ctx):
    g = np.load(os.path.join(GOLDEN, "sort_cases.npz"))
    for name in ("full", "ragged", "small"):
        count, n, G = (int(x) for x in g[name + "_args"])
        assert H.same_bytes(ctx.sort_pairs(g[name + "_in"], n, G), g[name + "_out"]), name


@pytest.mark.parametrize("group_count", [512, 1536, 9728, 70144])
@pytest.mark.parametrize("count,n", [(8192, 8192), (8192, 0), (8192, 1), (8192, 511), (8192, 513), (8192, 8191),
                                     (16384, 8193), (24576, 20000), (32768, 32768), (32768, 32736)])
def test_sort_matches_oracle(ctx, count, n, group_count):
    from oracle import pyoracle as O
    rng = np.random.default_rng(count + 3 * n + group_count)
    pairs = H.random_pairs(rng, count, group_count)
    assert H.same_bytes(ctx.sort_pairs(pairs, n, group_count), O.sort_pairs(pairs, n, group_count))


def test_sort_skewed_keys(ctx):
    rng = np.random.default_rng(9)
    for keys in ([5], [0, 1], [9727, 0, 5, 6, 1030]):
        pairs = H.random_pairs(rng, 32768, 9728, with_high_bits=False)
        pairs["key"] = rng.choice(np.array(keys, dtype=np.uint32), 32768)
        assert H.same_bytes(ctx.sort_pairs(pairs, 32000, 9728), H.expected_sort(pairs, 32000, 9728))


@pytest.mark.parametrize("group_count", [2, 512, 9728, 70144])
def test_plain_order_sort_of_the_free_schedule(group_count):
    """A context with RT_FLAG_FREE_SCHEDULE sorts with small tiles and keeps equal group ids in their
    plain order: exactly numpy's stable sort by group id."""
    sc = S.build_scene([S.BUNNY], 1920, 1080, 1920 * 1080, n_obj=4)
    rt = api.RayTracer(sc, free_schedule=True)
    rng = np.random.default_rng(group_count)
    for count, n in ((8192, 8192), (8192, 1), (8192, 0), (16384, 8193), (32768, 32736), (rt.pool, rt.pool - 64 * 999 - 7)):
        pairs = H.random_pairs(rng, count, group_count)
        want = pairs.copy()
        order = np.argsort(pairs["key"][:n] & L.GROUP_ID_MASK, kind="stable")
        want[:n] = pairs[:n][order]
        assert H.same_bytes(rt.sort_pairs(pairs, n, group_count), want), (count, n)
    rt.close()


def test_sort_rejects_bad_arguments(ctx):
    pairs = np.zeros(8192, L.index_pair_dtype)
    with pytest.raises(api.RtError, match="RT_ERR_INVALID_VALUE"):
        ctx.sort_pairs(pairs[:100], 100, 512)            # not a multiple of the tile
    with pytest.raises(api.RtError, match="RT_ERR_INVALID_VALUE"):
        ctx.sort_pairs(pairs, 9000, 512)                 # n > count
    with pytest.raises(api.RtError, match="RT_ERR_INVALID_VALUE"):
        ctx.sort_pairs(np.zeros(65536, L.index_pair_dtype), 10, 512)  # beyond the pool


def test_sort_full_pool_size(big_ctx):
    """2,080,768 pairs, 9,728 groups: the output is exactly the stable sort of the thread-major reading."""
    rng = np.random.default_rng(77)
    count = big_ctx.pool
    for n in (count, count - 64 * 1000 - 32):
        pairs = H.random_pairs(rng, count, 9728)
        got = big_ctx.sort_pairs(pairs, n, 9728)
        assert H.same_bytes(got, H.expected_sort(pairs, n, 9728))


@pytest.mark.parametrize("warp", [1, 32, 64])
def test_compaction_matches_oracle_over_several_rounds(ctx, warp):
    from oracle import pyoracle as O
    rng = np.random.default_rng(100 + warp)
    G, count = 2560, 32768
    pairs = H.random_pairs(rng, count, G - 1)
    grp = np.zeros(G, L.group_data_dtype)
    grp["offset"][:, 1] = 0xFFFFFFFF
    glob = np.zeros(1, L.global_dtype)
    glob["group_count"], glob["ray_count"], glob["pixel_offset"] = G, count, count
    for step in range(5):
        n = int(glob["ray_count"][0])
        pairs["key"][:n] = rng.integers(0, G - 1, n, dtype=np.uint32) | (rng.integers(0, 16, n, dtype=np.uint32) << np.uint32(28))
        srt = O.sort_pairs(pairs, n, G)
        want = O.compact_groups(srt, grp, glob, warp)
        got = ctx.compact_groups(srt, grp, glob, warp)
        for a, b in zip(got, want):
            assert H.same_bytes(a, b), step
        pairs, grp, glob = want


def test_compaction_with_no_live_rays(ctx):
    G, count = 512, 8192
    pairs = np.zeros(count, L.index_pair_dtype)
    pairs["key"] = G - 1
    pairs["slot"] = np.arange(count)
    grp = np.zeros(G, L.group_data_dtype)
    grp["count"][G - 1, 1] = count
    glob = np.zeros(1, L.global_dtype)
    glob["group_count"] = G
    out, grp2, glob2 = ctx.compact_groups(pairs, grp, glob, 32)
    assert int(glob2["ray_count"][0]) == 0 and H.same_bytes(out, pairs)


def test_compaction_full_size_definition(big_ctx):
    rng = np.random.default_rng(5)
    G, count = 9728, big_ctx.pool
    pairs = H.random_pairs(rng, count, G - 1)
    n = count - 4096
    order = np.argsort(pairs["key"][:n] & L.GROUP_ID_MASK, kind="stable")
    pairs[:n] = pairs[:n][order]
    grp = np.zeros(G, L.group_data_dtype)
    # previously parked tails: lay them out after position n by group, as a real frame would
    ids_tail = np.sort(pairs["key"][n:] & L.GROUP_ID_MASK)
    pairs["key"][n:] = ids_tail
    tails = np.bincount(ids_tail, minlength=G).astype(np.uint32)
    grp["count"][:, 1] = tails
    grp["offset"][:, 1] = n + np.concatenate([[0], np.cumsum(tails)[:-1]]).astype(np.uint32)
    glob = np.zeros(1, L.global_dtype)
    glob["group_count"], glob["ray_count"], glob["old_count"] = G, n, n
    got = big_ctx.compact_groups(pairs, grp, glob, 64)
    want = H.expected_compaction(pairs, grp, glob, 64)
    for a, b in zip(got, want[:3]):
        assert H.same_bytes(a, b)
