"""The oracle's sort and queue compaction against independent numpy statements of their outputs, and
the small helpers (sub-pixel sequence, hit sorting network) against their definitions."""
import numpy as np
import pytest

from oracle import pyoracle as O
from raytracer_amd import layout as L

import helpers as H


@pytest.mark.parametrize("group_count", [512, 1536, 2560, 9728, 70144])
@pytest.mark.parametrize("count,n", [(8192, 8192), (8192, 0), (8192, 1), (8192, 511), (8192, 513), (8192, 8191),
                                     (16384, 16384), (16384, 8192), (16384, 8193), (24576, 20000), (24576, 16384 + 1537)])
def test_sort_output_definition(count, n, group_count):
    rng = np.random.default_rng(count * 7 + n + group_count)
    pairs = H.random_pairs(rng, count, group_count)
    got = O.sort_pairs(pairs, n, group_count)
    assert H.same_bytes(got, H.expected_sort(pairs, n, group_count))


def test_sort_is_a_permutation_and_sorted():
    rng = np.random.default_rng(3)
    pairs = H.random_pairs(rng, 32768, 9728)
    n = 30000
    got = O.sort_pairs(pairs, n, 9728)
    ids = got["key"][:n] & 0x7FFF
    assert np.all(np.diff(ids.astype(np.int64)) >= 0)
    assert np.array_equal(np.sort(got["slot"][:n]), np.sort(pairs["slot"][:n]))
    assert H.same_bytes(got[n:], pairs[n:])


def test_sort_few_distinct_keys_and_all_equal():
    rng = np.random.default_rng(4)
    for keys in ([5], [0, 1], [9727, 0, 5]):
        pairs = H.random_pairs(rng, 16384, 9728, with_high_bits=False)
        pairs["key"] = rng.choice(np.array(keys, dtype=np.uint32), 16384)
        assert H.same_bytes(O.sort_pairs(pairs, 16000, 9728), H.expected_sort(pairs, 16000, 9728))


@pytest.mark.parametrize("n", [1, 511, 512, 513, 1000, 8191, 8192, 8193, 20000])
def test_sort_block_index_is_a_bijection_per_tile(n):
    pos = np.arange(n)
    phys = H.sort_block_index_np(pos, n)
    assert np.array_equal(np.sort(phys), pos)
    for p in (0, n // 2, n - 1):
        assert O.sort_block_index(p, n) == phys[p]


def _sorted_case(rng, count, n, G, parked=True):
    pairs = H.random_pairs(rng, count, G - 1, with_high_bits=True)
    order = np.argsort(pairs["key"][:n] & L.GROUP_ID_MASK, kind="stable")
    pairs[:n] = pairs[:n][order]
    return pairs


@pytest.mark.parametrize("warp", [1, 32, 64])
def test_compaction_definition_first_step(warp):
    rng = np.random.default_rng(warp)
    G, count = 1536, 16384
    pairs = _sorted_case(rng, count, count, G)
    grp = np.zeros(G, L.group_data_dtype)
    grp["offset"][:, 1] = 0xFFFFFFFF
    glob = np.zeros(1, L.global_dtype)
    glob["group_count"], glob["ray_count"], glob["pixel_offset"] = G, count, count
    out, grp2, glob2 = O.compact_groups(pairs, grp, glob, warp)
    e_out, e_grp, e_glob, _ = H.expected_compaction(pairs, grp, glob, warp)
    assert H.same_bytes(out, e_out) and H.same_bytes(grp2, e_grp) and H.same_bytes(glob2, e_glob)
    n_new = int(glob2["ray_count"][0])
    assert n_new % warp == 0 and n_new + int(grp2["count"][:, 1].sum()) == count


def test_compaction_chain_keeps_invariants():
    """Several rounds: re-key the live part at random, sort, compact; parked tails persist."""
    rng = np.random.default_rng(11)
    G, count, warp = 2560, 32768, 32
    pairs = H.random_pairs(rng, count, G - 1)
    grp = np.zeros(G, L.group_data_dtype)
    grp["offset"][:, 1] = 0xFFFFFFFF
    glob = np.zeros(1, L.global_dtype)
    glob["group_count"], glob["ray_count"], glob["pixel_offset"] = G, count, count
    for step in range(6):
        n = int(glob["ray_count"][0])
        pairs["key"][:n] = rng.integers(0, G - 1, n, dtype=np.uint32)
        srt = O.sort_pairs(pairs, n, G)
        out, grp2, glob2 = O.compact_groups(srt, grp, glob, warp)
        e_out, e_grp, e_glob, dst = H.expected_compaction(srt, grp, glob, warp)
        assert H.same_bytes(out, e_out) and H.same_bytes(grp2, e_grp) and H.same_bytes(glob2, e_glob), step
        assert np.array_equal(np.sort(dst), np.arange(count))  # a permutation
        n_new = int(glob2["ray_count"][0])
        assert n_new % warp == 0 and n_new + int(grp2["count"][:, 1].sum()) == count
        ids = out["key"][:n_new] & L.GROUP_ID_MASK
        assert np.all(np.diff(ids.astype(np.int64)) >= 0)
        # every warp of the live part holds one group
        assert np.all(ids.reshape(-1, warp).min(axis=1) == ids.reshape(-1, warp).max(axis=1))
        pairs, grp, glob = out, grp2, glob2


def test_compaction_with_no_live_rays():
    G, count = 512, 8192
    pairs = np.zeros(count, L.index_pair_dtype)
    pairs["key"] = G - 1
    pairs["slot"] = np.arange(count)
    grp = np.zeros(G, L.group_data_dtype)
    grp["count"][G - 1, 1] = count
    grp["offset"][:, 1] = 0
    glob = np.zeros(1, L.global_dtype)
    glob["group_count"], glob["ray_count"], glob["old_count"] = G, 0, 0
    out, grp2, glob2 = O.compact_groups(pairs, grp, glob, 32)
    assert int(glob2["ray_count"][0]) == 0 and H.same_bytes(out, pairs)


def test_subpixel_sequence():
    vals = [O.subpixel(v) for v in range(15)]
    for v, got in enumerate(vals):
        k = v % 15 + 1
        rev = (k & 1) << 3 | (k & 2) << 1 | (k & 4) >> 1 | (k & 8) >> 3
        assert got == np.float32((rev - 0.5) / 15)          # the double-precision form gives the same float
        assert got == np.float32(np.float32(rev - 0.5) / np.float32(15))
    assert len(set(vals)) == 15 and all(0 < x < 1 for x in vals)
    assert O.subpixel(15) == vals[0]


def test_deinterleave():
    for val in (0, 1, 2, 3, 0xAAAAAAAA, 0x55555555, 0xDEADBEEF, 0xFFFFFFFF):
        x, y = O.deinterleave(val)
        ex = sum(((val >> (2 * b)) & 1) << b for b in range(16))
        ey = sum(((val >> (2 * b + 1)) & 1) << b for b in range(16))
        assert (x, y) == (ex, ey)


@pytest.mark.parametrize("n", list(range(0, 20)) + [31, 32, 33, 63, 64])
def test_sort_hits_network(n):
    rng = np.random.default_rng(n)
    hits = np.zeros(n, L.ray_hit_dtype)
    hits["pos"] = rng.choice(np.array([0.001, 0.5, 0.75, 1.25, 2.0], np.float32), n) if n % 2 else rng.random(n, dtype=np.float32)
    hits["group_id"] = np.arange(n)
    got = O.sort_hits(hits)
    assert np.all(np.diff(got["pos"]) >= 0)
    assert np.array_equal(np.sort(got["group_id"]), np.arange(n))
    # the network, written out directly
    a = list(zip(hits["pos"].tolist(), range(n)))
    base = 1
    while base < n:
        for i in range(base, n):
            if i & base:
                j = i - 2 * (i & (base - 1)) - 1
                if a[j][0] > a[i][0]:
                    a[j], a[i] = a[i], a[j]
        offs = base // 2
        while offs:
            for i in range(offs, n):
                if i & offs and a[i - offs][0] > a[i][0]:
                    a[i - offs], a[i] = a[i], a[i - offs]
            offs //= 2
        base *= 2
    assert [x[1] for x in a] == got["group_id"].tolist()


@pytest.mark.parametrize("group_count", [1, 2, 32, 33, 1024, 1025, 1 << 24])
def test_sort_at_the_ends_of_the_key_range(group_count):
    """One group (the pass loop of main.cpp:444-445 runs no pass at all), group counts either side of a digit boundary
    (32 | 33: one or two 5-bit passes; 1,024 | 1,025: two or three), and the largest id the 24 id bits hold."""
    rng = np.random.default_rng(group_count)
    pairs = H.random_pairs(rng, 16384, group_count)
    if group_count == 1 << 24:
        pairs["key"][:64] = (pairs["key"][:64] & np.uint32(0xFF000000)) | np.uint32((1 << 24) - 1)   # the top id is there
    for n in (16384, 8193):
        got = O.sort_pairs(pairs, n, group_count)
        assert H.same_bytes(got, H.expected_sort(pairs, n, group_count)), n
    if group_count == 1:
        assert H.same_bytes(O.sort_pairs(pairs, 16384, 1), pairs)        # nothing moves
