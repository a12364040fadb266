"""rt_sort_pairs_static, the sort entry point that allocates nothing (its workspace is a fixed device
array inside the library), through the C ABI: its output must be bit-identical to rt_sort_pairs' on
the same device buffers, on two calls in a row, the second with a smaller input."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import api, layout as L, scene as S

import helpers as H

pytestmark = pytest.mark.gpu


def _device_pairs(torch, pairs):
    return torch.from_numpy(np.ascontiguousarray(pairs).view(np.int64).copy()).cuda()


def _host_pairs(t):
    return t.cpu().numpy().view(L.index_pair_dtype)


def test_static_sort_equals_the_allocating_sort_twice_in_a_row():
    import torch
    lib = api.hip_lib()
    sc = S.build_scene([S.BUNNY], 256, 128, 131072, n_obj=4)
    rt = api.RayTracer(sc)                      # only for the reference call; the static call takes no context
    rng = np.random.default_rng(2024)
    for count, n, groups in ((131072, 131072 - 96, 9728), (24576, 20001, 1536)):   # second call: smaller input
        pairs = H.random_pairs(rng, count, groups)
        d_in = _device_pairs(torch, pairs)
        d_want = torch.zeros_like(d_in)
        d_got = torch.full_like(d_in, -1)
        torch.cuda.synchronize()
        rt.sort_pairs_device(d_in.data_ptr(), d_want.data_ptr(), count, n, groups)
        rc = lib.rt_sort_pairs_static(C.c_void_p(d_in.data_ptr()), C.c_void_p(d_got.data_ptr()), count, n, groups, None)
        assert rc == 0
        torch.cuda.synchronize()
        want, got = _host_pairs(d_want), _host_pairs(d_got)
        assert H.same_bytes(want, H.expected_sort(pairs, n, groups))
        assert H.same_bytes(got, want), (count, n, groups)
    rt.close()


def test_static_sort_rejects_what_it_cannot_hold():
    import torch
    lib = api.hip_lib()
    limit = 1048576
    buf = torch.zeros(2 * (limit + 8192), dtype=torch.int64, device="cuda")
    a, b = buf.data_ptr(), buf.data_ptr() + 8 * (limit + 8192)
    vp = C.c_void_p
    assert lib.rt_sort_pairs_static(vp(a), vp(b), limit + 8192, 10, 512, None) == 6      # RT_ERR_UNSUPPORTED: beyond the workspace
    assert lib.rt_sort_pairs_static(vp(a), vp(b), 8192 + 1, 10, 512, None) == 1          # not whole tiles
    assert lib.rt_sort_pairs_static(vp(a), vp(b), 8192, 8193, 512, None) == 1            # n > count
    assert lib.rt_sort_pairs_static(vp(a), vp(a + 64), 8192, 10, 512, None) == 1         # overlap
    assert lib.rt_sort_pairs_static(None, vp(b), 8192, 10, 512, None) == 1
    # the largest size it takes
    rng = np.random.default_rng(3)
    pairs = H.random_pairs(rng, limit, 9728)
    d_in = _device_pairs(torch, pairs)
    d_out = torch.zeros_like(d_in)
    assert lib.rt_sort_pairs_static(vp(d_in.data_ptr()), vp(d_out.data_ptr()), limit, limit - 5, 9728, None) == 0
    torch.cuda.synchronize()
    assert H.same_bytes(_host_pairs(d_out), H.expected_sort(pairs, limit - 5, 9728))
