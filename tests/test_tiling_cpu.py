"""The N > 1 path on the CPU: two gloo ranks each render a row band with the oracle, all-gather the
bands and rank 0 compares the assembled frame with a single full-frame render."""
import os
import socket
import sys

import numpy as np
import pytest

from raytracer_amd import tiling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_partition_covers_the_frame():
    for height, world in ((1080, 8), (1080, 7), (64, 2), (5, 8), (2160, 4)):
        rows = [tiling.band_of(r, world, height) for r in range(world)]
        assert rows[0][0] == 0 and sum(n for _, n in rows) == height
        for (a, n), (b, _) in zip(rows, rows[1:]):
            assert a + n == b
        assert max(n for _, n in rows) == tiling.max_band_rows(world, height)
        assert max(n for _, n in rows) - min(n for _, n in rows) <= 1


def _rank_main(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as O
    from raytracer_amd import scene as S, tiling as T

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 64, 50  # 25 rows each
    first, rows = T.band_of(rank, world, H)
    sc = S.build_scene([S.BUNNY, S.BUNNY], W, H, 8192, band_first=first, band_rows=rows)
    o = O.Oracle(sc, warp_width=32, sample_limit=W * rows)
    o.init_frame()
    o.step(12)
    o.set_warp_width(1)
    o.step(1)
    while int(o.global_data["ray_count"][0]):
        o.step(1)
    band = torch.zeros((T.max_band_rows(world, H), W, 4), dtype=torch.uint8)
    band[:rows] = torch.from_numpy(o.resolve_rgba8())
    frame = T.gather_frame(dist, band, world, H)
    counts = torch.tensor([float(o.counters()["primary_spawned"]), float(o.counters()["shadow_spawned"])], dtype=torch.float64)
    dist.all_reduce(counts)
    if rank == 0:
        np.savez(result_path, frame=frame.numpy(), counts=counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_full_frame(tmp_path):
    import torch.multiprocessing as mp
    from oracle import pyoracle as O
    from raytracer_amd import scene as S

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    result = str(tmp_path / "frame.npz")
    mp.spawn(_rank_main, args=(2, port, result), nprocs=2, join=True)
    got = np.load(result)

    W, H = 64, 50
    sc = S.build_scene([S.BUNNY, S.BUNNY], W, H, 8192)
    o = O.Oracle(sc, warp_width=32, sample_limit=W * H)
    o.init_frame()
    o.step(12)
    o.set_warp_width(1)
    o.step(1)
    while int(o.global_data["ray_count"][0]):
        o.step(1)
    want = o.resolve_rgba8()
    assert got["frame"].shape == want.shape
    assert np.abs(got["frame"].astype(int) - want.astype(int)).max() <= 1   # one sample per pixel: same samples
    assert got["counts"][0] == W * H == float(o.counters()["primary_spawned"])
    assert got["counts"][1] == float(o.counters()["shadow_spawned"])
