"""Screen-tile data parallelism: a frame is cut into row bands, one per rank; each rank renders its
band with the scene replicated, and the resolved bands are gathered into the full frame.

A band is the full camera with fewer rows and a row offset (ray-tracer.cl:59-71 uses only top_left, dx,
dy, width and height; the per-pixel sample sequence depends only on the sample number), so a pixel
gets the same samples whichever rank owns it.  The gather is one collective per frame on the RGBA8
image: all_gather over RCCL on GPUs ("nccl" backend), over gloo in the CPU tests."""
import numpy as np


def band_of(rank, world, height):
    """(first row, row count) of `rank`: rows are dealt as evenly as possible, earlier ranks first."""
    base, extra = divmod(height, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def max_band_rows(world, height):
    return -(-height // world)


def assemble(gathered, world, height):
    """Full frame from the all-gathered, equally padded bands: gathered[r, :rows_r] are rank r's rows."""
    parts = []
    for r in range(world):
        _, rows = band_of(r, world, height)
        parts.append(gathered[r, :rows])
    if hasattr(gathered, "cpu"):
        import torch
        return torch.cat(parts, dim=0)
    return np.concatenate(parts, axis=0)


def gather_frame(dist, band, world, height):
    """all_gather of one padded band tensor per rank (torch tensors, any backend) -> full frame."""
    import torch
    out = torch.empty((world,) + tuple(band.shape), dtype=band.dtype, device=band.device)
    dist.all_gather(list(out.unbind(0)), band.contiguous())
    return assemble(out, world, height)
