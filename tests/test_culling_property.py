"""The property the box kernel's wave-level culling rests on (raytracer_amd/csrc: build_clusters, visit_aabb): a ray
that enters a box enters every bound that encloses the box strictly -- with the slab test's own float arithmetic,
rounding and not-a-number cases included, in both arithmetic modes.  So skipping the boxes of a cluster whose bound
no lane of a wave enters cannot change any lane's candidates.  Checked here on millions of random and deliberately
degenerate cases in numpy (strict mode) and on tens of thousands with an exact fused multiply-add (fused mode); the
bound is built as at upload: the members' extremes moved one float step outwards.  No GPU."""
import numpy as np

import pyprocess as P

f32 = np.float32


def slab_strict(lo, hi, start, inv, ray_min, ray_max):
    """aabb_shader's test (shader.cl:86-91) as the kernel evaluates it: accepted iff t_max > max(t_min, ray.min) and
    t_min < ray.max.  lo/hi/start/inv: (..., 3) float32."""
    with np.errstate(invalid="ignore", over="ignore"):
        p1 = ((lo - start).astype(f32) * inv).astype(f32)
        p2 = ((hi - start).astype(f32) * inv).astype(f32)
    return accept(p1, p2, ray_min, ray_max)


def accept(p1, p2, ray_min, ray_max):
    near, far = np.fmin(p1, p2), np.fmax(p1, p2)
    t_min = np.fmax(np.fmax(near[..., 0], near[..., 1]), near[..., 2])
    t_max = np.fmin(np.fmin(far[..., 0], far[..., 1]), far[..., 2])
    with np.errstate(invalid="ignore"):
        return (t_max > np.fmax(t_min, ray_min)) & (t_min < ray_max)


def reference_accept(lo, hi, start, inv, ray_min, ray_max):
    """The reference's three comparisons (shader.cl:91), for the equivalence the kernel's form relies on."""
    with np.errstate(invalid="ignore", over="ignore"):
        p1 = ((lo - start).astype(f32) * inv).astype(f32)
        p2 = ((hi - start).astype(f32) * inv).astype(f32)
        near, far = np.fmin(p1, p2), np.fmax(p1, p2)
        t_min = np.fmax(np.fmax(near[..., 0], near[..., 1]), near[..., 2])
        t_max = np.fmin(np.fmin(far[..., 0], far[..., 1]), far[..., 2])
        return (t_max > t_min) & (t_max > ray_min) & (t_min < ray_max)


def cases(rng, n, members):
    """n clusters of `members` boxes each with one ray per cluster, many of them degenerate on purpose."""
    grid = rng.integers(-8, 9, (n, members, 2, 3)).astype(f32) * f32(0.25)       # coordinates on a coarse grid: shared faces, flat boxes
    free = rng.uniform(-2, 2, (n, members, 2, 3)).astype(f32)
    corners = np.where(rng.random((n, members, 1, 1)) < 0.5, grid, free)
    lo, hi = corners.min(axis=2), corners.max(axis=2)
    start = rng.uniform(-2.5, 2.5, (n, 3)).astype(f32)
    snap = rng.random((n, 3)) < 0.3                                               # start exactly on some box's face
    start = np.where(snap, lo[np.arange(n), rng.integers(0, members, n)], start).astype(f32)
    d = rng.normal(size=(n, 3)).astype(f32)
    kind = rng.random((n, 3))
    d = np.where(kind < 0.15, f32(0), d)                                          # parallel to a pair of faces
    d = np.where((kind >= 0.15) & (kind < 0.2), f32(-0.0), d)
    d = np.where((kind >= 0.2) & (kind < 0.25), f32(1e-30), d).astype(f32)
    with np.errstate(divide="ignore"):
        inv = (f32(1) / d).astype(f32)
    ray_min = np.where(rng.random(n) < 0.5, f32(0.001), rng.uniform(0, 3, n).astype(f32)).astype(f32)
    ray_max = np.where(rng.random(n) < 0.5, f32(np.inf), rng.uniform(0, 6, n).astype(f32)).astype(f32)
    bound_lo = np.nextafter(lo.min(axis=1), f32(-np.inf)).astype(f32)             # cluster_bounds: one float step outwards
    bound_hi = np.nextafter(hi.max(axis=1), f32(np.inf)).astype(f32)
    return lo, hi, bound_lo, bound_hi, start, inv, ray_min, ray_max


def test_entering_a_box_means_entering_its_clusters_bound_strict_arithmetic():
    rng = np.random.default_rng(8)
    total = entered = 0
    for members in (1, 2, 5, 8):
        lo, hi, blo, bhi, start, inv, ray_min, ray_max = cases(rng, 400_000, members)
        in_bound = slab_strict(blo, bhi, start, inv, ray_min, ray_max)
        for k in range(members):
            in_box = slab_strict(lo[:, k], hi[:, k], start, inv, ray_min, ray_max)
            assert not np.any(in_box & ~in_bound), members
            # the kernel's two comparisons are the reference's three (for a ray.min that is a number)
            assert np.array_equal(in_box, reference_accept(lo[:, k], hi[:, k], start, inv, ray_min, ray_max))
            total += len(in_box)
            entered += int(in_box.sum())
    assert total > 6_000_000 and entered > 200_000          # the property was not vacuous


def test_a_bound_that_merely_touches_is_not_enough():
    """Why the step outwards: a ray in the plane of a box's face, parallel to it, enters the box (0 * inf is not a
    number, and the minimum of a number and not-a-number is the number: no constraint from that face) but not a bound
    whose face lies in the same plane and whose other faces are farther out... unless that bound is also touched.  The
    case that breaks a touching bound: the box's face at the start, the bound's equal face, the bound's far face beyond."""
    start = np.array([[0.5, 0, 0]], f32)
    d = np.array([[0.0, 1, 0]], f32)
    with np.errstate(divide="ignore"):
        inv = (f32(1) / d).astype(f32)
    lo, hi = np.array([[0.5, 1, -1]], f32), np.array([[0.5, 2, 1]], f32)          # a flat box in the plane x = 0.5
    assert slab_strict(lo, hi, start, inv, f32(0.001), f32(np.inf))[0]
    touching = slab_strict(lo, np.array([[0.75, 2, 1]], f32), start, inv, f32(0.001), f32(np.inf))[0]
    stepped = slab_strict(np.nextafter(lo, f32(-np.inf)).astype(f32), np.nextafter(np.array([[0.75, 2, 1]], f32), f32(np.inf)).astype(f32),
                          start, inv, f32(0.001), f32(np.inf))[0]
    assert stepped and not touching


def fused_products(corner, inv, neg):
    out = np.empty_like(corner)
    for i in range(corner.shape[0]):
        for k in range(3):
            out[i, k] = P.fma(corner[i, k], inv[i, k], neg[i, k])
    return out


def test_entering_a_box_means_entering_its_clusters_bound_fused_arithmetic():
    """(corner - start) * inv_dir as fma(corner, inv_dir, -(start * inv_dir)), the second product once per ray: the same
    property for every ray whose direction has no component that is exactly zero (rounding is monotonic, so the fused
    form is monotonic in the corner as long as 1 / dir is a number)."""
    rng = np.random.default_rng(9)
    lo, hi, blo, bhi, start, inv, ray_min, ray_max = cases(rng, 8_000, 4)
    regular = np.isfinite(inv).all(axis=1)
    lo, hi, blo, bhi, start, inv, ray_min, ray_max = (a[regular] for a in (lo, hi, blo, bhi, start, inv, ray_min, ray_max))
    assert len(inv) > 3_000
    neg = (-(start * inv).astype(f32)).astype(f32)
    in_bound = accept(fused_products(blo, inv, neg), fused_products(bhi, inv, neg), ray_min, ray_max)
    entered = 0
    for k in range(4):
        in_box = accept(fused_products(lo[:, k], inv, neg), fused_products(hi[:, k], inv, neg), ray_min, ray_max)
        assert not np.any(in_box & ~in_bound)
        entered += int(in_box.sum())
    assert entered > 300


def test_fused_slab_test_and_a_direction_component_that_is_exactly_zero():
    """A LIMIT of the fused mode, written down as a test so that it stays known: with dir.z == 0 the reciprocal is
    infinite and fma(corner, inf, -(start * inf)) is not a number for corners on one side of zero and infinite for those
    on the other, whatever the start.  The fused test then (a) ignores the axis for a box whose z corners are both
    positive even when the ray passes beside it, (b) rejects a box with z corners either side of zero that the ray runs
    through -- in the oracle and in the HIP path alike, box for box -- and (c) can reject a cluster's bound while
    accepting a box inside it, so the culled scan and the full scan may differ for such a ray.  The strict mode has none
    of this.  Camera, shadow and bounce rays never have an exactly zero component (a sub-pixel offset is never a whole
    number; an instance rotation would have to cancel exactly); an axis-parallel ray handed to rt_trace_closest does:
    trace those on a strict context."""
    start = np.array([[0.1, -1.0, 0.5]], f32)
    d = np.array([[0.0, 1.0, 0.0]], f32)                     # along +y: x and z components exactly zero
    with np.errstate(divide="ignore"):
        inv = (f32(1) / d).astype(f32)
    with np.errstate(invalid="ignore"):
        neg = (-(start * inv).astype(f32)).astype(f32)
    everywhere = (f32(0.001), f32(np.inf))

    def fused(lo, hi):
        return bool(accept(fused_products(np.array([lo], f32), inv, neg), fused_products(np.array([hi], f32), inv, neg), *everywhere)[0])

    def strict(lo, hi):
        return bool(slab_strict(np.array([lo], f32), np.array([hi], f32), start, inv, *everywhere)[0])

    through = ([0.05, 1, 0.25], [0.2, 2, 1.0])               # the ray runs through it; all corners positive
    beside = ([0.05, 1, 0.75], [0.2, 2, 1.0])                # the ray passes below it (z = 0.5 < 0.75)
    straddling = ([0.05, 1, -0.5], [0.2, 2, 1.5])            # the ray runs through it; z corners either side of zero
    assert strict(*through) and not strict(*beside) and strict(*straddling)
    assert fused(*through)
    assert fused(*beside)                                     # (a)
    assert not fused(*straddling)                             # (b), and with `through` inside `straddling`: (c)
