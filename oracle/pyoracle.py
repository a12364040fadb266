"""TEST INFRASTRUCTURE ONLY: ctypes front end of oracle/liboracle.so (the CPU restatement of the
reference kernels) and, when built, of oracle/_ref/libref_model.so (the reference's own model.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
ORACLE_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_MODEL_LIB_PATH = os.path.join(_HERE, "_ref", "libref_model.so")

import sys
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from raytracer_amd import layout as L  # structure layouts only (numpy dtypes), no product code paths


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("ray_count", C.c_uint32),
                ("warp_width", C.c_uint32), ("sample_limit", C.c_uint32), ("hit_capacity", C.c_uint32),
                ("row_offset", C.c_uint32), ("flags", C.c_uint32)]


work_dtype = np.dtype([(n, np.uint64) for n in ("box_tests", "tri_tests", "shade_calls", "node_visits", "spawned", "shadows")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB_PATH):
            raise RuntimeError(f"{ORACLE_LIB_PATH} is missing: run `make -C oracle`")
        o = C.CDLL(ORACLE_LIB_PATH)
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        o.orc_create.restype = vp
        o.orc_create.argtypes = [C.POINTER(Config)]
        o.orc_destroy.argtypes = [vp]
        o.orc_set_threads.argtypes = [vp, C.c_int]
        o.orc_upload_scene.restype = C.c_int
        o.orc_upload_scene.argtypes = [vp, vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz]
        o.orc_init_frame.argtypes = [vp]
        o.orc_step.argtypes = [vp, u32]
        o.orc_set_warp_width.argtypes = [vp, u32]
        o.orc_resolve_rgba8.argtypes = [vp, vp]
        for name in ("global", "area", "ray_list", "ray_index", "grp_data", "hits"):
            getattr(o, "orc_" + name).restype = vp
            getattr(o, "orc_" + name).argtypes = [vp]
        o.orc_pool_size.restype = u32
        o.orc_pool_size.argtypes = [vp]
        o.orc_get_counters.argtypes = [vp, vp]
        o.orc_get_work.argtypes = [vp, vp]
        o.orc_get_hit_histogram.argtypes = [vp, vp]
        o.orc_trace_closest.restype = C.c_int
        o.orc_trace_closest.argtypes = [vp, vp, u32, vp]
        o.orc_camera_rays.argtypes = [vp, vp, u32, vp]
        o.orc_init_frame_samples.restype = C.c_int
        o.orc_init_frame_samples.argtypes = [vp, vp, u32]
        o.orc_sort_pairs.argtypes = [vp, vp, u32, u32, u32]
        o.orc_sort_block_index.restype = u32
        o.orc_sort_block_index.argtypes = [u32, u32]
        o.orc_compact_groups.argtypes = [vp, vp, u32, vp, vp, u32]
        o.orc_sort_hits.argtypes = [vp, u32]
        o.orc_deinterleave.argtypes = [u32, vp]
        o.orc_subpixel.restype = C.c_float
        o.orc_subpixel.argtypes = [u32]
        _lib = o
    return _lib


def _view(ptr, count, dtype):
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class Oracle:
    """The reference pipeline on the CPU: same calls as raytracer_amd.RayTracer."""

    def __init__(self, scene, warp_width=32, sample_limit=0, hit_capacity=0, threads=1, shading=0, mad=False):
        self.scene = scene
        cfg = Config(0, scene.width, scene.height, scene.pool_size, warp_width, sample_limit, hit_capacity,
                     scene.row_offset, shading | (2 if mad else 0))  # RT_FLAG_ARITH_MAD
        self.h = lib().orc_create(C.byref(cfg))
        if not self.h:
            raise RuntimeError("orc_create failed")
        lib().orc_set_threads(self.h, threads)
        s = scene
        rc = lib().orc_upload_scene(self.h, s.data.ctypes.data, s.groups.ctypes.data, len(s.groups),
                                    s.matrices.ctypes.data, len(s.matrices), s.aabbs.ctypes.data, len(s.aabbs),
                                    s.vertices.ctypes.data, len(s.vertices), s.triangles.ctypes.data, len(s.triangles))
        if rc:
            raise RuntimeError(f"orc_upload_scene failed: {rc}")
        self.pool = lib().orc_pool_size(self.h)
        self.hit_capacity = hit_capacity

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def init_frame(self):
        lib().orc_init_frame(self.h)

    def step(self, n=1):
        lib().orc_step(self.h, n)

    def camera_rays(self, samples):
        """The camera rays of the given sample numbers, as init_ray sets them up."""
        samples = np.ascontiguousarray(samples, dtype=np.uint32)
        rays = np.zeros(len(samples), L.ray_dtype)
        lib().orc_camera_rays(self.h, samples.ctypes.data, len(samples), rays.ctypes.data)
        return rays

    def init_frame_samples(self, samples):
        """A frame in which slot i runs sample number samples[i] and nothing else is ever started."""
        samples = np.ascontiguousarray(samples, dtype=np.uint32)
        rc = lib().orc_init_frame_samples(self.h, samples.ctypes.data, len(samples))
        if rc:
            raise RuntimeError(f"orc_init_frame_samples failed: {rc}")

    def set_warp_width(self, w):
        lib().orc_set_warp_width(self.h, w)

    # live views (valid until the next call that swaps buffers: copy if kept)
    @property
    def global_data(self):
        return _view(lib().orc_global(self.h), 1, L.global_dtype)

    @property
    def area(self):
        return _view(lib().orc_area(self.h), self.scene.width * self.scene.height * 4, np.float32).reshape(-1, 4)

    @property
    def ray_list(self):
        return _view(lib().orc_ray_list(self.h), self.pool, L.ray_queue_dtype)

    @property
    def ray_index(self):
        return _view(lib().orc_ray_index(self.h), self.pool, L.index_pair_dtype)

    @property
    def grp_data(self):
        return _view(lib().orc_grp_data(self.h), self.scene.group_count, L.group_data_dtype)

    @property
    def hits(self):
        return _view(lib().orc_hits(self.h), self.hit_capacity, L.hit_record_dtype)

    def counters(self):
        out = np.zeros(1, L.counters_dtype)
        lib().orc_get_counters(self.h, out.ctypes.data)
        return out[0]

    def work(self):
        out = np.zeros(1, work_dtype)
        lib().orc_get_work(self.h, out.ctypes.data)
        return out[0]

    def hit_histogram(self):
        """Box-group visits by number of accepted boxes, bins 0..64 and "more than 64" (single-threaded runs only)."""
        out = np.zeros(L.MAX_HITS + 2, np.uint64)
        lib().orc_get_hit_histogram(self.h, out.ctypes.data)
        return out

    def resolve_rgba8(self):
        out = np.zeros((self.scene.height, self.scene.width, 4), np.uint8)
        lib().orc_resolve_rgba8(self.h, out.ctypes.data)
        return out

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=L.ray_dtype)
        hits = np.zeros(len(rays), L.hit_record_dtype)
        rc = lib().orc_trace_closest(self.h, rays.ctypes.data, len(rays), hits.ctypes.data)
        if rc:
            raise RuntimeError(f"orc_trace_closest failed: {rc}")
        return hits


def sort_pairs(pairs, n, group_count):
    pairs = np.ascontiguousarray(pairs, dtype=L.index_pair_dtype)
    out = np.zeros_like(pairs)
    lib().orc_sort_pairs(pairs.ctypes.data, out.ctypes.data, len(pairs), n, group_count)
    return out


def sort_block_index(pos, n):
    return lib().orc_sort_block_index(pos, n)


def compact_groups(sorted_pairs, grp_data, global_data, warp_width):
    sorted_pairs = np.ascontiguousarray(sorted_pairs, dtype=L.index_pair_dtype)
    out = np.zeros_like(sorted_pairs)
    grp = np.array(grp_data, dtype=L.group_data_dtype, copy=True)
    glob = np.array(global_data, dtype=L.global_dtype, copy=True)
    lib().orc_compact_groups(sorted_pairs.ctypes.data, out.ctypes.data, len(sorted_pairs), grp.ctypes.data,
                             glob.ctypes.data, warp_width)
    return out, grp, glob


def sort_hits(hits):
    hits = np.array(hits, dtype=L.ray_hit_dtype, copy=True)
    lib().orc_sort_hits(hits.ctypes.data, len(hits))
    return hits


def deinterleave(val):
    out = np.zeros(2, np.uint32)
    lib().orc_deinterleave(val, out.ctypes.data)
    return int(out[0]), int(out[1])


def subpixel(val):
    return float(lib().orc_subpixel(val))


# ---------------------------------------------------------------- the reference's own model.cpp

_ref = None


def ref_model_lib():
    """oracle/_ref/libref_model.so, or None when it has not been built (no /root/reference at build time)."""
    global _ref
    if _ref is None and os.path.exists(REF_MODEL_LIB_PATH):
        r = C.CDLL(REF_MODEL_LIB_PATH)
        r.refm_build.restype = C.c_void_p
        r.refm_build.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        r.refm_free.argtypes = [C.c_void_p]
        for n in ("group", "aabb", "vertex", "triangle"):
            getattr(r, f"refm_{n}_count").restype = C.c_size_t
            getattr(r, f"refm_{n}_count").argtypes = [C.c_void_p]
        for n in ("groups", "aabbs", "vertices", "triangles"):
            getattr(r, "refm_" + n).restype = C.c_void_p
            getattr(r, "refm_" + n).argtypes = [C.c_void_p]
        r.refm_root_id.restype = C.c_uint
        r.refm_root_id.argtypes = [C.c_void_p]
        r.refm_set_camera.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _ref = r
    return _ref


def ref_layout():
    """Sizes, member offsets and constants of the reference's ray-tracer.h as the reference build sees them."""
    r = ref_model_lib()
    r.refm_layout.restype = C.c_char_p
    return {k: int(v) for k, v in (item.split("=") for item in r.refm_layout().decode().split(";") if item)}


def ref_model_build(files, matrices, tri_threshold=128, aabb_threshold=128):
    """Runs the reference's Model/ResourceManager classes; returns (groups, aabbs, vertices, triangles)."""
    r = ref_model_lib()
    arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
    matrices = np.ascontiguousarray(matrices, dtype=L.matrix_dtype)
    h = r.refm_build(arr, len(files), matrices.ctypes.data, len(matrices), tri_threshold, aabb_threshold)
    if not h:
        raise RuntimeError("reference model build failed")
    try:
        def get(name, cnt, dtype):
            count = getattr(r, f"refm_{cnt}_count")(h)
            raw = C.string_at(getattr(r, "refm_" + name)(h), count * np.dtype(dtype).itemsize)
            return np.frombuffer(raw, dtype=dtype).copy()
        return (get("groups", "group", L.group_dtype), get("aabbs", "aabb", L.aabb_dtype),
                get("vertices", "vertex", L.vertex_dtype), get("triangles", "triangle", np.uint32))
    finally:
        r.refm_free(h)


def ref_set_camera(width, height, tan_fov, pos, view, up=(0, 0, 1)):
    cam = np.zeros(1, L.camera_dtype)
    f3 = lambda v: (C.c_float * 3)(*v)
    ref_model_lib().refm_set_camera(cam.ctypes.data, width, height, tan_fov, f3(pos), f3(view), f3(up))
    return cam
