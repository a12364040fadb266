"""ctypes binding of the C ABI in include/rt_hip.h (raytracer_amd/csrc/librt_hip.so) and a Python
mirror of the reference's host-side renderer object (class RayTracer, main.cpp:50-207): init,
init_frame, make_step, draw_frame, current_ray.  The product host is C++ (raytracer_amd/host); this
module is the plumbing tests and the benchmark use.  There is no CPU fallback: if the HIP library is
missing or no GPU is present, construction fails."""
import ctypes as C
import os

import numpy as np

from . import layout as L

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "csrc", "librt_hip.so")

STAGE_NAMES = ("traverse", "shade", "sort", "compact", "leaves")
FLAG_FREE_SCHEDULE = 1  # RT_FLAG_FREE_SCHEDULE
FLAG_ARITH_MAD = 2      # RT_FLAG_ARITH_MAD
SHADE_REFERENCE, SHADE_MIRROR, SHADE_DIFFUSE = 0, 1, 2


def shading_flags(mode, bounces, seed=0x5EED):
    """RT_FLAG_SHADING(mode, bounces, seed)."""
    return (mode & 15) << 8 | (bounces & 15) << 12 | (seed & 0xFFFF) << 16

EXPORTS = (
    "rt_create", "rt_destroy", "rt_upload_scene", "rt_init_frame", "rt_step", "rt_set_warp_width", "rt_sync",
    "rt_resolve_rgba8", "rt_resolve_rgba8_enqueue", "rt_read_area", "rt_current_ray", "rt_get_global", "rt_get_counters", "rt_sort_pairs",
    "rt_compact_groups", "rt_dump_state", "rt_read_hits", "rt_trace_closest", "rt_set_profiling",
    "rt_get_stage_times", "rt_debug_stamps", "rt_sort_pairs_static", "rt_image_device", "rt_stream", "rt_status_string", "rt_last_error",
)


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("ray_count", C.c_uint32),
                ("warp_width", C.c_uint32), ("sample_limit", C.c_uint32), ("hit_capacity", C.c_uint32),
                ("row_offset", C.c_uint32), ("flags", C.c_uint32)]


class RtError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"{status}: {detail}")
        self.status = status


_lib = None


def hip_lib():
    """Loads librt_hip.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        path = os.environ.get("RT_HIP_LIB", HIP_LIB_PATH)  # a diagnostic build of the same library, if given
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make device` (or __graft_entry__.build())")
        lib = C.CDLL(path)
        vp, u32, sz, i = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
        sig = {
            "rt_create": (i, [C.POINTER(Config), C.POINTER(vp)]),
            "rt_destroy": (i, [vp]),
            "rt_upload_scene": (i, [vp, vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz]),
            "rt_init_frame": (i, [vp]),
            "rt_step": (i, [vp, u32]),
            "rt_set_warp_width": (i, [vp, u32]),
            "rt_sync": (i, [vp]),
            "rt_resolve_rgba8": (i, [vp, vp, i]),
            "rt_resolve_rgba8_enqueue": (i, [vp, vp]),
            "rt_read_area": (i, [vp, vp]),
            "rt_current_ray": (i, [vp, C.POINTER(u32)]),
            "rt_get_global": (i, [vp, vp]),
            "rt_get_counters": (i, [vp, vp]),
            "rt_sort_pairs": (i, [vp, vp, vp, u32, u32, u32, i]),
            "rt_sort_pairs_static": (i, [vp, vp, u32, u32, u32, vp]),
            "rt_compact_groups": (i, [vp, vp, vp, u32, vp, vp, u32]),
            "rt_dump_state": (i, [vp, vp, vp, vp]),
            "rt_read_hits": (i, [vp, vp, u32, u32]),
            "rt_trace_closest": (i, [vp, vp, u32, vp]),
            "rt_set_profiling": (i, [vp, i]),
            "rt_get_stage_times": (i, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
            "rt_debug_stamps": (i, [vp, C.POINTER(C.c_uint64), i]),
            "rt_image_device": (vp, [vp]),
            "rt_stream": (vp, [vp]),
            "rt_status_string": (C.c_char_p, [i]),
            "rt_last_error": (C.c_char_p, [vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class RayTracer:
    """One GPU context.  Mirrors the reference's RayTracer: init (constructor + upload), init_frame,
    make_step, draw_frame, current_ray (main.cpp:185-206)."""

    def __init__(self, scene, device=0, warp_width=0, sample_limit=0, hit_capacity=0, free_schedule=False, shading=0, mad=False):
        self.lib = hip_lib()
        self.scene = scene
        self.h = C.c_void_p()
        cfg = Config(device, scene.width, scene.height, scene.pool_size, warp_width, sample_limit, hit_capacity,
                     scene.row_offset, (FLAG_FREE_SCHEDULE if free_schedule else 0) | (FLAG_ARITH_MAD if mad else 0) | shading)
        rc = self.lib.rt_create(C.byref(cfg), C.byref(self.h))
        if rc:
            detail = self.lib.rt_last_error(self.h).decode() if self.h else ""
            if self.h:
                self.lib.rt_destroy(self.h)
                self.h = C.c_void_p()
            raise RtError(self.lib.rt_status_string(rc).decode(), detail)
        self.pool = L.align_up(scene.pool_size, L.SORT_TILE)
        self.hit_capacity = hit_capacity
        self.mad = bool(mad)
        self.upload(scene)

    def _check(self, rc):
        if rc:
            raise RtError(self.lib.rt_status_string(rc).decode(), self.lib.rt_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, s):
        self._check(self.lib.rt_upload_scene(
            self.h, s.data.ctypes.data, s.groups.ctypes.data, len(s.groups), s.matrices.ctypes.data, len(s.matrices),
            s.aabbs.ctypes.data, len(s.aabbs), s.vertices.ctypes.data, len(s.vertices), s.triangles.ctypes.data,
            len(s.triangles)))

    # ---- the reference's interface
    def init_frame(self):
        self._check(self.lib.rt_init_frame(self.h))

    def make_step(self, n=1):
        self._check(self.lib.rt_step(self.h, n))

    step = make_step

    def draw_frame(self, out=None):
        """RGBA8 image, rows top to bottom as the reference's texture (height, width, 4)."""
        if out is None:
            out = np.zeros((self.scene.height, self.scene.width, 4), np.uint8)
        self._check(self.lib.rt_resolve_rgba8(self.h, out.ctypes.data, 0))
        return out

    def draw_frame_device(self, device_ptr=None):
        """Resolves into device memory: the caller's buffer, or the context's own image (image_device)."""
        self._check(self.lib.rt_resolve_rgba8(self.h, C.c_void_p(device_ptr) if device_ptr else None, 1))

    def enqueue_draw_frame_device(self, device_ptr=None):
        """The same without waiting for it (rt_sync does)."""
        self._check(self.lib.rt_resolve_rgba8_enqueue(self.h, C.c_void_p(device_ptr) if device_ptr else None))

    def image_device(self):
        return self.lib.rt_image_device(self.h)

    def current_ray(self):
        v = C.c_uint32()
        self._check(self.lib.rt_current_ray(self.h, C.byref(v)))
        return v.value

    # ---- beyond the reference's interface
    def sync(self):
        self._check(self.lib.rt_sync(self.h))

    def set_warp_width(self, w):
        self._check(self.lib.rt_set_warp_width(self.h, w))

    def global_data(self):
        out = np.zeros(1, L.global_dtype)
        self._check(self.lib.rt_get_global(self.h, out.ctypes.data))
        return out

    def counters(self):
        out = np.zeros(1, L.counters_dtype)
        self._check(self.lib.rt_get_counters(self.h, out.ctypes.data))
        return out[0]

    def area(self):
        out = np.zeros((self.scene.width * self.scene.height, 4), np.float32)
        self._check(self.lib.rt_read_area(self.h, out.ctypes.data))
        return out

    def dump_state(self, rays=True, index=True, groups=True):
        ray_list = np.zeros(self.pool, L.ray_queue_dtype) if rays else None
        ray_index = np.zeros(self.pool, L.index_pair_dtype) if index else None
        grp_data = np.zeros(self.scene.group_count, L.group_data_dtype) if groups else None
        ptr = lambda a: a.ctypes.data if a is not None else None
        self._check(self.lib.rt_dump_state(self.h, ptr(ray_list), ptr(ray_index), ptr(grp_data)))
        return ray_list, ray_index, grp_data

    def hits(self, first=0, count=None):
        count = self.hit_capacity - first if count is None else count
        out = np.zeros(count, L.hit_record_dtype)
        self._check(self.lib.rt_read_hits(self.h, out.ctypes.data, first, count))
        return out

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=L.ray_dtype)
        if self.mad and np.any(rays["dir_max"][:, :3] == 0):
            # the fused slab test needs 1 / dir to be a number on every axis (include/rt_hip.h, RT_FLAG_ARITH_MAD)
            raise ValueError("a ray with an exactly zero direction component on a context with fused arithmetic: trace it on a strict context")
        hits = np.zeros(len(rays), L.hit_record_dtype)
        self._check(self.lib.rt_trace_closest(self.h, rays.ctypes.data, len(rays), hits.ctypes.data))
        return hits

    def sort_pairs(self, pairs, n, group_count):
        pairs = np.ascontiguousarray(pairs, dtype=L.index_pair_dtype)
        out = np.zeros_like(pairs)
        self._check(self.lib.rt_sort_pairs(self.h, pairs.ctypes.data, out.ctypes.data, len(pairs), n, group_count, 0))
        return out

    def sort_pairs_device(self, in_ptr, out_ptr, count, n, group_count):
        self._check(self.lib.rt_sort_pairs(self.h, C.c_void_p(in_ptr), C.c_void_p(out_ptr), count, n, group_count, 1))

    def compact_groups(self, sorted_pairs, grp_data, global_data, warp_width):
        sorted_pairs = np.ascontiguousarray(sorted_pairs, dtype=L.index_pair_dtype)
        out = np.zeros_like(sorted_pairs)
        grp = np.array(grp_data, dtype=L.group_data_dtype, copy=True)
        glob = np.array(global_data, dtype=L.global_dtype, copy=True)
        self._check(self.lib.rt_compact_groups(self.h, sorted_pairs.ctypes.data, out.ctypes.data, len(sorted_pairs),
                                               grp.ctypes.data, glob.ctypes.data, warp_width))
        return out, grp, glob

    def set_profiling(self, mode):
        """0 off, 1 stage timers on the concurrent schedule, 2 stage timers with the kernels serialised."""
        self._check(self.lib.rt_set_profiling(self.h, int(mode)))

    def stage_times(self):
        """({stage: total ms}, timed steps) since profiling was enabled."""
        ms = (C.c_double * len(STAGE_NAMES))()
        steps = C.c_uint64()
        self._check(self.lib.rt_get_stage_times(self.h, ms, C.byref(steps)))
        return dict(zip(STAGE_NAMES, list(ms))), steps.value

    def debug_stamps(self, reset=False):
        out = (C.c_uint64 * 8)()
        self._check(self.lib.rt_debug_stamps(self.h, out, int(reset)))
        return list(out)

    def stream(self):
        return self.lib.rt_stream(self.h)
