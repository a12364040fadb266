"""Scene construction through the C++ host library (raytracer_amd/host/librt_host.so).

The scene builder is C++ (as in the reference, main.cpp:268-343 / model.cpp); this module only
marshals its flat arrays into numpy for the tests and the benchmark."""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import layout as L

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "host", "librt_host.so")
ASSETS = os.path.join(os.path.dirname(_HERE), "assets")
BUNNY = os.path.join(ASSETS, "bun_zipper.ply")

_lib = None


class _Spec(C.Structure):
    _fields_ = [("model_files", C.c_char_p), ("n_obj", C.c_uint32), ("tri_threshold", C.c_uint32),
                ("aabb_threshold", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("ray_count", C.c_uint32), ("band_first", C.c_uint32), ("band_rows", C.c_uint32),
                ("grid", C.c_uint32)]


def host_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run `make host` (or __graft_entry__.build())")
        lib = C.CDLL(HOST_LIB_PATH)
        lib.rth_scene_build.restype = C.c_void_p
        lib.rth_scene_build.argtypes = [C.POINTER(_Spec), C.c_char_p, C.c_size_t]
        lib.rth_scene_free.argtypes = [C.c_void_p]
        for name in ("global", "groups", "matrices", "aabbs", "vertices", "triangles"):
            fn = getattr(lib, "rth_scene_" + name)
            fn.restype = C.c_void_p
            fn.argtypes = [C.c_void_p] if name == "global" else [C.c_void_p, C.POINTER(C.c_size_t)]
        lib.rth_scene_pool_size.restype = C.c_uint32
        lib.rth_scene_pool_size.argtypes = [C.c_void_p]
        lib.rth_scene_row_offset.restype = C.c_uint32
        lib.rth_scene_row_offset.argtypes = [C.c_void_p]
        lib.rth_scene_resident_triangles.restype = C.c_uint64
        lib.rth_scene_resident_triangles.argtypes = [C.c_void_p]
        lib.rth_scene_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.rth_write_standin_ply.restype = C.c_int
        lib.rth_write_standin_ply.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
        lib.rth_set_camera.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = lib
    return _lib


@dataclass
class Scene:
    data: np.ndarray       # global_dtype, shape (1,)
    groups: np.ndarray     # group_dtype
    matrices: np.ndarray   # matrix_dtype
    aabbs: np.ndarray      # aabb_dtype
    vertices: np.ndarray   # vertex_dtype
    triangles: np.ndarray  # uint32
    pool_size: int
    row_offset: int
    resident_triangles: int
    ingest_seconds: float = 0.0      # reading the model files (text parse, or the host's binary mesh cache)
    build_seconds: float = 0.0       # hierarchy build + scene assembly
    models_from_cache: int = 0

    @property
    def group_count(self):
        return int(self.data["group_count"][0])

    @property
    def width(self):
        return int(self.data["cam"]["width"][0])

    @property
    def height(self):
        return int(self.data["cam"]["height"][0])


def _copy(ptr, count, dtype):
    """An owned numpy array of `count` records copied from the library's memory (one copy, no intermediate bytes object)."""
    if not count:
        return np.zeros(0, dtype)
    view = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(view, dtype=dtype, count=count).copy()


def build_scene(models, width=1024, height=1024, ray_count=1024 * 1024, n_obj=256, tri_threshold=128,
                aabb_threshold=128, band_first=0, band_rows=0, grid=0):
    """Builds the reference's instanced scene for the given list of PLY files."""
    lib = host_lib()
    spec = _Spec(";".join(models).encode(), n_obj, tri_threshold, aabb_threshold, width, height, ray_count,
                 band_first, band_rows, grid)
    err = C.create_string_buffer(512)
    handle = lib.rth_scene_build(C.byref(spec), err, len(err))
    if not handle:
        raise RuntimeError("scene build failed: " + err.value.decode())
    try:
        n = C.c_size_t()

        def arr(name, dtype):
            ptr = getattr(lib, "rth_scene_" + name)(handle, C.byref(n))
            return _copy(ptr, n.value, np.dtype(dtype))

        times = (C.c_double * 3)()
        lib.rth_scene_timings(handle, times)
        return Scene(
            ingest_seconds=times[0], build_seconds=times[1], models_from_cache=int(times[2]),
            data=_copy(lib.rth_scene_global(handle), 1, L.global_dtype),
            groups=arr("groups", L.group_dtype), matrices=arr("matrices", L.matrix_dtype),
            aabbs=arr("aabbs", L.aabb_dtype), vertices=arr("vertices", L.vertex_dtype),
            triangles=arr("triangles", np.uint32), pool_size=lib.rth_scene_pool_size(handle),
            row_offset=lib.rth_scene_row_offset(handle),
            resident_triangles=lib.rth_scene_resident_triangles(handle))
    finally:
        lib.rth_scene_free(handle)


def write_standin_ply(path, around=934, rings=466):
    """Writes the synthetic stand-in for dragon_vrip.ply (about 870k triangles by default)."""
    if host_lib().rth_write_standin_ply(path.encode(), around, rings) != 0:
        raise RuntimeError(f"cannot write {path}")
    return path


def set_camera(width, height, tan_fov, pos, view, up=(0, 0, 1)):
    cam = np.zeros(1, L.camera_dtype)
    f3 = lambda v: (C.c_float * 3)(*v)
    host_lib().rth_set_camera(cam.ctypes.data, width, height, tan_fov, f3(pos), f3(view), f3(up))
    return cam
