"""The C-ABI library loads and exports every function include/rt_hip.h declares; without a GPU its
lifecycle calls fail cleanly instead of computing anything."""
import ctypes as C
import os
import re

import pytest

from raytracer_amd import api, layout as L

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rt_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    assert set(names) == set(api.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(api.HIP_LIB_PATH)
    for name in declared_functions():
        assert getattr(lib, name) is not None, name


def test_struct_sizes_match_the_reference_layout():
    # sizes the reference's ray-tracer.h structs have with 16-byte cl_float3
    assert L.group_dtype.itemsize == 16
    assert L.aabb_dtype.itemsize == 32
    assert L.vertex_dtype.itemsize == 32
    assert L.matrix_dtype.itemsize == 48
    assert L.camera_dtype.itemsize == 80
    assert L.global_dtype.itemsize == 112
    assert L.global_dtype.fields["ray_count"][1] == 16 and L.global_dtype.fields["cam"][1] == 32
    assert L.group_data_dtype.itemsize == 24
    assert L.ray_queue_dtype.itemsize == 240
    off = {k: v[1] for k, v in L.ray_queue_dtype.fields.items()}
    assert (off["weight"], off["pixel"], off["start_min"], off["norm"], off["root"], off["orig"], off["queue"]) == \
        (0, 16, 32, 64, 80, 96, 112)


def test_status_strings():
    lib = api.hip_lib()
    assert lib.rt_status_string(0) == b"RT_OK"
    assert lib.rt_status_string(2) == b"RT_ERR_NO_DEVICE"
    assert lib.rt_status_string(12345) == b"RT_ERR_UNKNOWN"


def test_invalid_config_is_rejected_without_touching_a_gpu():
    lib = api.hip_lib()
    h = C.c_void_p()
    assert lib.rt_create(None, C.byref(h)) == 1
    cfg = api.Config(0, 0, 16, 8192, 0, 0, 0, 0, 0)  # zero width
    assert lib.rt_create(C.byref(cfg), C.byref(h)) == 1
    cfg = api.Config(0, 16, 16, 8192, 0, 0, 0, 0, 7)  # reserved flags
    assert lib.rt_create(C.byref(cfg), C.byref(h)) == 1
    assert lib.rt_step(None, 1) == 1
    assert lib.rt_destroy(None) == 1


def test_without_a_gpu_the_product_fails_loudly(bunny_scene_64):
    """There is no CPU path behind the C ABI: on a machine without a GPU a context cannot be created, the Python
    mirror raises, and the C++ host exits with an error instead of rendering anything."""
    import subprocess
    lib = api.hip_lib()
    h = C.c_void_p()
    cfg = api.Config(0, 64, 64, 8192, 0, 0, 0, 0, 0)
    rc = lib.rt_create(C.byref(cfg), C.byref(h))
    if rc == 0:
        lib.rt_destroy(h)
        pytest.skip("this machine has a GPU")
    assert lib.rt_status_string(rc) == b"RT_ERR_NO_DEVICE" and not h
    with pytest.raises(api.RtError, match="RT_ERR_NO_DEVICE"):
        api.RayTracer(bunny_scene_64)
    exe = os.path.join(os.path.dirname(api.HIP_LIB_PATH), "..", "host", "rt_main")
    run = subprocess.run([exe, "--width", "64", "--height", "64", "--frames", "1", "--out", os.devnull],
                         capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(HEADER)))
    assert run.returncode == 1 and "RT_ERR_NO_DEVICE" in run.stdout
    assert subprocess.run([exe, "--spp", "600", "--width", "3840", "--height", "2160"], capture_output=True).returncode == 2
    assert subprocess.run([exe, "--no-such-flag"], capture_output=True).returncode == 2
