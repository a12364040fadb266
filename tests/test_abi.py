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


PROBE = r"""
#include <stddef.h>
#include <stdio.h>
#include "rt_hip.h"
#define SIZE(name, T) printf("sizeof %s=%zu;", name, sizeof(T))
#define OFFS(name, T, m) printf("%s=%zu;", name, offsetof(T, m))
#define PUT(name, v) printf("%s=%zu;", name, (size_t)(v))
int main(void)
{
    SIZE("MatShader", rt_mat_shader); SIZE("AABB", rt_aabb); SIZE("AABBShader", rt_aabb_shader); SIZE("Vertex", rt_vertex);
    SIZE("MeshShader", rt_mesh_shader); SIZE("Matrix", rt_matrix); SIZE("Group", rt_group); SIZE("Camera", rt_camera);
    SIZE("GlobalData", rt_global_data); SIZE("GroupData", rt_group_data); SIZE("Ray", rt_ray); SIZE("RayHit", rt_ray_hit);
    SIZE("RayQueue", rt_ray_queue);
    OFFS("AABB.min", rt_aabb, min); OFFS("AABB.max", rt_aabb, max); OFFS("AABB.group_id", rt_aabb, group_id); OFFS("AABB.local_id", rt_aabb, local_id);
    OFFS("AABBShader.aabb_offs", rt_aabb_shader, aabb_offs); OFFS("AABBShader.aabb_count", rt_aabb_shader, aabb_count); OFFS("AABBShader.flags", rt_aabb_shader, flags);
    OFFS("Vertex.pos", rt_vertex, pos); OFFS("Vertex.norm", rt_vertex, norm);
    OFFS("MeshShader.vtx_offs", rt_mesh_shader, vtx_offs); OFFS("MeshShader.tri_offs", rt_mesh_shader, tri_offs);
    OFFS("MeshShader.tri_count", rt_mesh_shader, tri_count); OFFS("MeshShader.material_id", rt_mesh_shader, material_id);
    OFFS("Matrix.x", rt_matrix, x); OFFS("Matrix.y", rt_matrix, y); OFFS("Matrix.z", rt_matrix, z);
    OFFS("Camera.eye", rt_camera, eye); OFFS("Camera.top_left", rt_camera, top_left); OFFS("Camera.dx", rt_camera, dx); OFFS("Camera.dy", rt_camera, dy);
    OFFS("Camera.width", rt_camera, width); OFFS("Camera.height", rt_camera, height); OFFS("Camera.root_group", rt_camera, root_group);
    OFFS("Camera.root_local", rt_camera, root_local);
    OFFS("GlobalData.pixel_offset", rt_global_data, pixel_offset); OFFS("GlobalData.pixel_count", rt_global_data, pixel_count);
    OFFS("GlobalData.group_count", rt_global_data, group_count); OFFS("GlobalData.old_count", rt_global_data, old_count);
    OFFS("GlobalData.ray_count", rt_global_data, ray_count); OFFS("GlobalData.cam", rt_global_data, cam);
    OFFS("GroupData.base", rt_group_data, base); OFFS("GroupData.count", rt_group_data, count); OFFS("GroupData.offset", rt_group_data, offset);
    OFFS("Ray.start_min", rt_ray, start_min); PUT("Ray.min", offsetof(rt_ray, start_min) + 12); OFFS("Ray.dir_max", rt_ray, dir_max);
    PUT("Ray.max", offsetof(rt_ray, dir_max) + 12);
    OFFS("RayHit.pos", rt_ray_hit, pos); OFFS("RayHit.group_id", rt_ray_hit, group_id); OFFS("RayHit.local_id", rt_ray_hit, local_id);
    OFFS("RayQueue.weight", rt_ray_queue, weight); OFFS("RayQueue.pixel", rt_ray_queue, pixel); OFFS("RayQueue.type", rt_ray_queue, type);
    OFFS("RayQueue.material_id", rt_ray_queue, material_id); OFFS("RayQueue.queue_len", rt_ray_queue, queue_len);
    OFFS("RayQueue.ray", rt_ray_queue, start_min); OFFS("RayQueue.norm", rt_ray_queue, norm); OFFS("RayQueue.root", rt_ray_queue, root);
    OFFS("RayQueue.orig", rt_ray_queue, orig); OFFS("RayQueue.queue", rt_ray_queue, queue);
    PUT("GROUP_ID_MASK", RT_GROUP_ID_MASK); PUT("GROUP_TR_SHIFT", RT_GROUP_TR_SHIFT); PUT("GROUP_TR_MASK", RT_GROUP_TR_MASK);
    PUT("GROUP_SH_SHIFT", RT_GROUP_SH_SHIFT); PUT("GROUP_SH_MASK", RT_GROUP_SH_MASK);
    PUT("tr_none", RT_TR_NONE); PUT("tr_identity", RT_TR_IDENTITY); PUT("tr_ortho", RT_TR_ORTHO); PUT("tr_affine", RT_TR_AFFINE);
    PUT("sh_spawn", RT_SH_SPAWN); PUT("sh_sky", RT_SH_SKY); PUT("sh_light", RT_SH_LIGHT); PUT("sh_material", RT_SH_MATERIAL);
    PUT("sh_aabb", RT_SH_AABB); PUT("sh_mesh", RT_SH_MESH);
    PUT("spawn_group", RT_SPAWN_GROUP); PUT("sky_group", RT_SKY_GROUP); PUT("light_group", RT_LIGHT_GROUP);
    PUT("f_local0", RT_F_LOCAL0); PUT("f_local1", RT_F_LOCAL1); PUT("rt_primary", RT_RAY_PRIMARY); PUT("rt_shadow", RT_RAY_SHADOW);
    PUT("MAX_HITS", RT_MAX_HITS); PUT("MAX_QUEUE_LEN", RT_MAX_QUEUE_LEN);
    return 0;
}
"""


def test_header_layout_equals_the_reference_headers(tmp_path):
    """include/rt_hip.h, compiled as C, against the reference's ray-tracer.h as the reference's own build sees it
    (oracle/_ref, g++ with the image's CL headers): every structure size, member offset and constant."""
    import subprocess
    from oracle import pyoracle as O
    if O.ref_model_lib() is None:
        pytest.skip("oracle/_ref/libref_model.so not built (needs /root/reference)")
    want = O.ref_layout()
    src, exe = tmp_path / "probe.c", tmp_path / "probe"
    src.write_text(PROBE)
    subprocess.run(["gcc", "-std=c11", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {k: int(v) for k, v in (item.split("=") for item in out.split(";") if item)}
    missing = sorted(set(want) - set(got) - {"RADIX_SHIFT"})       # (the sort's digit width is the kernels' own business)
    assert not missing, missing
    assert {k: got[k] for k in got} == {k: want[k] for k in got}
    # and the numpy views the tests use
    assert L.ray_queue_dtype.itemsize == want["sizeof RayQueue"] and L.global_dtype.itemsize == want["sizeof GlobalData"]
    for name, dtype, fields in (("RayQueue", L.ray_queue_dtype, {"weight": "weight", "pixel": "pixel", "type": "type", "material_id": "material_id",
                                                                  "queue_len": "queue_len", "ray": "start_min", "norm": "norm", "root": "root",
                                                                  "orig": "orig", "queue": "queue"}),
                                ("GlobalData", L.global_dtype, {k: k for k in ("pixel_offset", "pixel_count", "group_count", "old_count", "ray_count", "cam")}),
                                ("Camera", L.camera_dtype, {k: k for k in ("eye", "top_left", "dx", "dy", "width", "height", "root_group", "root_local")}),
                                ("AABB", L.aabb_dtype, {k: k for k in ("min", "max", "group_id", "local_id")}),
                                ("GroupData", L.group_data_dtype, {k: k for k in ("base", "count", "offset")}),
                                ("RayHit", L.ray_hit_dtype, {k: k for k in ("pos", "group_id", "local_id")})):
        for ref_field, mine in fields.items():
            assert dtype.fields[mine][1] == want[f"{name}.{ref_field}"], (name, ref_field)


def test_python_mirror_refuses_axis_parallel_rays_on_a_fused_context():
    """The fused slab test needs 1 / dir to be a number on every axis (include/rt_hip.h): the Python mirror says so
    instead of letting such a ray circle until rt_trace_closest gives up."""
    import numpy as np
    rt = api.RayTracer.__new__(api.RayTracer)      # no context needed: the check comes before the library is called
    rt.mad, rt.h = True, None
    rays = np.zeros(2, L.ray_dtype)
    rays["dir_max"][:, :3] = ((0.6, 0.8, 0.0), (0.1, 0.2, 0.3))
    with pytest.raises(ValueError, match="zero direction component"):
        rt.trace_closest(rays)
