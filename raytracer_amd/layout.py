"""numpy views of the structures declared in include/rt_hip.h (byte-compatible with the reference's
ray-tracer.h: Group, AABB, Vertex, Matrix, Camera, GlobalData, GroupData, RayHit, RayQueue)."""
import numpy as np

u32, f32 = np.uint32, np.float32

GROUP_ID_MASK = 0xFFFFFF
GROUP_TR_SHIFT = 24
GROUP_SH_SHIFT = 28
TR_NONE, TR_IDENTITY, TR_ORTHO, TR_AFFINE = range(4)
SH_SPAWN, SH_SKY, SH_LIGHT, SH_MATERIAL, SH_AABB, SH_MESH = range(6)
SPAWN_GROUP = 0
SKY_GROUP = 1 | SH_SKY << GROUP_SH_SHIFT
LIGHT_GROUP = 2 | SH_LIGHT << GROUP_SH_SHIFT
MAX_HITS = 64
MAX_QUEUE_LEN = 8
SORT_TILE = 8192

group_dtype = np.dtype([("w", u32, 4)])  # union: material.color / aabb / mesh, 16 bytes
aabb_dtype = np.dtype([("min", f32, 3), ("group_id", u32), ("max", f32, 3), ("local_id", u32)])
vertex_dtype = np.dtype([("pos", f32, 4), ("norm", f32, 4)])
matrix_dtype = np.dtype([("x", f32, 4), ("y", f32, 4), ("z", f32, 4)])
camera_dtype = np.dtype([("eye", f32, 4), ("top_left", f32, 4), ("dx", f32, 4), ("dy", f32, 4),
                         ("width", u32), ("height", u32), ("root_group", u32), ("root_local", u32)])
global_dtype = np.dtype([("pixel_offset", u32), ("pixel_count", u32), ("group_count", u32), ("old_count", u32),
                         ("ray_count", u32), ("pad_", u32, 3), ("cam", camera_dtype)])
group_data_dtype = np.dtype([("base", u32, 2), ("count", u32, 2), ("offset", u32, 2)])
ray_hit_dtype = np.dtype([("pos", f32), ("group_id", u32), ("local_id", u32, 2)])
ray_queue_dtype = np.dtype([("weight", f32, 4), ("pixel", u32), ("type", u32), ("material_id", u32),
                            ("queue_len", u32), ("start_min", f32, 4), ("dir_max", f32, 4), ("norm", f32, 4),
                            ("root", ray_hit_dtype), ("orig", ray_hit_dtype),
                            ("queue", ray_hit_dtype, MAX_QUEUE_LEN)])
index_pair_dtype = np.dtype([("key", u32), ("slot", u32)])
hit_record_dtype = np.dtype([("t", f32), ("leaf_group", u32), ("instance", u32), ("tri", u32)])
ray_dtype = np.dtype([("start_min", f32, 4), ("dir_max", f32, 4)])
counters_dtype = np.dtype([("primary_spawned", np.uint64), ("shadow_spawned", np.uint64), ("steps", np.uint64),
                           ("box_tests", np.uint64), ("tri_tests", np.uint64)])

assert group_dtype.itemsize == 16 and aabb_dtype.itemsize == 32 and vertex_dtype.itemsize == 32
assert matrix_dtype.itemsize == 48 and camera_dtype.itemsize == 80 and global_dtype.itemsize == 112
assert group_data_dtype.itemsize == 24 and ray_hit_dtype.itemsize == 16 and ray_queue_dtype.itemsize == 240
assert index_pair_dtype.itemsize == 8 and hit_record_dtype.itemsize == 16 and ray_dtype.itemsize == 32


def group_shader(key):
    return (np.asarray(key) >> GROUP_SH_SHIFT) & 0xF


def group_transform(key):
    return (np.asarray(key) >> GROUP_TR_SHIFT) & 0xF


def align_up(value, unit):
    return (value + unit - 1) // unit * unit
