/*
 * rt_hip.h -- C ABI of the MI355X (gfx950) ray-casting layer.
 *
 * This library replaces the OpenCL layer the reference host drives through
 * cl-helper.h (cl-helper.h:11-64) and class RayTracer (main.cpp:50-207):
 * context/queue/program/buffer/kernel objects become one opaque rt_ctx, the
 * eleven OpenCL kernels (ray-tracer.cl, shader.cl, sort.cl) become precompiled
 * gfx950 code objects, and every clEnqueue* call becomes one of the entry
 * points below.  Plain pointers and sizes only; every call returns an int
 * status (0 = RT_OK), the convention that replaces the reference's
 * "return bool + print cl_error_string(err)" (main.cpp:44-47, 87-127).
 *
 * The scene arrays keep the reference's host-side AoS layout (ray-tracer.h),
 * so the structures below are byte-compatible with Group, AABB, Vertex,
 * Matrix, Camera, GlobalData, GroupData, RayHit and RayQueue.  cl_float3 is a
 * 16-byte type, hence the 4-float arrays.
 *
 * Threading: one rt_ctx per GPU, not thread-safe per context (the reference
 * uses one host thread and one in-order queue, main.cpp:240).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ layout */

/* ray-tracer.h:62-66 */
#define RT_GROUP_ID_MASK 0xFFFFFFu
#define RT_GROUP_TR_SHIFT 24
#define RT_GROUP_TR_MASK 0xFu
#define RT_GROUP_SH_SHIFT 28
#define RT_GROUP_SH_MASK 0xFu

/* ray-tracer.h:68-76 */
enum { RT_TR_NONE = 0, RT_TR_IDENTITY = 1, RT_TR_ORTHO = 2, RT_TR_AFFINE = 3 };
enum { RT_SH_SPAWN = 0, RT_SH_SKY = 1, RT_SH_LIGHT = 2, RT_SH_MATERIAL = 3, RT_SH_AABB = 4, RT_SH_MESH = 5 };

/* ray-tracer.h:78-83 */
#define RT_SPAWN_GROUP 0u
#define RT_SKY_GROUP (1u | (uint32_t)RT_SH_SKY << RT_GROUP_SH_SHIFT)
#define RT_LIGHT_GROUP (2u | (uint32_t)RT_SH_LIGHT << RT_GROUP_SH_SHIFT)

/* ray-tracer.h:36-39 */
enum { RT_F_LOCAL0 = 1, RT_F_LOCAL1 = 2 };
/* ray-tracer.h:157-160 */
enum { RT_RAY_PRIMARY = 0, RT_RAY_SHADOW = 1 };

#define RT_MAX_HITS 64      /* ray-tracer.h:144 */
#define RT_MAX_QUEUE_LEN 8  /* ray-tracer.h:162 */
#define RT_SORT_TILE 8192   /* UNIT_WIDTH * SORT_BLOCK, main.cpp:186-189, sort.cl:5 */

typedef struct { float color[4]; } rt_mat_shader;                                   /* ray-tracer.h:15-18 */
typedef struct { float min[3]; uint32_t group_id; float max[3]; uint32_t local_id; } rt_aabb; /* :23-34 */
typedef struct { uint32_t aabb_offs, aabb_count, flags; } rt_aabb_shader;           /* :41-44 */
typedef struct { float pos[4]; float norm[4]; } rt_vertex;                          /* :49-52 */
typedef struct { uint32_t vtx_offs, tri_offs, tri_count, material_id; } rt_mesh_shader; /* :54-57 */
typedef union { rt_mat_shader material; rt_aabb_shader aabb; rt_mesh_shader mesh; } rt_group; /* :90-95 */
typedef struct { float x[4], y[4], z[4]; } rt_matrix;                               /* :85-88 */

typedef struct {                                                                    /* :100-104 */
    float eye[4], top_left[4], dx[4], dy[4];
    uint32_t width, height, root_group, root_local;
} rt_camera;

typedef struct {                                                                    /* :106-111 */
    uint32_t pixel_offset, pixel_count;
    uint32_t group_count, old_count, ray_count;
    uint32_t pad_[3];
    rt_camera cam;
} rt_global_data;

typedef struct { uint32_t base[2], count[2], offset[2]; } rt_group_data;            /* :116-119 */
typedef struct { float pos; uint32_t group_id; uint32_t local_id[2]; } rt_ray_hit;  /* :146-155 */

typedef struct {                                                                    /* :164-170 */
    float weight[4];
    uint32_t pixel, type, material_id, queue_len;
    float start_min[4], dir_max[4];
    float norm[4];
    rt_ray_hit root, orig;
    rt_ray_hit queue[RT_MAX_QUEUE_LEN];
} rt_ray_queue;

typedef struct { uint32_t key, slot; } rt_index_pair; /* one uint2 of ray_index, main.cpp:352-353 */

/* Closest-hit record of one primary sample.  The reference computes these
 * values in mesh_shader (shader.cl:132) and the hit commit
 * (ray-tracer.cl:180-182) but keeps only t, leaf and instance in
 * RayQueue.orig; tri is the index of the triangle inside its leaf. */
typedef struct { float t; uint32_t leaf_group; uint32_t instance; uint32_t tri; } rt_hit_record;

/* One ray handed to rt_trace_closest: (start, t_min), (dir, t_max) as in Ray (ray-tracer.h:122-142). */
typedef struct { float start_min[4]; float dir_max[4]; } rt_ray;

typedef struct {
    uint64_t primary_spawned; /* samples started: sum of pixel_count, cf. main.cpp:538-540 */
    uint64_t shadow_spawned;  /* secondary rays started: mat_shader invocations (shader.cl:20-35), shadow rays and bounces alike */
    uint64_t steps;           /* make_step calls so far */
    uint64_t box_tests;       /* boxes scanned by aabb_shader, summed over rays (shader.cl:84-99) */
    uint64_t tri_tests;       /* triangles scanned by mesh_shader, summed over rays (shader.cl:123-133) */
} rt_counters;

/* -------------------------------------------------------------- lifecycle */

typedef struct rt_ctx rt_ctx;

typedef struct {
    int32_t device;        /* HIP device ordinal */
    uint32_t width;        /* image width in pixels (of this context's band) */
    uint32_t height;       /* image height in pixels; width*height must be > 0 */
    uint32_t ray_count;    /* ray slots; rounded up to a multiple of RT_SORT_TILE like main.cpp:188 */
    uint32_t warp_width;   /* queue-compaction granularity, WARP_WIDTH of main.cpp:186 (reference 32); 0 selects 64 */
    uint32_t sample_limit; /* sample ids >= limit are parked in the dead group (ray-tracer.cl:61); 0 = unlimited */
    uint32_t hit_capacity; /* keep rt_hit_record for sample ids < hit_capacity; 0 = none */
    uint32_t row_offset;   /* first image row of this context's band: pixel row y is sampled as row_offset + y,
                              so a band renders the same samples as the full frame (0 for a whole frame) */
    uint32_t flags;        /* RT_FLAG_* bits; other bits must be 0 */
} rt_config;

/* rt_config.flags.  RT_FLAG_FREE_SCHEDULE: the context may advance rays in any order.  A ray's path
 * through the hierarchy depends only on its own state and the scene (process, ray-tracer.cl:119-232),
 * so every sample's result -- its hit record and what it adds to its pixel -- is the same as with the
 * reference's schedule; what differs is which slot holds which sample after a given step, i.e.
 * rt_dump_state no longer equals the reference's buffers step by step.  With the flag set the rays are
 * grouped by key rather than sorted (the order inside a group is not defined and may differ between runs:
 * it decides which sample a finished slot starts next, and with it the order in which a pixel's samples are
 * added up), sky/light/material shaders run inside the kernel that empties a ray's queue instead of as a
 * step of their own, a finished slot is re-initialised and sent through the scene root in one launch, and
 * the ray slots keep only what the kernels read: rt_dump_state shows a ray's live queue entries (those
 * behind queue_len as zeros; entries beyond the last hit may still be listed), stale `orig` fields, and
 * the weight of a camera ray that has not bounced is implied (1).  Without the flag (0) every step
 * reproduces the reference's make_step exactly. */
#define RT_FLAG_FREE_SCHEDULE 1u

/* RT_FLAG_ARITH_MAD: multiply-adds are fused, as the reference asks of its OpenCL compiler with -cl-mad-enable
 * (main.cpp:253).  Without the flag every product and sum is rounded on its own.  With it, one rounding is
 * saved at stated places and nowhere else: the three-term sums of products of `transform` (shader.cl:63-65), of
 * the normal's interpolation and rotation (shader.cl:137, ray-tracer.cl:180) and of every dot product are
 * fma(c, w, fma(b, v, a*u)); a component p*q - r*s of a cross product is fma(p, q, -(r*s)); start + dir*t and the
 * camera ray's top_left + dx*x + dy*y are chains of fma; and the slab test's (corner - start) * inv_dir
 * (shader.cl:86-87) is fma(corner, inv_dir, -(start*inv_dir)) with the second product taken once per ray.
 * Division, sqrt, pow and all comparisons are the same in both modes.  A sample's hit can differ between the modes
 * where a ray grazes an edge; within a mode every result is reproducible bit for bit.
 * Limit: the fused slab test needs 1 / dir to be a number on every axis.  For a direction component that is exactly
 * zero, fma(corner, +-inf, -+inf) is not a number for corners on one side of zero and infinite for those on the other,
 * whatever the ray's start: boxes beside the ray are entered, boxes around it can be missed, pending queues overflow
 * and the ray may never finish (rt_trace_closest then gives up with RT_ERR_NOT_READY).  The rays the library makes
 * itself -- camera, shadow and bounce rays -- have no exactly zero component; axis-parallel rays of a caller belong on
 * a context without this flag, where (corner - start) * (1 / dir) handles them as the reference does. */
#define RT_FLAG_ARITH_MAD 2u

/* Shading beyond the reference's one shadow ray per sample, also in rt_config.flags: bits 8-11 say what a
 * material group does with a sample that has hit it, bits 12-15 how many times (bounces), bits 16-31 seed
 * the random directions.  After `bounces` bounces, and always with RT_SHADE_REFERENCE, a hit is shaded as
 * the reference compiles it: direct light through one shadow ray (mat_shader, shader.cl:20-35).
 *   RT_SHADE_MIRROR   the reflection the reference left commented out (shader.cl:37-39): the sample goes
 *                     on from the hit point along dir - 2 dot(dir, n) n with its weight unchanged and
 *                     ends in the sky group;
 *   RT_SHADE_DIFFUSE  the sample goes on in a cosine-weighted random direction about the normal (turned
 *                     towards the side the ray came from), its weight multiplied by the material colour.
 * The bounce count of a sample is kept in bits 8 and up of RayQueue.type (ray-tracer.h:157-160 uses 0, 1). */
enum { RT_SHADE_REFERENCE = 0, RT_SHADE_MIRROR = 1, RT_SHADE_DIFFUSE = 2 };
#define RT_FLAG_SHADING(mode, bounces, seed) \
    (((uint32_t)(mode) & 15u) << 8 | ((uint32_t)(bounces) & 15u) << 12 | ((uint32_t)(seed) & 0xFFFFu) << 16)
#define RT_FLAG_SHADING_MASK 0xFFFFFF00u

enum {
    RT_OK = 0,
    RT_ERR_INVALID_VALUE = 1,
    RT_ERR_NO_DEVICE = 2,
    RT_ERR_OUT_OF_MEMORY = 3,
    RT_ERR_LAUNCH = 4,
    RT_ERR_NOT_READY = 5,
    RT_ERR_UNSUPPORTED = 6,
    RT_ERR_HIP = 7
};

/* Replaces init_cl + build_program + create_kernels (main.cpp:219-265, 372-428). */
int rt_create(const rt_config *cfg, rt_ctx **out);
int rt_destroy(rt_ctx *ctx);

/* Replaces create_buffers(data, ...) (main.cpp:345-370).  The arrays are
 * copied before the call returns (CL_MEM_COPY_HOST_PTR semantics); group_count
 * entries are read from grp only up to n_groups, the rest of the device table
 * up to data->group_count is zero.  data->ray_count is ignored: the live count
 * starts at the context's pool size (main.cpp:330). */
int rt_upload_scene(rt_ctx *ctx, const rt_global_data *data,
                    const rt_group *grp, size_t n_groups,
                    const rt_matrix *mat, size_t n_mat,
                    const rt_aabb *aabb, size_t n_aabb,
                    const rt_vertex *vtx, size_t n_vtx,
                    const uint32_t *tri, size_t n_tri);

/* Replaces init_frame (main.cpp:431-437): init_groups, init_rays, init_image. */
int rt_init_frame(rt_ctx *ctx);

/* Replaces make_step (main.cpp:439-477), n_steps times, asynchronously on the context's stream. */
int rt_step(rt_ctx *ctx, uint32_t n_steps);

/* Changes WARP_WIDTH for the following steps (1 drains parked tails). No reference counterpart:
 * the reference fixes it at program build time (main.cpp:252-253). */
int rt_set_warp_width(rt_ctx *ctx, uint32_t warp_width);

/* Blocks until the context's stream is idle (the reference blocks in glFinish / clEnqueueReadBuffer). */
int rt_sync(rt_ctx *ctx);

/* Replaces draw_frame + update_image (main.cpp:479-490, ray-tracer.cl:305-309): RGBA8, row-major,
 * width*height*4 bytes.  dst is a host pointer, or a device pointer when dst_is_device != 0; with
 * dst == NULL and dst_is_device != 0 the image stays in the context's own device buffer
 * (rt_image_device), as the reference's stays in its GL texture. */
int rt_resolve_rgba8(rt_ctx *ctx, void *dst, int dst_is_device);
/* The same resolve into device memory (dst, or the context's own image when dst is NULL), only enqueued on the
 * context's stream: the call does not wait, as the reference's clEnqueueNDRangeKernel of update_image does not
 * (main.cpp:485; its glFinish and release come later).  A host that drives several contexts enqueues all their
 * resolves and then waits for each (rt_sync), so that the GPUs resolve side by side. */
int rt_resolve_rgba8_enqueue(rt_ctx *ctx, void *dst_device);
/* Device address of the context's RGBA8 image (width*height*4 bytes), valid until rt_destroy. */
void *rt_image_device(rt_ctx *ctx);

/* Reads the float4 accumulator `area` (main.cpp:349) to host memory, width*height*4 floats. */
int rt_read_area(rt_ctx *ctx, float *dst);

/* Replaces current_ray (main.cpp:201-206): GlobalData.pixel_offset. */
int rt_current_ray(rt_ctx *ctx, uint32_t *pixel_offset);

/* Reads the whole GlobalData block, as debug_print does (main.cpp:130-135). */
int rt_get_global(rt_ctx *ctx, rt_global_data *dst);

int rt_get_counters(rt_ctx *ctx, rt_counters *dst);

/* ------------------------------------------------------ stage-level entries */

/* Replaces the radix-sort pass loop of make_step (main.cpp:444-463; sort.cl:19-102) for one call:
 * sorts the first n of `count` pairs by the low key bits needed for group_count groups, reading the
 * input block-wise in thread-major order as local_count does (sort.cl:26-27), and copies pairs
 * [n, count) through.  in/out are host pointers, or device pointers when on_device != 0; they must
 * not overlap.  count must be a multiple of RT_SORT_TILE and at most the context's pool size.
 * On a context created with RT_FLAG_FREE_SCHEDULE the sort is stable over the plain order of the first n
 * pairs, by the bits of group_count - 1 (what such a context's steps use for more than 32,768 groups; with
 * fewer they group the pairs without sorting them). */
int rt_sort_pairs(rt_ctx *ctx, const rt_index_pair *in, rt_index_pair *out,
                  uint32_t count, uint32_t n, uint32_t group_count, int on_device);

/* The sort of rt_sort_pairs (device pointers) for callers that cannot allocate device memory: code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in
 * the reference and needs no rt_ctx: its pass workspace is a fixed array of 8.1 MB inside the library's
 * device image (each GPU the library runs on holds it until the process ends), so the call performs no
 * hipMalloc/hipFree.  in_dev/out_dev are device pointers on the CURRENT device, must not overlap, and
 * the kernels are enqueued on `stream` (a hipStream_t of that device, NULL = the default stream).
 * count must be a multiple of RT_SORT_TILE and at most RT_SORT_STATIC_MAX_PAIRS (RT_ERR_UNSUPPORTED
 * beyond).  All calls on one GPU share that GPU's workspace: calls that may overlap in time on the
 * same GPU (different streams or threads) race and must be serialised by the caller.
 * Under stream capture: the first call on a device looks the workspace's address up (a module load, which is
 * not capturable), so make one call on that device -- or create an rt_ctx on it -- before capturing; and the
 * value of n is stored by a captured launch as it was at capture time (count, n and group_count are baked into
 * the graph like any other kernel argument). */
#define RT_SORT_STATIC_MAX_PAIRS 1048576u
int rt_sort_pairs_static(const rt_index_pair *in_dev, rt_index_pair *out_dev,
                         uint32_t count, uint32_t n, uint32_t group_count, void *stream);

/* Replaces count_groups + update_groups + set_ray_index (ray-tracer.cl:235-302; main.cpp:464-472)
 * for one call on caller data (host pointers): `sorted` holds count pairs, the first
 * global->ray_count of them sorted; grp_data holds global->group_count entries and is updated in
 * place; `global` is updated in place; `out` receives the permuted pairs. */
int rt_compact_groups(rt_ctx *ctx, const rt_index_pair *sorted, rt_index_pair *out, uint32_t count,
                      rt_group_data *grp_data, rt_global_data *global, uint32_t warp_width);

/* Reads back the state the reference keeps in ray_list, ray_index[0] and grp_data
 * (main.cpp:350-353), converted to the reference layouts.  Any pointer may be NULL. */
int rt_dump_state(rt_ctx *ctx, rt_ray_queue *ray_list, rt_index_pair *ray_index, rt_group_data *grp_data);

/* Reads rt_hit_record[first, first+count) of the per-sample hit buffer (needs hit_capacity). */
int rt_read_hits(rt_ctx *ctx, rt_hit_record *dst, uint32_t first, uint32_t count);

/* Traces n_rays caller-supplied rays to their closest hit through the uploaded scene, running the
 * traversal automaton of process (ray-tracer.cl:119-232) until every ray has left the hierarchy.
 * Host pointers.  n_rays must not exceed the pool size.  Resets the frame state. */
int rt_trace_closest(rt_ctx *ctx, const rt_ray *rays, uint32_t n_rays, rt_hit_record *hits);

/* -------------------------------------------------------------- profiling */

/* Stages of a step: box-group visits, shading + spawn, ray-key sort, queue compaction, leaf (triangle)
 * visits.  The leaf kernel runs on a second stream beside the box and shading kernels. */
enum { RT_STAGE_TRAVERSE = 0, RT_STAGE_SHADE = 1, RT_STAGE_SORT = 2, RT_STAGE_COMPACT = 3, RT_STAGE_LEAVES = 4, RT_STAGE_COUNT = 5 };

/* RT_PROFILE_STAGES: rt_step brackets every stage with HIP events on the stream the stage is launched
 * on; the schedule is unchanged, so the box, shading and leaf spans overlap.  RT_PROFILE_SERIAL: the
 * same, with the leaf kernel on the main stream between the box and shading kernels, so each span is
 * that kernel's duration alone.  Both cost throughput (event records between kernels). */
enum { RT_PROFILE_OFF = 0, RT_PROFILE_STAGES = 1, RT_PROFILE_SERIAL = 2 };
int rt_set_profiling(rt_ctx *ctx, int mode);
/* Sum of stage durations in milliseconds and number of timed steps since profiling was enabled. */
int rt_get_stage_times(rt_ctx *ctx, double ms[RT_STAGE_COUNT], uint64_t *steps);

/* Diagnostic builds of the library (-DRT_STAMPS) sum shader-clock cycles per kernel phase over all
 * waves; this reads (and optionally clears) those eight sums.  All zero in the shipped library. */
int rt_debug_stamps(rt_ctx *ctx, uint64_t dst[8], int reset);

/* The hipStream_t of the context, as an opaque pointer. */
void *rt_stream(rt_ctx *ctx);

/* ----------------------------------------------------------------- errors */

/* Replaces cl_error_string (cl-helper.h:69-130). */
const char *rt_status_string(int status);
/* Human-readable detail of the last failing call on this context ("" if none). */
const char *rt_last_error(rt_ctx *ctx);

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_H */
