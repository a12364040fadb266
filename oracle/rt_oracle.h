/*
 * rt_oracle.h -- TEST INFRASTRUCTURE ONLY: the checker for the HIP path.
 *
 * A plain-C, CPU restatement of the reference's device kernels (ray-tracer.cl, shader.cl, sort.cl) and
 * of the launch sequence that drives them (main.cpp:431-477), on the reference's own data layouts
 * (ray-tracer.h, restated in include/rt_hip.h).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product never does.
 *
 * PARITY UNPINNED for the kernels: the reference ships no tests, golden vectors or fixtures for this
 * path, and its kernels are OpenCL C that cannot run in this image (no OpenCL device, no CPU OpenCL
 * runtime), so this restatement is checked against the reference's source text only.  (The host-side
 * scene build, by contrast, is pinned against a real build of the reference's model.cpp.)
 *
 * Floating point: every expression is evaluated in float, left to right as written, without fused
 * multiply-add (-ffp-contract=off); division and sqrt are IEEE.  Three OpenCL built-ins whose
 * rounding the standard leaves open are given one definition here and in the HIP path:
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   normalize(v)  = v * (1 / sqrt(dot(v,v)))
 *   pow(x, 100), pow(x, 5) = products of repeated squares (integer exponents, shader.cl:26-27)
 * With RT_FLAG_ARITH_MAD in rt_config.flags the multiply-adds named in include/rt_hip.h are fused (fmaf),
 * as the reference's -cl-mad-enable allows (main.cpp:253); nothing else changes.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

typedef struct {
    uint64_t box_tests;     /* boxes scanned by aabb_shader (shader.cl:84) */
    uint64_t tri_tests;     /* triangles scanned by mesh_shader (shader.cl:123) */
    uint64_t shade_calls;   /* mat/sky/light shader invocations */
    uint64_t node_visits;   /* aabb + mesh group visits */
    uint64_t spawned;       /* init_ray calls that started a sample (ray-tracer.cl:56-74) */
    uint64_t shadows;       /* mat_shader calls = shadow rays started (shader.cl:33-35) */
} orc_work;

/* Mirrors rt_create: same configuration, same defaults (warp_width 0 -> 64, pool rounded to 8192). */
orc_ctx *orc_create(const rt_config *cfg);
void orc_destroy(orc_ctx *ctx);
void orc_set_threads(orc_ctx *ctx, int threads); /* work-groups are spread over this many host threads (1 = deterministic) */

int orc_upload_scene(orc_ctx *ctx, const rt_global_data *data, const rt_group *grp, size_t n_groups,
                     const rt_matrix *mat, size_t n_mat, const rt_aabb *aabb, size_t n_aabb,
                     const rt_vertex *vtx, size_t n_vtx, const uint32_t *tri, size_t n_tri);
void orc_init_frame(orc_ctx *ctx);              /* main.cpp:431-437 */
void orc_step(orc_ctx *ctx, uint32_t n_steps);  /* main.cpp:439-477 */
void orc_set_warp_width(orc_ctx *ctx, uint32_t warp_width);
void orc_resolve_rgba8(orc_ctx *ctx, uint8_t *dst); /* ray-tracer.cl:305-309 */

/* Views of the live state (owned by the context). */
rt_global_data *orc_global(orc_ctx *ctx);
float *orc_area(orc_ctx *ctx);
rt_ray_queue *orc_ray_list(orc_ctx *ctx);
rt_index_pair *orc_ray_index(orc_ctx *ctx);   /* the buffer make_step would read next */
rt_group_data *orc_grp_data(orc_ctx *ctx);
rt_hit_record *orc_hits(orc_ctx *ctx);
uint32_t orc_pool_size(orc_ctx *ctx);
void orc_get_counters(orc_ctx *ctx, rt_counters *dst);
void orc_get_work(orc_ctx *ctx, orc_work *dst);
/* Visits of box groups by the number of boxes aabb_shader returned, RT_MAX_HITS + 2 bins (single-threaded runs only); a
 * node given up because it accepted more than RT_MAX_HITS boxes returns 0 and is counted there. */
void orc_get_hit_histogram(orc_ctx *ctx, uint64_t *dst);

/* Loads caller rays as primary samples 0..n-1 (replacing init_rays) and steps with granularity 1
 * until every ray has left the hierarchy; mirrors rt_trace_closest. */
int orc_trace_closest(orc_ctx *ctx, const rt_ray *rays, uint32_t n_rays, rt_hit_record *hits);

/* Camera rays of the given sample numbers, as init_ray builds them (ray-tracer.cl:56-74). */
void orc_camera_rays(orc_ctx *ctx, const uint32_t *samples, uint32_t n, rt_ray *out);
/* Starts a frame whose first n slots run the given sample numbers and nothing else; see rt_oracle.c. */
int orc_init_frame_samples(orc_ctx *ctx, const uint32_t *samples, uint32_t n);

/* ---- single stages on caller data ---- */

/* sort.cl:19-102 driven by the pass loop of main.cpp:444-463.  `count` pairs, first n sorted. */
void orc_sort_pairs(const rt_index_pair *in, rt_index_pair *out, uint32_t count, uint32_t n, uint32_t group_count);
/* sort.cl:72-82 */
uint32_t orc_sort_block_index(uint32_t pos, uint32_t n);
/* ray-tracer.cl:235-302 driven as main.cpp:464-472. */
void orc_compact_groups(const rt_index_pair *sorted, rt_index_pair *out, uint32_t count, rt_group_data *grp_data,
                        rt_global_data *global, uint32_t warp_width);
/* ray-tracer.cl:90-117 */
void orc_sort_hits(rt_ray_hit *hit, uint32_t n);
/* ray-tracer.cl:39-54 */
void orc_deinterleave(uint32_t val, uint32_t out[2]);
float orc_subpixel(uint32_t val);

#ifdef __cplusplus
}
#endif

#endif
