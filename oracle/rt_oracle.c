/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h): CPU restatement of the reference's
 * device kernels, one function per kernel, work-items as loop iterations, each barrier region of a
 * kernel as its own loop.  Citations are to files under the reference tree.
 *
 * PARITY UNPINNED for these kernels (no reference fixtures exist and the kernels cannot run here).
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define UNIT_WIDTH 512u /* main.cpp:186 */
#define SORT_BLOCK 16u
#define SORT_WIDTH (SORT_BLOCK * UNIT_WIDTH) /* sort.cl:5 */
#define RADIX_SHIFT 5u                         /* ray-tracer.h:173-175 */
#define RADIX_MAX 32u
#define RADIX_MASK 31u

#define ORC_TRACE_ONLY 1u

/* The two scan loops are also compiled for CPUs with a fused multiply-add instruction, so that fmaf() is one
 * instruction there instead of a library call; which clone runs is chosen when the library is loaded.  The
 * arithmetic is the same either way: contraction stays off (-ffp-contract=off), fmaf() is exact by definition. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define HOT_LOOP __attribute__((target_clones("fma", "default")))
#else
#define HOT_LOOP
#endif

typedef struct { float x, y, z; } v3;

struct orc_ctx {
    uint32_t pool, area_size, group_count, block_count, width, height;
    uint32_t warp_width, sample_limit, hit_capacity, row_offset, mode;
    int threads;
    rt_global_data global;
    float *area;
    rt_ray_queue *ray_list;
    rt_group_data *grp_data;
    rt_index_pair *ray_index[2];
    uint32_t *local_index, *global_index;
    rt_group *grp_list;
    rt_matrix *mat_list;
    rt_aabb *aabb;
    rt_vertex *vtx;
    uint32_t *tri;
    size_t n_groups, n_mat, n_aabb, n_vtx, n_tri;
    uint32_t *sample_id;
    uint32_t shading; /* bits 8-31 of rt_config.flags: mode, bounces, seed */
    int mad;          /* RT_FLAG_ARITH_MAD: multiply-adds fused at the places the *_m helpers name */
    rt_hit_record *hits;
    rt_counters counters;
    orc_work work;
    uint64_t hit_hist[RT_MAX_HITS + 2]; /* visits by the number of boxes aabb_shader returned (a node given up for more than 64 returns 0); last bin unused */
};

/* ------------------------------------------------------------------ vector helpers */

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_load(const float *p) { return v3_make(p[0], p[1], p[2]); }
static inline void v3_store(float *p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* The two arithmetic modes.  mad == 0: every product and sum rounded on its own (the header's contract).  mad != 0
 * (RT_FLAG_ARITH_MAD, the reference's -cl-mad-enable, main.cpp:253): a product that feeds a sum is fused with it
 * (fmaf, one rounding) at exactly the places these helpers are used; everything else is as in the other mode. */
static inline float mad1(int mad, float a, float b, float c) { return mad ? fmaf(a, b, c) : a * b + c; }
/* dot(a, b) = (a.x*b.x + a.y*b.y) + a.z*b.z */
static inline float v3_dot_m(int mad, v3 a, v3 b)
{
    return mad ? fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)) : v3_dot(a, b);
}
/* cross(a, b), each component p*q - r*s */
static inline v3 v3_cross_m(int mad, v3 a, v3 b)
{
    if (!mad) return v3_cross(a, b);
    return v3_make(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
/* a*s + b */
static inline v3 v3_madd_m(int mad, v3 a, float s, v3 b)
{
    return v3_make(mad1(mad, a.x, s, b.x), mad1(mad, a.y, s, b.y), mad1(mad, a.z, s, b.z));
}
/* (a*s + b*t) + c*u */
static inline v3 v3_lin3_m(int mad, v3 a, float s, v3 b, float t, v3 c, float u)
{
    if (!mad) return v3_add(v3_add(v3_scale(a, s), v3_scale(b, t)), v3_scale(c, u));
    return v3_make(fmaf(c.x, u, fmaf(b.x, t, a.x * s)), fmaf(c.y, u, fmaf(b.y, t, a.y * s)), fmaf(c.z, u, fmaf(b.z, t, a.z * s)));
}
static inline v3 v3_normalize_m(int mad, v3 a) { return v3_scale(a, 1.0f / sqrtf(v3_dot_m(mad, a, a))); }
/* the direction of a camera ray: (top_left + dx*x) + dy*y, normalised (ray-tracer.cl:64-68) */
static inline v3 camera_dir(int mad, const rt_camera *cam, float x, float y)
{
    const v3 tl = v3_load(cam->top_left), dx = v3_load(cam->dx), dy = v3_load(cam->dy);
    return v3_normalize_m(mad, v3_madd_m(mad, dy, y, v3_madd_m(mad, dx, x, tl)));
}
static inline v3 v3_min(v3 a, v3 b) { return v3_make(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline v3 v3_max(v3 a, v3 b) { return v3_make(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

/* pow(x, 5) and pow(x, 100) for x >= 0 (shader.cl:26-27) as fixed products of repeated squares. */
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
static inline float pow100(float x)
{
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16, x64 = x32 * x32;
    return (x64 * x32) * x4;
}

/* ------------------------------------------------------------------ ray set-up */

/* ray-tracer.cl:10-16 */
static uint32_t reset_ray(rt_ray_queue *ray, uint32_t group_id, const uint32_t local_id[2], uint32_t end_group)
{
    ray->orig.group_id = ray->queue[0].group_id = group_id;
    ray->orig.local_id[0] = ray->queue[0].local_id[0] = local_id[0];
    ray->orig.local_id[1] = ray->queue[0].local_id[1] = local_id[1];
    ray->queue[0].pos = ray->start_min[3] = 0.001f;
    ray->dir_max[3] = INFINITY;
    ray->queue_len = 1;
    ray->material_id = end_group;
    return group_id;
}

/* ray-tracer.cl:39-47 */
void orc_deinterleave(uint32_t val, uint32_t out[2])
{
    uint32_t r[2] = {val & 0x55555555u, (val >> 1) & 0x55555555u};
    for (int k = 0; k < 2; k++) {
        r[k] = (r[k] | r[k] >> 1) & 0x33333333u;
        r[k] = (r[k] | r[k] >> 2) & 0x0F0F0F0Fu;
        r[k] = (r[k] | r[k] >> 4) & 0x00FF00FFu;
        r[k] = (r[k] | r[k] >> 8) & 0x0000FFFFu;
        out[k] = r[k];
    }
}

/* ray-tracer.cl:49-54: 4-bit reversal of (val % 15) + 1, centred in its fifteenth. */
float orc_subpixel(uint32_t val)
{
    val = (val % 15) + 1;
    val = (val & 1) << 3 | (val & 2) << 1 | (val & 4) >> 1 | (val & 8) >> 3;
    return ((float)val - 0.5f) / 15.0f;
}

/* ray-tracer.cl:56-74.  `sample` is the sample id; ids at or beyond sample_limit take the dead-ray
 * exit the reference left commented out at ray-tracer.cl:61 and leave the slot untouched. */
static uint32_t init_ray(orc_ctx *c, uint32_t slot, uint32_t sample)
{
    rt_ray_queue *ray = &c->ray_list[slot];
    const rt_camera *cam = &c->global.cam;
    const uint32_t total = cam->width * cam->height;
    if (c->sample_limit && sample >= c->sample_limit) return c->global.group_count - 1;
    uint32_t sub[2];
    orc_deinterleave(sample / total, sub);
    const uint32_t pixel = sample % total;
    const float x = (float)(pixel % cam->width) + orc_subpixel(sub[0]);
    const float y = (float)(pixel / cam->width + c->row_offset) + orc_subpixel(sub[1]);

    ray->pixel = pixel;
    ray->weight[0] = ray->weight[1] = ray->weight[2] = ray->weight[3] = 1;
    ray->type = RT_RAY_PRIMARY;
    v3_store(ray->start_min, v3_load(cam->eye));
    v3_store(ray->dir_max, camera_dir(c->mad, cam, x, y));
    ray->root.group_id = cam->root_group;
    ray->root.local_id[0] = cam->root_local;
    ray->root.local_id[1] = 0;

    c->sample_id[slot] = sample;
    if (sample < c->hit_capacity) {
        rt_hit_record miss = {INFINITY, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        c->hits[sample] = miss;
    }
    return reset_ray(ray, ray->root.group_id, ray->root.local_id, RT_SKY_GROUP);
}

/* ------------------------------------------------------------------ shaders (shader.cl) */

static void area_add(orc_ctx *c, uint32_t pixel, float r, float g, float b, float w)
{
    float *px = &c->area[4 * (size_t)pixel];
    px[0] += r; px[1] += g; px[2] += b; px[3] += w;
}

/* shader.cl:6-18: sky colour (0.5, 1, 1), light colour (1, 1, 1). */
static uint32_t sky_or_light_shader(orc_ctx *c, rt_ray_queue *ray, int is_light)
{
    const float cr = is_light ? 1.0f : 0.5f;
    area_add(c, ray->pixel, ray->weight[0] * cr, ray->weight[1] * 1.0f, ray->weight[2] * 1.0f, ray->weight[3] * 1.0f);
    return ray->queue[0].group_id = RT_SPAWN_GROUP;
}

/* Counter-based random numbers of the diffuse bounce (beyond the reference; the HIP path uses the same). */
static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;  x *= 0x7FEB352Du;
    x ^= x >> 15;  x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
static float uniform01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

/* The bounce a material group performs instead of shading while a sample has bounces left
 * (rt_config.flags, RT_FLAG_SHADING).  Mirror: the lines the reference left commented out, shader.cl:37-39.
 * Diffuse: normal (turned towards the incoming side) plus a uniform point of the unit sphere (Marsaglia's
 * rejection method on hashed counters), normalised; weight times the material colour. */
static uint32_t bounce_shader(rt_ray_queue *ray, const rt_mat_shader *shader, uint32_t shading, uint32_t depth, uint32_t sample, int mad)
{
    const v3 dir = v3_load(ray->dir_max), norm = v3_normalize_m(mad, v3_load(ray->norm));
    v3 out;
    if ((shading & 15u) == RT_SHADE_MIRROR) {
        out = v3_madd_m(mad, norm, -(2 * v3_dot_m(mad, dir, norm)), dir); /* dir - norm * (2 dot(dir, norm)) */
    } else {
        const v3 up = v3_dot_m(mad, dir, norm) > 0 ? v3_make(-norm.x, -norm.y, -norm.z) : norm;
        const uint32_t base = hash32((shading >> 8) ^ sample * 0x9E3779B9u ^ depth * 0x85EBCA6Bu);
        float x1 = 0, x2 = 0, s = 0;
        for (uint32_t k = 0; k < 8; k++) {
            const float a = 2 * uniform01(hash32(base + 2 * k)) - 1, b = 2 * uniform01(hash32(base + 2 * k + 1)) - 1;
            const float q = a * a + b * b;
            if (q < 1) { x1 = a; x2 = b; s = q; break; }
        }
        const float r = 2 * sqrtf(1 - s);
        const v3 v = v3_add(up, v3_make(x1 * r, x2 * r, 1 - 2 * s));
        const float len2 = v3_dot_m(mad, v, v);
        out = len2 > 1e-12f ? v3_scale(v, 1.0f / sqrtf(len2)) : up;
        ray->weight[0] *= shader->color[0]; ray->weight[1] *= shader->color[1]; ray->weight[2] *= shader->color[2];
    }
    ray->type = RT_RAY_PRIMARY | (depth + 1) << 8;
    v3_store(ray->start_min, v3_madd_m(mad, dir, ray->dir_max[3], v3_load(ray->start_min)));
    v3_store(ray->dir_max, out);
    return reset_ray(ray, ray->root.group_id, ray->root.local_id, RT_SKY_GROUP);
}

/* shader.cl:20-35 */
static uint32_t mat_shader(rt_ray_queue *ray, const rt_mat_shader *shader, uint32_t shading, uint32_t sample, int mad)
{
    if (shading & 15u) {
        const uint32_t depth = ray->type >> 8;
        if (depth < ((shading >> 4) & 15u)) return bounce_shader(ray, shader, shading, depth, sample, mad);
    }
    const float l = 0.57735026f; /* normalize((1,-1,1)) */
    const v3 light = v3_make(l, -l, l);
    const float alpha = 100, f0 = 0.04f;
    const v3 dir = v3_load(ray->dir_max), norm = v3_normalize_m(mad, v3_load(ray->norm));
    const v3 hvec = v3_normalize_m(mad, v3_sub(light, dir));
    float spec = (alpha + 2) / 8 * pow100(fmaxf(0.0f, v3_dot_m(mad, norm, hvec)));
    spec *= f0 + (1 - f0) * pow5(fmaxf(0.0f, -v3_dot_m(mad, dir, hvec)));
    const float lambert = fmaxf(0.0f, v3_dot_m(mad, light, norm));
    const float sw = spec * shader->color[3];
    const v3 color = v3_scale(v3_make(shader->color[0] + sw, shader->color[1] + sw, shader->color[2] + sw), lambert);

    ray->weight[0] *= color.x; ray->weight[1] *= color.y; ray->weight[2] *= color.z; ray->weight[3] *= 1.0f;
    ray->type = RT_RAY_SHADOW;
    v3_store(ray->start_min, v3_madd_m(mad, dir, ray->dir_max[3], v3_load(ray->start_min)));
    v3_store(ray->dir_max, light);
    return reset_ray(ray, ray->root.group_id, ray->root.local_id, RT_LIGHT_GROUP);
}

typedef struct { v3 start, dir; float min, max; } local_ray;

/* shader.cl:43-73.  mat[0..2] = columns of the rotation, mat[3] = translation. */
static void transform(uint32_t group_id, const rt_ray_queue *ray, local_ray *res, v3 mat[4], const rt_matrix *mat_list, int mad)
{
    res->start = v3_load(ray->start_min); res->min = ray->start_min[3];
    res->dir = v3_load(ray->dir_max);     res->max = ray->dir_max[3];
    switch ((group_id >> RT_GROUP_TR_SHIFT) & RT_GROUP_TR_MASK) {
    case RT_TR_IDENTITY:
        mat[0] = v3_make(1, 0, 0); mat[1] = v3_make(0, 1, 0); mat[2] = v3_make(0, 0, 1); mat[3] = v3_make(0, 0, 0);
        break;
    case RT_TR_ORTHO: {
        const rt_matrix *m = &mat_list[ray->queue[0].local_id[0]];
        mat[0] = v3_make(m->x[0], m->y[0], m->z[0]);
        mat[1] = v3_make(m->x[1], m->y[1], m->z[1]);
        mat[2] = v3_make(m->x[2], m->y[2], m->z[2]);
        mat[3] = v3_make(m->x[3], m->y[3], m->z[3]);
        const v3 rel = v3_sub(res->start, mat[3]);
        const v3 mx = v3_load(m->x), my = v3_load(m->y), mz = v3_load(m->z);
        res->start = v3_lin3_m(mad, mx, rel.x, my, rel.y, mz, rel.z);
        const v3 d = res->dir;
        res->dir = v3_lin3_m(mad, mx, d.x, my, d.y, mz, d.z);
        break;
    }
    default: /* tr_none, tr_affine: nothing (shader.cl:69-71) */
        mat[0] = mat[1] = mat[2] = mat[3] = v3_make(0, 0, 0);
        break;
    }
}

/* shader.cl:76-101 */
/* With mad, (corner - start) * inv_dir is evaluated as fma(corner, inv_dir, -(start * inv_dir)): the second
 * product once per ray, one fused operation per corner coordinate. */
HOT_LOOP static uint32_t aabb_shader(const local_ray *ray, const rt_aabb_shader *shader, const rt_ray_hit *cur, rt_ray_hit *hit,
                            const rt_aabb *aabb, int mad)
{
    aabb += shader->aabb_offs;
    const v3 inv_dir = v3_make(1 / ray->dir.x, 1 / ray->dir.y, 1 / ray->dir.z);
    const uint32_t n = shader->aabb_count;
    uint32_t hit_count = 0;
    const int flag0 = (shader->flags & RT_F_LOCAL0) != 0, flag1 = (shader->flags & RT_F_LOCAL1) != 0;
    const v3 neg = v3_make(-(ray->start.x * inv_dir.x), -(ray->start.y * inv_dir.y), -(ray->start.z * inv_dir.z));
    for (uint32_t i = 0; i < n; i++) {
        const v3 lo = v3_load(aabb[i].min), hi = v3_load(aabb[i].max);
        const v3 pos1 = mad ? v3_make(fmaf(lo.x, inv_dir.x, neg.x), fmaf(lo.y, inv_dir.y, neg.y), fmaf(lo.z, inv_dir.z, neg.z))
                            : v3_mul(v3_sub(lo, ray->start), inv_dir);
        const v3 pos2 = mad ? v3_make(fmaf(hi.x, inv_dir.x, neg.x), fmaf(hi.y, inv_dir.y, neg.y), fmaf(hi.z, inv_dir.z, neg.z))
                            : v3_mul(v3_sub(hi, ray->start), inv_dir);
        const v3 pos_min = v3_min(pos1, pos2), pos_max = v3_max(pos1, pos2);
        const float t_min = fmaxf(fmaxf(pos_min.x, pos_min.y), pos_min.z);
        const float t_max = fminf(fminf(pos_max.x, pos_max.y), pos_max.z);
        if (!(t_max > t_min && t_max > ray->min && t_min < ray->max)) continue;
        if (hit_count >= RT_MAX_HITS) return 0; /* artifacts */
        hit[hit_count].pos = fmaxf(t_min, ray->min);
        hit[hit_count].group_id = aabb[i].group_id;
        hit[hit_count].local_id[0] = flag0 ? aabb[i].local_id : cur->local_id[0];
        hit[hit_count].local_id[1] = flag1 ? aabb[i].local_id : cur->local_id[1];
        hit_count++;
    }
    return hit_count;
}

/* shader.cl:115-139.  Returns material id or 0xFFFFFFFF; norm_pos = (object-space normal, t). */
HOT_LOOP static uint32_t mesh_shader(const local_ray *ray, const rt_mesh_shader *shader, float norm_pos[4], const rt_vertex *vtx,
                            const uint32_t *tri, uint32_t *hit_tri, int mad)
{
    vtx += shader->vtx_offs;
    tri += shader->tri_offs;
    uint32_t hit_index = 0xFFFFFFFFu;
    const uint32_t n = shader->tri_count;
    float hit_u = 0, hit_v = 0;
    norm_pos[3] = ray->max;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t i0 = tri[i] & 0x3FF, i1 = (tri[i] >> 10) & 0x3FF, i2 = (tri[i] >> 20) & 0x3FF;
        v3 r = v3_load(vtx[i0].pos);
        const v3 p = v3_sub(v3_load(vtx[i1].pos), r), q = v3_sub(v3_load(vtx[i2].pos), r);
        r = v3_sub(r, ray->start);
        const v3 nrm = v3_cross_m(mad, p, q);
        float w = v3_dot_m(mad, ray->dir, nrm);
        w = 1 / w;
        const float t = v3_dot_m(mad, r, nrm) * w;
        if (!(t > ray->min && t < norm_pos[3])) continue;
        const v3 dr = v3_cross_m(mad, ray->dir, r);
        const float u = -v3_dot_m(mad, q, dr) * w, v = v3_dot_m(mad, p, dr) * w;
        if (!(u >= 0 && v >= 0 && u + v <= 1)) continue;
        norm_pos[3] = t; hit_u = u; hit_v = v; hit_index = i;
    }
    if (hit_index == 0xFFFFFFFFu) return 0xFFFFFFFFu;
    const uint32_t i0 = tri[hit_index] & 0x3FF, i1 = (tri[hit_index] >> 10) & 0x3FF, i2 = (tri[hit_index] >> 20) & 0x3FF;
    const v3 nn = v3_lin3_m(mad, v3_load(vtx[i0].norm), 1 - hit_u - hit_v, v3_load(vtx[i1].norm), hit_u, v3_load(vtx[i2].norm), hit_v);
    norm_pos[0] = nn.x; norm_pos[1] = nn.y; norm_pos[2] = nn.z;
    *hit_tri = hit_index;
    return shader->material_id;
}

/* ------------------------------------------------------------------ process (ray-tracer.cl:90-232) */

static void swap_hits(rt_ray_hit *a, rt_ray_hit *b) { rt_ray_hit t = *a; *a = *b; *b = t; }

/* ray-tracer.cl:90-117: the fixed compare-exchange network, exchange iff hit[j].pos > hit[i].pos. */
void orc_sort_hits(rt_ray_hit *hit, uint32_t n)
{
    for (uint32_t base = 1; base < n; base *= 2) {
        for (uint32_t i = base; i < n; i++) if (i & base) {            /* bitonic_flip */
            const uint32_t j = i - 2 * (i & (base - 1)) - 1;
            if (hit[j].pos > hit[i].pos) swap_hits(&hit[j], &hit[i]);
        }
        for (uint32_t offs = base / 2; offs; offs /= 2)                /* bitonic_step */
            for (uint32_t i = offs; i < n; i++) if (i & offs) {
                const uint32_t j = i - offs;
                if (hit[j].pos > hit[i].pos) swap_hits(&hit[j], &hit[i]);
            }
    }
}

/* save_queue + copy_queue (ray-tracer.cl:157-168).  When the new queue is empty the head becomes the
 * ray's end group; the reference leaves queue[0].pos indeterminate there, here it keeps its old value. */
static uint32_t save_queue(rt_ray_queue *ray, rt_ray_hit *hit, uint32_t queue_len, uint32_t end_group)
{
    if (queue_len) {
        ray->queue_len = queue_len;
        ray->start_min[3] = ray->queue[0].pos;
    } else {
        hit[0].pos = ray->queue[0].pos;
        hit[0].group_id = end_group;
        hit[0].local_id[0] = hit[0].local_id[1] = 0;
        queue_len = 1;
    }
    for (uint32_t i = 0; i < queue_len; i++) ray->queue[i] = hit[i];
    return hit[0].group_id;
}

/* One automaton step of the ray at sorted position `index`. */
static void process_one(orc_ctx *c, uint32_t index, rt_index_pair *ray_index, orc_work *work)
{
    uint32_t group_id = ray_index[index].key;
    const uint32_t offs = ray_index[index].slot;
    rt_ray_queue *ray = &c->ray_list[offs];
    const rt_group *grp = &c->grp_list[group_id & RT_GROUP_ID_MASK];
    const uint32_t dead = c->global.group_count - 1;

    local_ray cur;
    v3 mat[4];
    transform(group_id, ray, &cur, mat, c->mat_list, c->mad);
    rt_ray_hit hit[RT_MAX_QUEUE_LEN], new_hit[RT_MAX_HITS];
    uint32_t queue_len;

    switch ((group_id >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK) {
    case RT_SH_SPAWN:
        group_id = init_ray(c, offs, index + c->global.pixel_offset);
        if (group_id != dead) work->spawned++;
        break;

    case RT_SH_SKY:
    case RT_SH_LIGHT:
        if (c->mode & ORC_TRACE_ONLY) { group_id = ray->queue[0].group_id = dead; break; }
        group_id = sky_or_light_shader(c, ray, ((group_id >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK) == RT_SH_LIGHT);
        work->shade_calls++;
        break;

    case RT_SH_MATERIAL:
        if (c->mode & ORC_TRACE_ONLY) { group_id = ray->queue[0].group_id = dead; break; }
        group_id = mat_shader(ray, &grp->material, c->shading, c->sample_id[offs], c->mad);
        work->shade_calls++;
        work->shadows++;
        break;

    case RT_SH_AABB: { /* insert_hits, ray-tracer.cl:205-231 */
        const uint32_t n = aabb_shader(&cur, &grp->aabb, &ray->queue[0], new_hit, c->aabb, c->mad);
        if (c->threads <= 1) c->hit_hist[n]++;
        work->box_tests += grp->aabb.aabb_count;
        work->node_visits++;
        orc_sort_hits(new_hit, n);
        queue_len = 0;
        const uint32_t old_len = ray->queue_len;
        uint32_t next = 0;
        int overflow = 0;
        for (uint32_t i = 1; i < old_len && !overflow; i++) {
            const float pos = ray->queue[i].pos;
            while (next < n && new_hit[next].pos < pos) {
                hit[queue_len++] = new_hit[next++];
                if (queue_len == RT_MAX_QUEUE_LEN) { overflow = 1; break; }
            }
            if (overflow) break;
            hit[queue_len].pos = pos;
            if (++queue_len == RT_MAX_QUEUE_LEN) { overflow = 1; break; }
            hit[queue_len - 1].group_id = ray->queue[i].group_id;
            hit[queue_len - 1].local_id[0] = ray->queue[i].local_id[0];
            hit[queue_len - 1].local_id[1] = ray->queue[i].local_id[1];
        }
        while (!overflow && next < n) {
            hit[queue_len++] = new_hit[next++];
            if (queue_len == RT_MAX_QUEUE_LEN) overflow = 1;
        }
        if (overflow) { /* ray-tracer.cl:228-231: the last entry restarts from the root later */
            hit[RT_MAX_QUEUE_LEN - 1].group_id = ray->root.group_id;
            hit[RT_MAX_QUEUE_LEN - 1].local_id[0] = ray->root.local_id[0];
            hit[RT_MAX_QUEUE_LEN - 1].local_id[1] = ray->root.local_id[1];
        }
        group_id = save_queue(ray, hit, queue_len, ray->material_id);
        break;
    }

    case RT_SH_MESH: {
        float norm_pos[4];
        uint32_t hit_tri = 0;
        uint32_t material_id = mesh_shader(&cur, &grp->mesh, norm_pos, c->vtx, c->tri, &hit_tri, c->mad);
        work->tri_tests += grp->mesh.tri_count;
        work->node_visits++;
        if (material_id == 0xFFFFFFFFu) { /* miss: pop the head, ray-tracer.cl:154-165 */
            queue_len = ray->queue_len - 1;
            for (uint32_t i = 0; i < queue_len; i++) hit[i] = ray->queue[i + 1];
            group_id = save_queue(ray, hit, queue_len, ray->material_id);
            break;
        }
        /* insert_stop, ray-tracer.cl:173-203 */
        if ((ray->type & 0xFFu) == RT_RAY_SHADOW) { /* (bounce count above the low byte) */
            area_add(c, ray->pixel, 0, 0, 0, ray->weight[3]);
            material_id = RT_SPAWN_GROUP;
        } else {
            const v3 nw = v3_lin3_m(c->mad, mat[0], norm_pos[0], mat[1], norm_pos[1], mat[2], norm_pos[2]);
            v3_store(ray->norm, nw);
            ray->dir_max[3] = norm_pos[3];
            ray->orig.pos = norm_pos[3];
            ray->orig.group_id = ray->queue[0].group_id;
            ray->orig.local_id[0] = ray->queue[0].local_id[0];
            ray->orig.local_id[1] = ray->queue[0].local_id[1];
            const uint32_t sid = c->sample_id[offs];
            if (ray->type == RT_RAY_PRIMARY && sid < c->hit_capacity) { /* the camera ray's hit; a bounced sample keeps it */
                rt_hit_record rec = {norm_pos[3], ray->queue[0].group_id, ray->queue[0].local_id[0], hit_tri};
                c->hits[sid] = rec;
            }
        }
        queue_len = ray->queue_len - 1;
        for (uint32_t i = 0; i < queue_len; i++) {
            hit[i].pos = ray->queue[i + 1].pos;
            if (hit[i].pos >= norm_pos[3]) { queue_len = i; break; }
            hit[i].group_id = ray->queue[i + 1].group_id;
            hit[i].local_id[0] = ray->queue[i + 1].local_id[0];
            hit[i].local_id[1] = ray->queue[i + 1].local_id[1];
        }
        ray->material_id = material_id; /* ray-tracer.cl:203; save_queue's empty case reads the new value (:201) */
        group_id = save_queue(ray, hit, queue_len, material_id);
        break;
    }

    default: /* unknown shader type: falls through to the pop, ray-tracer.cl:152-165 */
        queue_len = ray->queue_len - 1;
        for (uint32_t i = 0; i < queue_len; i++) hit[i] = ray->queue[i + 1];
        group_id = save_queue(ray, hit, queue_len, ray->material_id);
        break;
    }
    ray_index[index].key = group_id; /* assign_index, ray-tracer.cl:171 */
    ray_index[index].slot = offs;
}

static void k_process(orc_ctx *c, rt_index_pair *ray_index)
{
    const uint32_t n = c->global.ray_count;
    orc_work total = {0, 0, 0, 0, 0, 0};
    if (c->threads <= 1) {
        for (uint32_t index = 0; index < n; index++) process_one(c, index, ray_index, &total);
    } else {
        /* Work-groups of UNIT_WIDTH rays spread over threads.  area[] is updated without atomics, as
         * in the reference (ray-tracer.cl:176, shader.cl:9,16), so this mode is for timing, not for
         * bit-exact checks. */
        uint64_t box = 0, tri = 0, shade = 0, visits = 0, spawned = 0, shadows = 0;
        const int64_t groups = (n + UNIT_WIDTH - 1) / UNIT_WIDTH;
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->threads) reduction(+ : box, tri, shade, visits, spawned, shadows)
        for (int64_t g = 0; g < groups; g++) {
            orc_work w = {0, 0, 0, 0, 0, 0};
            const uint32_t end = (uint32_t)((g + 1) * UNIT_WIDTH < n ? (g + 1) * UNIT_WIDTH : n);
            for (uint32_t index = (uint32_t)(g * UNIT_WIDTH); index < end; index++) process_one(c, index, ray_index, &w);
            box += w.box_tests; tri += w.tri_tests; shade += w.shade_calls; visits += w.node_visits;
            spawned += w.spawned; shadows += w.shadows;
        }
        total.box_tests = box; total.tri_tests = tri; total.shade_calls = shade; total.node_visits = visits;
        total.spawned = spawned; total.shadows = shadows;
    }
    c->work.box_tests += total.box_tests;
    c->work.tri_tests += total.tri_tests;
    c->work.shade_calls += total.shade_calls;
    c->work.node_visits += total.node_visits;
    c->work.spawned += total.spawned;
    c->work.shadows += total.shadows;
    c->counters.primary_spawned += total.spawned;
    c->counters.shadow_spawned += total.shadows;
    c->counters.box_tests += total.box_tests;
    c->counters.tri_tests += total.tri_tests;
}

/* ------------------------------------------------------------------ sort (sort.cl) */

/* sort.cl:72-82: where rank `pos` of a pass's output must be stored so that the next pass, reading
 * its block thread-major, meets the elements in rank order. */
uint32_t orc_sort_block_index(uint32_t pos, uint32_t n)
{
    uint32_t local_pos = pos % SORT_WIDTH;
    pos -= local_pos;
    n = n - pos < SORT_WIDTH ? n - pos : SORT_WIDTH;
    uint32_t block_size = n / UNIT_WIDTH;
    const uint32_t rem = n % UNIT_WIDTH, bound = rem * (block_size + 1);
    if (local_pos < bound) block_size++;
    else { local_pos -= bound; pos += rem; }
    return pos + local_pos / block_size + local_pos % block_size * UNIT_WIDTH;
}

/* sort.cl:19-43 for one block: per-element rank among same-digit elements of the block in
 * thread-major order (local_index) and per-digit totals of the block (global_index). */
static void k_local_count_block(const rt_index_pair *val, uint32_t n, uint32_t *local_index, uint32_t *global_index,
                                uint32_t block, uint32_t shift, uint32_t mask, uint32_t max_val)
{
    const uint32_t offs = block * SORT_WIDTH;
    if (offs >= n) return;
    val += offs; local_index += offs; global_index += block * RADIX_MAX;
    static _Thread_local uint32_t count[UNIT_WIDTH][RADIX_MAX];
    static _Thread_local uint32_t pos[UNIT_WIDTH][SORT_BLOCK];
    static _Thread_local uint32_t size[UNIT_WIDTH];
    for (uint32_t index = 0; index < UNIT_WIDTH; index++) { /* region before the first barrier */
        uint32_t block_size = (n - offs + (UNIT_WIDTH - 1) - index) / UNIT_WIDTH;
        if (block_size > SORT_BLOCK) block_size = SORT_BLOCK;
        size[index] = block_size;
        for (uint32_t i = 0; i < max_val; i++) count[index][i] = 0;
        for (uint32_t i = 0; i < block_size; i++) pos[index][i] = count[index][(val[i * UNIT_WIDTH + index].key >> shift) & mask]++;
    }
    for (uint32_t i = 0; i < max_val; i++) { /* local_scan per digit: exclusive prefix over threads + total */
        uint32_t running = 0;
        for (uint32_t index = 0; index < UNIT_WIDTH; index++) {
            const uint32_t mine = count[index][i];
            count[index][i] = running;
            running += mine;
        }
        global_index[i] = running;
    }
    for (uint32_t index = 0; index < UNIT_WIDTH; index++)
        for (uint32_t i = 0; i < size[index]; i++)
            local_index[i * UNIT_WIDTH + index] = pos[index][i] + count[index][(val[i * UNIT_WIDTH + index].key >> shift) & mask];
}

/* sort.cl:45-70: exclusive scan of global_index in digit-major, block-minor order. */
static void k_global_count(uint32_t *global_index, uint32_t n, uint32_t max_val)
{
    const uint32_t block_count = (n + SORT_WIDTH - 1) / SORT_WIDTH;
    uint32_t running = 0;
    for (uint32_t i = 0; i < max_val; i++)
        for (uint32_t block = 0; block < block_count; block++) {
            const uint32_t k = block * RADIX_MAX + i, mine = global_index[k];
            global_index[k] = running;
            running += mine;
        }
}

/* sort.cl:84-102 for one block. */
static void k_shuffle_block(const rt_index_pair *src, rt_index_pair *dst, uint32_t n, const uint32_t *local_index,
                            const uint32_t *global_index, uint32_t block, uint32_t shift, uint32_t mask, uint32_t last)
{
    const uint32_t offs = block * SORT_WIDTH;
    src += offs; local_index += offs; global_index += block * RADIX_MAX;
    for (uint32_t index = 0; index < UNIT_WIDTH; index++) {
        uint32_t block_size = ((offs > n ? offs : n) - offs + (UNIT_WIDTH - 1) - index) / UNIT_WIDTH;
        if (block_size > SORT_BLOCK) block_size = SORT_BLOCK;
        for (uint32_t i = 0; i < block_size; i++) {
            const rt_index_pair data = src[i * UNIT_WIDTH + index];
            const uint32_t pos = global_index[(data.key >> shift) & mask] + local_index[i * UNIT_WIDTH + index];
            dst[last ? pos : orc_sort_block_index(pos, n)] = data;
        }
        for (uint32_t i = block_size; i < SORT_BLOCK; i++) {
            const uint32_t pos = i * UNIT_WIDTH + index;
            dst[offs + pos] = src[pos];
        }
    }
}

/* The pass loop of make_step (main.cpp:444-463).  buf[0] holds the input; returns the index of the
 * buffer holding the result. */
static int sort_passes(rt_index_pair *buf[2], uint32_t count, uint32_t n, uint32_t group_count, uint32_t *local_index,
                       uint32_t *global_index, int threads)
{
    const uint32_t block_count = count / SORT_WIDTH;
    int cur = 0;
    (void)threads;
    for (uint32_t shift = 0, mask = RT_GROUP_ID_MASK, max = group_count - 1; max;
         shift += RADIX_SHIFT, mask >>= RADIX_SHIFT, max >>= RADIX_SHIFT) {
        const uint32_t max_val = max + 1 < RADIX_MAX ? max + 1 : RADIX_MAX;
        const uint32_t last = max < (1u << RADIX_SHIFT);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (threads > 1)
        for (int64_t block = 0; block < (int64_t)block_count; block++)
            k_local_count_block(buf[cur], n, local_index, global_index, (uint32_t)block, shift, mask & RADIX_MASK, max_val);
        k_global_count(global_index, n, max_val);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (threads > 1)
        for (int64_t block = 0; block < (int64_t)block_count; block++)
            k_shuffle_block(buf[cur], buf[cur ^ 1], n, local_index, global_index, (uint32_t)block, shift, mask & RADIX_MASK, last);
        cur ^= 1;
    }
    return cur;
}

void orc_sort_pairs(const rt_index_pair *in, rt_index_pair *out, uint32_t count, uint32_t n, uint32_t group_count)
{
    rt_index_pair *buf[2];
    buf[0] = malloc((size_t)count * sizeof(rt_index_pair));
    buf[1] = malloc((size_t)count * sizeof(rt_index_pair));
    uint32_t *local_index = malloc((size_t)count * sizeof(uint32_t));
    uint32_t *global_index = calloc((size_t)(count / SORT_WIDTH) * RADIX_MAX, sizeof(uint32_t));
    memcpy(buf[0], in, (size_t)count * sizeof(rt_index_pair));
    memcpy(buf[1], in, (size_t)count * sizeof(rt_index_pair));
    const int cur = sort_passes(buf, count, n, group_count, local_index, global_index, 1);
    memcpy(out, buf[cur], (size_t)count * sizeof(rt_index_pair));
    free(buf[0]); free(buf[1]); free(local_index); free(global_index);
}

/* ------------------------------------------------------------------ compaction (ray-tracer.cl:235-302) */

/* ray-tracer.cl:235-253: count.s0[g] = number of sorted entries whose group id is <= g. */
static void k_count_groups(const rt_global_data *data, rt_group_data *grp_data, const rt_index_pair *ray_index)
{
    const uint32_t total = data->ray_count;
    for (uint32_t offs = 0; offs < total; offs++) {
        uint32_t prev, next;
        if (offs) prev = ray_index[offs - 1].key;
        else {
            prev = ray_index[total - 1].key & RT_GROUP_ID_MASK;
            next = data->group_count;
            for (; prev < next; prev++) grp_data[prev].count[0] = total;
            prev = 0;
        }
        prev &= RT_GROUP_ID_MASK;
        next = ray_index[offs].key & RT_GROUP_ID_MASK;
        for (; prev < next; prev++) grp_data[prev].count[0] = offs;
    }
}

/* ray-tracer.cl:255-291: per group, split (newly sorted + previously parked) rays into a part that
 * fills whole warps and a parked remainder; prefix sums give both layouts. */
static void k_update_groups(rt_global_data *data, rt_group_data *grp_data, uint32_t warp_width)
{
    const uint32_t n = data->group_count;
    uint32_t offset[2] = {0, 0}, prev = 0;
    for (uint32_t pos = 0; pos < n; pos++) {
        rt_group_data grp;
        grp.count[0] = grp_data[pos].count[0];
        grp.base[0] = prev;
        prev = grp.count[0];
        grp.count[0] -= grp.base[0];
        grp.base[1] = grp_data[pos].offset[1] - grp.count[0];
        grp.count[0] += grp_data[pos].count[1];
        grp.count[1] = grp.count[0];
        if (pos != n - 1) grp.count[1] %= warp_width;
        grp.count[0] -= grp.count[1];
        if (!pos) {
            data->pixel_offset += data->pixel_count;
            data->pixel_count = grp.count[0];
        }
        grp.offset[0] = offset[0];
        grp.offset[1] = offset[1];
        offset[0] += grp.count[0];
        offset[1] += grp.count[1];
        grp_data[pos] = grp;
    }
    for (uint32_t pos = 0; pos < n; pos++) grp_data[pos].offset[1] += offset[0];
    data->old_count = data->ray_count;
    data->ray_count = offset[0];
}

/* ray-tracer.cl:293-302 */
static void k_set_ray_index(const rt_global_data *data, const rt_group_data *grp_data, const rt_index_pair *src_index,
                            rt_index_pair *dst_index, uint32_t count)
{
    for (uint32_t i = 0; i < count; i++) {
        uint32_t index = i;
        const rt_index_pair src = src_index[index];
        const rt_group_data grp = grp_data[src.key & RT_GROUP_ID_MASK];
        index -= (index < data->old_count ? grp.base[0] : grp.base[1]);
        if (index < grp.count[1]) index += grp.offset[1];
        else index += grp.offset[0] - grp.count[1];
        /* (the reference stores unconditionally, ray-tracer.cl:301; the bound only matters for group data that does not
         * belong to the index given -- caller data of the stage entry point -- and keeps such a call from writing wild) */
        if (index < count) dst_index[index] = src;
    }
}

void orc_compact_groups(const rt_index_pair *sorted, rt_index_pair *out, uint32_t count, rt_group_data *grp_data,
                        rt_global_data *global, uint32_t warp_width)
{
    if (global->ray_count) k_count_groups(global, grp_data, sorted);
    else for (uint32_t g = 0; g < global->group_count; g++) grp_data[g].count[0] = 0;
    k_update_groups(global, grp_data, warp_width);
    k_set_ray_index(global, grp_data, sorted, out, count);
}

/* ------------------------------------------------------------------ frame kernels */

/* ray-tracer.cl:22-26, 76-87 */
void orc_init_frame(orc_ctx *c)
{
    for (uint32_t g = 0; g < c->group_count; g++) {
        c->grp_data[g].count[1] = 0;
        c->grp_data[g].offset[1] = 0xFFFFFFFFu;
    }
    c->global.ray_count = c->pool;
    for (uint32_t index = 0; index < c->pool; index++) {
        c->ray_index[0][index].key = init_ray(c, index, index);
        c->ray_index[0][index].slot = index;
    }
    c->global.pixel_offset = c->pool;
    c->global.pixel_count = 0;
    memset(c->area, 0, (size_t)c->area_size * 4 * sizeof(float));
    memset(&c->counters, 0, sizeof(c->counters));
    memset(&c->work, 0, sizeof(c->work));
    c->counters.primary_spawned = c->work.spawned = c->sample_limit && c->sample_limit < c->pool ? c->sample_limit : c->pool;
}

/* main.cpp:439-477.  The result of a step is always left in ray_index[0]. */
void orc_step(orc_ctx *c, uint32_t n_steps)
{
    for (uint32_t s = 0; s < n_steps; s++) {
        k_process(c, c->ray_index[0]);
        const int cur = sort_passes(c->ray_index, c->pool, c->global.ray_count, c->group_count, c->local_index,
                                    c->global_index, c->threads);
        if (cur) { rt_index_pair *t = c->ray_index[0]; c->ray_index[0] = c->ray_index[1]; c->ray_index[1] = t; }
        if (c->global.ray_count) k_count_groups(&c->global, c->grp_data, c->ray_index[0]);
        else for (uint32_t g = 0; g < c->group_count; g++) c->grp_data[g].count[0] = 0;
        k_update_groups(&c->global, c->grp_data, c->warp_width);
        k_set_ray_index(&c->global, c->grp_data, c->ray_index[0], c->ray_index[1], c->pool);
        { rt_index_pair *t = c->ray_index[0]; c->ray_index[0] = c->ray_index[1]; c->ray_index[1] = t; }
        c->counters.steps++;
    }
}

void orc_set_warp_width(orc_ctx *c, uint32_t warp_width) { c->warp_width = warp_width ? warp_width : 64; }

/* ray-tracer.cl:305-309 with the RGBA8 conversion write_imagef performs on a UNORM8 image. */
void orc_resolve_rgba8(orc_ctx *c, uint8_t *dst)
{
    for (uint32_t index = 0; index < c->area_size; index++) {
        const float *px = &c->area[4 * (size_t)index];
        const float denom = px[3] + 1e-6f;
        for (int k = 0; k < 4; k++) {
            float v = powf(px[k] / denom, 1 / 2.2f);
            v = v < 0 ? 0 : v > 1 ? 1 : v;
            if (v != v) v = 0;
            dst[4 * (size_t)index + k] = (uint8_t)lrintf(v * 255.0f);
        }
    }
}

/* ------------------------------------------------------------------ context */

orc_ctx *orc_create(const rt_config *cfg)
{
    if (!cfg || !cfg->width || !cfg->height || !cfg->ray_count) return NULL;
    orc_ctx *c = calloc(1, sizeof(*c));
    c->width = cfg->width; c->height = cfg->height;
    c->area_size = cfg->width * cfg->height;
    c->pool = (cfg->ray_count + SORT_WIDTH - 1) / SORT_WIDTH * SORT_WIDTH;
    c->block_count = c->pool / SORT_WIDTH;
    c->warp_width = cfg->warp_width ? cfg->warp_width : 64;
    c->sample_limit = cfg->sample_limit;
    c->hit_capacity = cfg->hit_capacity;
    c->row_offset = cfg->row_offset;
    c->shading = cfg->flags >> 8;
    c->mad = (cfg->flags & RT_FLAG_ARITH_MAD) != 0;
    c->threads = 1;
    c->area = calloc((size_t)c->area_size * 4, sizeof(float));
    c->ray_list = calloc(c->pool, sizeof(rt_ray_queue));
    c->ray_index[0] = calloc(c->pool, sizeof(rt_index_pair));
    c->ray_index[1] = calloc(c->pool, sizeof(rt_index_pair));
    c->local_index = calloc(c->pool, sizeof(uint32_t));
    c->global_index = calloc((size_t)c->block_count * RADIX_MAX, sizeof(uint32_t));
    c->sample_id = calloc(c->pool, sizeof(uint32_t));
    c->hits = calloc(c->hit_capacity ? c->hit_capacity : 1, sizeof(rt_hit_record));
    return c;
}

void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    free(c->area); free(c->ray_list); free(c->ray_index[0]); free(c->ray_index[1]); free(c->local_index);
    free(c->global_index); free(c->sample_id); free(c->hits); free(c->grp_data);
    free(c->grp_list); free(c->mat_list); free(c->aabb); free(c->vtx); free(c->tri);
    free(c);
}

void orc_set_threads(orc_ctx *c, int threads) { c->threads = threads < 1 ? 1 : threads; }

static void *dup_array(const void *src, size_t count, size_t size, size_t min_count)
{
    const size_t total = count > min_count ? count : min_count;
    void *dst = calloc(total ? total : 1, size);
    if (count) memcpy(dst, src, count * size);
    return dst;
}

int orc_upload_scene(orc_ctx *c, const rt_global_data *data, const rt_group *grp, size_t n_groups, const rt_matrix *mat,
                     size_t n_mat, const rt_aabb *aabb, size_t n_aabb, const rt_vertex *vtx, size_t n_vtx,
                     const uint32_t *tri, size_t n_tri)
{
    if (!data || data->cam.width * data->cam.height != c->area_size || n_groups > data->group_count) return RT_ERR_INVALID_VALUE;
    c->global = *data;
    c->global.ray_count = c->pool;
    c->group_count = data->group_count;
    free(c->grp_data); free(c->grp_list); free(c->mat_list); free(c->aabb); free(c->vtx); free(c->tri);
    c->grp_data = calloc(c->group_count, sizeof(rt_group_data));
    c->grp_list = dup_array(grp, n_groups, sizeof(rt_group), c->group_count);
    c->mat_list = dup_array(mat, n_mat, sizeof(rt_matrix), 0);
    c->aabb = dup_array(aabb, n_aabb, sizeof(rt_aabb), 0);
    c->vtx = dup_array(vtx, n_vtx, sizeof(rt_vertex), 0);
    c->tri = dup_array(tri, n_tri, sizeof(uint32_t), 0);
    c->n_groups = n_groups; c->n_mat = n_mat; c->n_aabb = n_aabb; c->n_vtx = n_vtx; c->n_tri = n_tri;
    return RT_OK;
}

rt_global_data *orc_global(orc_ctx *c) { return &c->global; }
float *orc_area(orc_ctx *c) { return c->area; }
rt_ray_queue *orc_ray_list(orc_ctx *c) { return c->ray_list; }
rt_index_pair *orc_ray_index(orc_ctx *c) { return c->ray_index[0]; }
rt_group_data *orc_grp_data(orc_ctx *c) { return c->grp_data; }
rt_hit_record *orc_hits(orc_ctx *c) { return c->hits; }
uint32_t orc_pool_size(orc_ctx *c) { return c->pool; }
void orc_get_counters(orc_ctx *c, rt_counters *dst) { *dst = c->counters; }
void orc_get_work(orc_ctx *c, orc_work *dst) { *dst = c->work; }
void orc_get_hit_histogram(orc_ctx *c, uint64_t *dst) { memcpy(dst, c->hit_hist, sizeof(c->hit_hist)); }

/* The camera ray of sample number `sample`, as init_ray sets it up (ray-tracer.cl:56-74): for tests that trace a
 * frame's own samples through orc_trace_closest. */
void orc_camera_rays(orc_ctx *c, const uint32_t *samples, uint32_t n, rt_ray *out)
{
    const rt_camera *cam = &c->global.cam;
    const uint32_t total = cam->width * cam->height;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t sub[2];
        orc_deinterleave(samples[i] / total, sub);
        const uint32_t pixel = samples[i] % total;
        const float x = (float)(pixel % cam->width) + orc_subpixel(sub[0]);
        const float y = (float)(pixel / cam->width + c->row_offset) + orc_subpixel(sub[1]);
        v3_store(out[i].start_min, v3_load(cam->eye));
        out[i].start_min[3] = 0.001f;
        v3_store(out[i].dir_max, camera_dir(c->mad, cam, x, y));
        out[i].dir_max[3] = INFINITY;
    }
}

/* Starts a frame in which slot i runs sample number samples[i] (i < n) and every other slot is parked in the
 * dead-ray bin; no further samples are started (finished slots find the sample range used up).  Needs a
 * sample_limit above every id given.  The frame then advances with orc_step like any other: shading, shadow
 * rays and bounces included.  Used to follow single samples, e.g. those that never finish. */
int orc_init_frame_samples(orc_ctx *c, const uint32_t *samples, uint32_t n)
{
    if (n > c->pool || !c->grp_list || !c->sample_limit) return RT_ERR_INVALID_VALUE;
    for (uint32_t i = 0; i < n; i++)
        if (samples[i] >= c->sample_limit) return RT_ERR_INVALID_VALUE;
    for (uint32_t g = 0; g < c->group_count; g++) { c->grp_data[g].count[1] = 0; c->grp_data[g].offset[1] = 0xFFFFFFFFu; }
    c->global.ray_count = c->pool;
    const uint32_t dead = c->group_count - 1;
    for (uint32_t index = 0; index < c->pool; index++) {
        c->ray_index[0][index].key = index < n ? init_ray(c, index, samples[index]) : dead;
        c->ray_index[0][index].slot = index;
    }
    c->global.pixel_offset = c->sample_limit; c->global.pixel_count = 0;
    memset(c->area, 0, (size_t)c->area_size * 4 * sizeof(float));
    memset(&c->counters, 0, sizeof(c->counters));
    memset(&c->work, 0, sizeof(c->work));
    c->counters.primary_spawned = c->work.spawned = n;
    rt_index_pair *sorted = malloc((size_t)c->pool * sizeof(rt_index_pair));
    orc_sort_pairs(c->ray_index[0], sorted, c->pool, c->pool, c->group_count);
    orc_compact_groups(sorted, c->ray_index[0], c->pool, c->grp_data, &c->global, c->warp_width);
    free(sorted);
    return RT_OK;
}

int orc_trace_closest(orc_ctx *c, const rt_ray *rays, uint32_t n_rays, rt_hit_record *hits)
{
    if (n_rays > c->pool || !c->grp_list) return RT_ERR_INVALID_VALUE;
    const uint32_t saved_limit = c->sample_limit, saved_cap = c->hit_capacity, saved_warp = c->warp_width;
    rt_hit_record *saved_hits = c->hits;
    c->hits = hits; c->hit_capacity = n_rays; c->sample_limit = n_rays ? n_rays : 1; c->warp_width = 1;
    c->mode |= ORC_TRACE_ONLY;

    for (uint32_t g = 0; g < c->group_count; g++) { c->grp_data[g].count[1] = 0; c->grp_data[g].offset[1] = 0xFFFFFFFFu; }
    c->global.ray_count = c->pool;
    const uint32_t dead = c->group_count - 1;
    for (uint32_t index = 0; index < c->pool; index++) {
        uint32_t key = dead;
        if (index < n_rays) {
            key = init_ray(c, index, index); /* root, ids, queue, end group; then the caller's geometry */
            memcpy(c->ray_list[index].start_min, rays[index].start_min, sizeof(float) * 4);
            memcpy(c->ray_list[index].dir_max, rays[index].dir_max, sizeof(float) * 4);
            c->ray_list[index].queue[0].pos = rays[index].start_min[3];
        }
        c->ray_index[0][index].key = key;
        c->ray_index[0][index].slot = index;
    }
    c->global.pixel_offset = c->pool; c->global.pixel_count = 0;
    /* park the unused slots: one compaction pass over the initial keys */
    {
        rt_index_pair *sorted = malloc((size_t)c->pool * sizeof(rt_index_pair));
        orc_sort_pairs(c->ray_index[0], sorted, c->pool, c->pool, c->group_count);
        orc_compact_groups(sorted, c->ray_index[0], c->pool, c->grp_data, &c->global, 1);
        free(sorted);
    }
    int status = RT_OK;
    for (uint32_t guard = 0; c->global.ray_count; guard++) {
        if (guard > 100000) { status = RT_ERR_NOT_READY; break; }
        orc_step(c, 1);
    }
    c->mode &= ~ORC_TRACE_ONLY;
    c->hits = saved_hits; c->hit_capacity = saved_cap; c->sample_limit = saved_limit; c->warp_width = saved_warp;
    return status;
}
