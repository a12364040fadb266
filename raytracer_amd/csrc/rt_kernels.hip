// rt_kernels.hip -- gfx950 kernels for traversal/intersection, shading/spawn, queue compaction and
// frame set-up/resolve.
//
// What they compute is what the reference's `process` kernel and its helpers compute
// (ray-tracer.cl:10-232, shader.cl:6-139) and what count_groups / update_groups / set_ray_index
// compute (ray-tracer.cl:235-302).  How they compute it is specific to CDNA4:
//   * rays arrive sorted by group, so a wave's lanes normally sit in ONE group; the group's payload
//     (box list, expanded triangles) is wave-uniform and is fetched with scalar loads from the
//     constant address space, while each lane keeps its own ray in VGPRs;
//   * a wave that does hold several groups (unaligned tails, drain mode) peels them off one at a time
//     (`while (todo) { g = readfirstlane(key); ... }`), so the kernels are correct for any order;
//   * the candidate boxes of a visit live in LDS as [entry][thread] columns: any per-lane entry index
//     maps lane -> bank (tid mod 32), conflict-free for 4-byte accesses; the box kernel keeps 16 per
//     lane, the rare visit that needs the reference's 64 is redone by a whole wave (second kernel);
//   * the pending-node queue of a ray is merged straight into its slot in HBM from a register
//     snapshot of the old entries -- no scratch memory;
//   * box groups, leaf groups and shader groups run as three kernels with different register/LDS
//     budgets; all read the step's keys from `index` and write the new ones to `next`, so a ray
//     advanced by one kernel is never picked up again by another in the same step, and the leaf
//     kernel can run beside the other two.
#include "rt_device.h"

namespace rtd {

// Phase stamps of diagnostic builds: -DRT_STAMPS=1 times the box-group kernel, =2 the leaf and shading
// kernels, =3 records start and end of every wave of the box kernel.
#if defined(RT_STAMPS) && RT_STAMPS == 1
#define AABB_STAMP(v) RT_STAMP(v)
#define AABB_STAMP_ADD(f, s, a, b) RT_STAMP_ADD(f, s, a, b)
#else
#define AABB_STAMP(v)
#define AABB_STAMP_ADD(f, s, a, b)
#endif
// =4: inside the merge of the box kernel: child ids landed, pending entries landed, ranks and stores issued, whole visit
#if defined(RT_STAMPS) && RT_STAMPS == 4
#define MERGE_STAMP(v) RT_STAMP(v)
#define MERGE_STAMP_ADD(f, s, a, b) RT_STAMP_ADD(f, s, a, b)
#else
#define MERGE_STAMP(v)
#define MERGE_STAMP_ADD(f, s, a, b)
#endif
#if defined(RT_STAMPS) && RT_STAMPS == 2
#define MISC_STAMP(v) RT_STAMP(v)
#define MISC_STAMP_ADD(f, s, a, b) RT_STAMP_ADD(f, s, a, b)
#else
#define MISC_STAMP(v)
#define MISC_STAMP_ADD(f, s, a, b)
#endif

struct ProcessArgs {
    Scene scene;
    Frame frame;
    rt_camera cam;
    uint32_t group_count;
};

// ---------------------------------------------------------------- ray set-up (ray-tracer.cl:10-74)

__device__ __forceinline__ void deinterleave(uint32_t val, uint32_t &a, uint32_t &b)
{
    a = val & 0x55555555u;
    b = (val >> 1) & 0x55555555u;
    a = (a | a >> 1) & 0x33333333u;  b = (b | b >> 1) & 0x33333333u;
    a = (a | a >> 2) & 0x0F0F0F0Fu;  b = (b | b >> 2) & 0x0F0F0F0Fu;
    a = (a | a >> 4) & 0x00FF00FFu;  b = (b | b >> 4) & 0x00FF00FFu;
    a = (a | a >> 8) & 0x0000FFFFu;  b = (b | b >> 8) & 0x0000FFFFu;
}

__device__ __forceinline__ float subpixel(uint32_t val)
{
    val = (val % 15) + 1;
    val = (val & 1) << 3 | (val & 2) << 1 | (val & 4) >> 1 | (val & 8) >> 3;
    return ((float)val - 0.5f) / 15.0f;
}

// reset_ray (ray-tracer.cl:10-16) for a ray whose hot fields are in registers: the first 64 bytes of the
// slot go out as four 16-byte requests.  LEAN (the free schedule's kernels): `orig` is not kept up -- no
// kernel reads it, the hit record holds what was hit.
template <bool LEAN>
__device__ __forceinline__ uint32_t reset_ray(RaySlot *ray, HotRay h, uint32_t group_id, uint32_t local0, uint32_t local1,
                                              uint32_t end_group)
{
    h.t_min = 0.001f;
    h.t_max = INFINITY;
    h.queue_len = 1;
    h.material_id = end_group;
    store_hot(ray, h);
    ray->queue[0] = RayHit{0.001f, group_id, local0, local1};
    if (!LEAN) {
        ray->orig.group_id = group_id;
        ray->orig.local0 = local0;
        ray->orig.local1 = local1;
    }
    return group_id;
}

// The hot fields of sample number `sample` as init_ray and reset_ray set them (ray-tracer.cl:10-16, 56-74):
// a camera ray through the sample's sub-pixel position, about to enter the camera's root group.
template <bool MAD>
__device__ __forceinline__ HotRay camera_ray(const rt_camera &cam, const DevGlobal *glob, uint32_t sample)
{
    const uint32_t total = cam.width * cam.height;
    uint32_t sx, sy;
    deinterleave(sample / total, sx, sy);
    const uint32_t pixel = sample % total;
    const float x = (float)(pixel % cam.width) + subpixel(sx);
    const float y = (float)(pixel / cam.width + step_constant(&glob->row_offset)) + subpixel(sy);
    const V3 tl = v3(cam.top_left[0], cam.top_left[1], cam.top_left[2]);
    const V3 dx = v3(cam.dx[0], cam.dx[1], cam.dx[2]), dy = v3(cam.dy[0], cam.dy[1], cam.dy[2]);
    HotRay h;
    h.pixel = pixel;
    h.sample_id = sample;
    h.material_id = RT_SKY_GROUP;
    h.type = RT_RAY_PRIMARY;
    h.queue_len = 1;
    h.t_min = 0.001f;
    h.t_max = INFINITY;
    h.start = v3(cam.eye[0], cam.eye[1], cam.eye[2]);
    h.dir = Arith<MAD>::normalize(Arith<MAD>::madd(dy, y, Arith<MAD>::madd(dx, x, tl)));
    return h;
}

// Which of the samples [first, first + count) the finished slot at position `index` of the re-packed index starts
// (free schedule).  The reference hands them out in index order (first + index, ray-tracer.cl:149), which puts 64
// pixels of one image row into a wave.  Here every aligned run of eight image rows inside the range is handed out
// in blocks of 8 x 8 pixels instead: a wave's rays stay close together on their way down the hierarchy, so
// fewer of a box list's clusters are entered by some lane of the wave.  A permutation of the range: every sample is
// started exactly once either way, and what a sample computes does not depend on the slot that runs it.
__device__ __forceinline__ uint32_t blocked_sample(const rt_camera &cam, uint32_t first, uint32_t count, uint32_t index)
{
    const uint32_t sample = first + index, width = cam.width, chunk = 8 * width;
    if (width % 8 || index >= count) return sample;
    const uint32_t start = sample / chunk * chunk;  // the run of eight rows this sample lies in, if whole and inside the range
    if (start < first || start + chunk > first + count || start + chunk < start) return sample;
    const uint32_t t = sample - start, block = t / 64, lane = t % 64;
    return start + (lane / 8) * width + 8 * block + lane % 8;
}

// init_ray (ray-tracer.cl:56-74).  Sample ids at or beyond the limit take the dead-ray exit the
// reference left commented out (ray-tracer.cl:61): the slot is left alone and parked in the last group.
// LEAN: `root` is not stored again -- every slot of the pool got the camera's root when the frame was
// initialised, and nothing else ever writes it -- and neither is the weight: a camera ray that has not
// bounced has weight one, which is what ray_weight() gives for it without looking at the slot.
template <bool LEAN, bool MAD>
__device__ __forceinline__ uint32_t init_ray(const rt_camera &cam, const DevGlobal *glob, RaySlot *ray,
                                             rt_hit_record *hits, uint32_t group_count, uint32_t sample, bool &started)
{
    const uint32_t limit = step_constant(&glob->sample_limit);
    started = !(limit && sample >= limit);
    if (!started) return group_count - 1;
    if (!LEAN) ray->weight = make_float4(1, 1, 1, 1);
    if (!LEAN) ray->root = RayHit{0.0f, cam.root_group, cam.root_local, 0u};
    if (sample < step_constant(&glob->hit_capacity)) hits[sample] = rt_hit_record{INFINITY, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    return reset_ray<LEAN>(ray, camera_ray<MAD>(cam, glob, sample), cam.root_group, cam.root_local, 0u, RT_SKY_GROUP);
}

// One atomic per wave for a per-lane event count (each counted lane adds `weight`, wave-uniform), to
// the counter shard of this workgroup.
__device__ __forceinline__ void count_lanes(const Frame &frame, Counter which, bool flag, uint32_t weight = 1)
{
    const unsigned long long mask = __ballot(flag);
    if (flag && lane_id() == (uint32_t)(__ffsll((long long)mask) - 1))
        atomicAdd(&frame.counters[(blockIdx.x % kCounterShards) * kCounterKinds + which], (unsigned long long)__popcll(mask) * weight);
}

// ---------------------------------------------------------------- frame kernels

// init_groups (ray-tracer.cl:22-26)
__global__ void k_init_groups(rt_group_data *grp_data, uint32_t group_count)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= group_count) return;
    grp_data[g].count[1] = 0;
    grp_data[g].offset[1] = 0xFFFFFFFFu;
}

// init_rays (ray-tracer.cl:76-82)
template <bool MAD>
__global__ void k_init_rays(ProcessArgs a, uint32_t pool)
{
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= pool) return;
    bool started;
    const uint32_t key = init_ray<false, MAD>(a.cam, a.frame.glob, &a.frame.rays[index], a.frame.hits, a.group_count, index, started);
    a.frame.index[index] = rt_index_pair{key, index};
    count_lanes(a.frame, kPrimarySpawned, started);
    if (index) return;
    a.frame.glob->g.pixel_offset = pool;
    a.frame.glob->g.pixel_count = 0;
    a.frame.glob->g.ray_count = pool;
    a.frame.glob->spill_count = 0;
}

// Loads caller rays as samples 0..n_rays-1 (rt_trace_closest); the remaining slots are parked.
__global__ void k_load_rays(ProcessArgs a, uint32_t pool, const rt_ray *rays, uint32_t n_rays)
{
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= pool) return;
    uint32_t key = a.group_count - 1;
    if (index < n_rays) {
        bool started;
        RaySlot *ray = &a.frame.rays[index];
        key = init_ray<false, false>(a.cam, a.frame.glob, ray, a.frame.hits, a.group_count, index, started);
        const rt_ray r = rays[index];
        asm volatile("" ::: "memory");  // the caller's ray goes over what init_ray has stored
        ray->t_min = r.start_min[3];
        ray->t_max = r.dir_max[3];
        for (int k = 0; k < 3; k++) {
            ray->start[k] = r.start_min[k];
            ray->dir[k] = r.dir_max[k];
        }
        ray->queue[0].pos = r.start_min[3];
    }
    a.frame.index[index] = rt_index_pair{key, index};
    if (index) return;
    a.frame.glob->g.pixel_offset = pool;
    a.frame.glob->g.pixel_count = 0;
    a.frame.glob->g.ray_count = pool;
    a.frame.glob->spill_count = 0;
}

// update_image (ray-tracer.cl:305-309), writing RGBA8 the way write_imagef converts to UNORM8.
__global__ void k_resolve_rgba8(const float4 *area, uchar4 *out, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4 c = area[i];
    const float denom = c.w + 1e-6f;
    const float e = 1 / 2.2f;
    float v[4] = {powf(c.x / denom, e), powf(c.y / denom, e), powf(c.z / denom, e), powf(c.w / denom, e)};
    unsigned char b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float x = v[k];
        x = x != x ? 0.0f : fminf(fmaxf(x, 0.0f), 1.0f);
        b[k] = (unsigned char)__float2int_rn(x * 255.0f);
    }
    out[i] = make_uchar4(b[0], b[1], b[2], b[3]);
}

// ---------------------------------------------------------------- transform (shader.cl:43-73)

struct LocalRay {
    V3 start, dir;
    float tmin, tmax;
    V3 m0, m1, m2;  // columns of the instance rotation (identity for world-space groups)
};

template <bool MAD>
__device__ __forceinline__ LocalRay make_local(uint32_t g, const HotRay &h, uint32_t local0, const rt_matrix *mat_list)
{
    LocalRay r;
    r.start = h.start;
    r.dir = h.dir;
    r.tmin = h.t_min;
    r.tmax = h.t_max;
    r.m0 = v3(1, 0, 0);
    r.m1 = v3(0, 1, 0);
    r.m2 = v3(0, 0, 1);
    if (((g >> RT_GROUP_TR_SHIFT) & RT_GROUP_TR_MASK) == RT_TR_ORTHO) {
        const float4 *m = reinterpret_cast<const float4 *>(&mat_list[local0]);
        const float4 mx = m[0], my = m[1], mz = m[2];
        r.m0 = v3(mx.x, my.x, mz.x);
        r.m1 = v3(mx.y, my.y, mz.y);
        r.m2 = v3(mx.z, my.z, mz.z);
        const V3 rel = r.start - v3(mx.w, my.w, mz.w);
        const V3 vx = xyz(mx), vy = xyz(my), vz = xyz(mz);
        r.start = Arith<MAD>::lin3(vx, rel.x, vy, rel.y, vz, rel.z);
        const V3 d = r.dir;
        r.dir = Arith<MAD>::lin3(vx, d.x, vy, d.y, vz, d.z);
    }
    return r;
}

// ---------------------------------------------------------------- queue helpers

// Snapshot of the pending entries queue[1..queue_len) in registers (static indices only).
struct OldQueue {
    RayHit e[RT_MAX_QUEUE_LEN];
};

__device__ __forceinline__ void load_pending(OldQueue &old, const RaySlot *ray, uint32_t queue_len)
{
#pragma unroll
    for (int i = 1; i < RT_MAX_QUEUE_LEN; i++) {
        old.e[i] = RayHit{0.0f, 0u, 0u, 0u};
        if ((uint32_t)i < queue_len) old.e[i] = ray->queue[i];
    }
}

// save_queue + copy_queue for a queue that is either the shifted old one or empty
// (ray-tracer.cl:154-168 and 184-203).  Returns the new sort key.
__device__ __forceinline__ uint32_t store_shifted(RaySlot *ray, const OldQueue &old, uint32_t type, uint32_t len, float head_pos,
                                                  uint32_t end_group)
{
#pragma unroll
    for (int i = 0; i < RT_MAX_QUEUE_LEN - 1; i++)
        if ((uint32_t)i < len) ray->queue[i] = old.e[i + 1];
    if (len) {
        store_progress(ray, type, len, head_pos);
        return old.e[1].group_id;
    }
    // Empty: the head becomes the ray's end group.  queue[0].pos keeps its value (the reference leaves
    // it indeterminate, ray-tracer.cl:164,201), queue_len and t_min are left alone.
    ray->queue[0].group_id = end_group;
    ray->queue[0].local0 = 0;
    ray->queue[0].local1 = 0;
    return end_group;
}

// ---------------------------------------------------------------- shaders (shader.cl:6-40)

__device__ __forceinline__ void area_add(float4 *area, uint32_t pixel, float r, float g, float b, float w)
{
    float *px = reinterpret_cast<float *>(&area[pixel]);
    unsafeAtomicAdd(px + 0, r);
    unsafeAtomicAdd(px + 1, g);
    unsafeAtomicAdd(px + 2, b);
    unsafeAtomicAdd(px + 3, w);
}

// The colour weight of a ray: one for a camera ray that has not bounced, else what the slot holds.
__device__ __forceinline__ float4 ray_weight(const RaySlot *ray, uint32_t type)
{
    float4 w = make_float4(1, 1, 1, 1);
    if (type != RT_RAY_PRIMARY) w = ray->weight;
    return w;
}

// LEAN (free schedule): the slot of a finished sample is not marked; its key says it all and the next
// sample overwrites the queue.
template <bool LEAN>
__device__ __forceinline__ uint32_t sky_or_light_shader(const ProcessArgs &a, RaySlot *ray, bool is_light)
{
    const uint4 meta = *reinterpret_cast<const uint4 *>(ray);  // pixel, sample, end group, type and queue length
    const float4 w = ray_weight(ray, meta.w & 0xFFFFFFu);
    const float cr = is_light ? 1.0f : 0.5f;
    area_add(a.frame.area, meta.x, w.x * cr, w.y * 1.0f, w.z * 1.0f, w.w * 1.0f);
    if (!LEAN) ray->queue[0].group_id = RT_SPAWN_GROUP;
    return RT_SPAWN_GROUP;
}

// Free schedule: the same for a ray whose visit has just emptied its queue, from what that visit holds in registers
// (pixel and type; the weight comes from the slot unless it is the implied one).  True if `end_group` was the sky
// or the light and the ray is done with; otherwise nothing has happened and the caller goes the general way.
__device__ __forceinline__ bool finish_in_sky_or_light(const ProcessArgs &a, const RaySlot *ray, uint32_t end_group, uint32_t pixel,
                                                       uint32_t type)
{
    const uint32_t shader = (end_group >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK;
    if (shader != RT_SH_SKY && shader != RT_SH_LIGHT) return false;
    if (step_constant(&a.frame.glob->mode) & kModeTraceOnly) return false;
    const float4 w = ray_weight(ray, type);
    const float cr = shader == RT_SH_LIGHT ? 1.0f : 0.5f;
    area_add(a.frame.area, pixel, w.x * cr, w.y * 1.0f, w.z * 1.0f, w.w * 1.0f);
    return true;
}

// A sample that bounces instead of being shaded (RT_SHADE_MIRROR: shader.cl:37-39; RT_SHADE_DIFFUSE: a
// cosine-weighted direction, the normal plus a uniform point of the unit sphere, drawn by Marsaglia's
// rejection method from hashed counters so that it only depends on seed, sample and bounce number).
template <bool LEAN, bool MAD>
__device__ __forceinline__ uint32_t bounce_shader(RaySlot *ray, HotRay h, float4 color, uint32_t shading, uint32_t depth)
{
    typedef Arith<MAD> A;
    const V3 dir = h.dir, norm = A::normalize(xyz(ray->norm));
    V3 out;
    if ((shading & 15u) == RT_SHADE_MIRROR) {
        out = A::madd(norm, -(2 * A::dot(dir, norm)), dir);  // dir - norm * (2 dot(dir, norm))
    } else {
        const V3 up = A::dot(dir, norm) > 0 ? v3(-norm.x, -norm.y, -norm.z) : norm;
        const uint32_t base = hash32((shading >> 8) ^ h.sample_id * 0x9E3779B9u ^ depth * 0x85EBCA6Bu);
        float x1 = 0, x2 = 0, s = 0;
        for (uint32_t k = 0; k < 8; k++) {
            const float a = 2 * uniform01(hash32(base + 2 * k)) - 1, b = 2 * uniform01(hash32(base + 2 * k + 1)) - 1;
            const float q = a * a + b * b;
            if (q < 1) {
                x1 = a;  x2 = b;  s = q;
                break;
            }
        }
        const float r = 2 * sqrtf(1 - s);
        const V3 v = up + v3(x1 * r, x2 * r, 1 - 2 * s);
        const float len2 = A::dot(v, v);
        out = len2 > 1e-12f ? v * (1.0f / sqrtf(len2)) : up;
        float4 w = ray_weight(ray, h.type);
        w.x *= color.x;
        w.y *= color.y;
        w.z *= color.z;
        ray->weight = w;
    }
    if ((shading & 15u) == RT_SHADE_MIRROR && h.type == RT_RAY_PRIMARY) ray->weight = make_float4(1, 1, 1, 1);  // no longer implied
    h.type = RT_RAY_PRIMARY | (depth + 1) << 8;
    h.start = A::madd(dir, h.t_max, h.start);
    h.dir = out;
    const RayHit root = ray->root;
    return reset_ray<LEAN>(ray, h, root.group_id, root.local0, root.local1, RT_SKY_GROUP);
}

// mat_shader's direct-light part (shader.cl:20-40) for a ray whose hot fields and stored normal are at hand.
template <bool LEAN, bool MAD>
__device__ __forceinline__ uint32_t direct_light_shader(RaySlot *ray, HotRay h, V3 stored_norm, float4 color)
{
    typedef Arith<MAD> A;
    const RayHit root = ray->root;  // (asked for before anything is stored)
    const float l = 0.57735026f;
    const V3 light = v3(l, -l, l);
    const float alpha = 100, f0 = 0.04f;
    const V3 dir = h.dir, norm = A::normalize(stored_norm);
    const V3 hvec = A::normalize(light - dir);
    float spec = (alpha + 2) / 8 * pow100(fmaxf(0.0f, A::dot(norm, hvec)));
    spec *= f0 + (1 - f0) * pow5(fmaxf(0.0f, -A::dot(dir, hvec)));
    const float lambert = fmaxf(0.0f, A::dot(light, norm));
    const float sw = spec * color.w;
    const V3 c = v3(color.x + sw, color.y + sw, color.z + sw) * lambert;

    float4 w = ray_weight(ray, h.type);
    w.x *= c.x;
    w.y *= c.y;
    w.z *= c.z;
    w.w *= 1.0f;
    ray->weight = w;
    h.type = RT_RAY_SHADOW;
    h.start = A::madd(dir, h.t_max, h.start);
    h.dir = light;
    return reset_ray<LEAN>(ray, h, root.group_id, root.local0, root.local1, RT_LIGHT_GROUP);
}

template <bool LEAN, bool MAD>
__device__ __forceinline__ uint32_t mat_shader(RaySlot *ray, float4 color, uint32_t shading)
{
    HotRay h = load_hot(ray);
    if (shading & 15u) {
        const uint32_t depth = h.type >> 8;
        if (depth < ((shading >> 4) & 15u)) return bounce_shader<LEAN, MAD>(ray, h, color, shading, depth);
    }
    return direct_light_shader<LEAN, MAD>(ray, h, xyz(ray->norm), color);
}

// Free schedule: a ray whose queue has emptied goes to its end group (sky, light, material or spawn).  The
// reference spends a step of its own on that group's shader; here the visiting kernel runs it on the
// spot.  Spawn and the dead-ray bin (shader bits 0) stay keys: a finished slot is re-initialised by the
// box kernel at the next step, when its position in the sorted index gives it its sample id.
template <bool MAD>
__device__ __forceinline__ uint32_t run_end_group(uint32_t key, RaySlot *ray, const ProcessArgs &a)
{
    const uint32_t shader = (key >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK;
    if (shader == RT_SH_SPAWN || shader >= RT_SH_AABB) return key;
    asm volatile("" ::: "memory");  // the shader reads what the visit has just stored to the slot
    if (step_constant(&a.frame.glob->mode) & kModeTraceOnly) return a.group_count - 1;
    if (shader != RT_SH_MATERIAL) return sky_or_light_shader<true>(a, ray, shader == RT_SH_LIGHT);
    const float4 color = *reinterpret_cast<const float4 *>(a.scene.grp + (key & kIdMask));
    count_lanes(a.frame, kShadowSpawned, true);
    return mat_shader<true, MAD>(ray, color, step_constant(&a.frame.glob->shading));
}

// Asks for the 64-byte lines of a wave-uniform list ahead of the scan that reads it through the scalar path one
// batch after the other: the wave's active lanes touch up to four lines each, all requests in flight together, so
// that by the time the scan gets to a line it is in the L2 cache.  touch_list() issues the loads; touch_sum() adds the
// touched words up (one register instead of four across the scan), and the sum has to be handed to touch_done() after
// the scan: that is what keeps the loads.
struct Touched {
    uint32_t v[4];
};
__device__ __forceinline__ Touched touch_list(const void *list, uint32_t bytes)
{
    const unsigned long long active = __ballot(true);
    const uint32_t workers = (uint32_t)__popcll(active), rank = (uint32_t)__popcll(active & ((1ull << lane_id()) - 1));
    const char *base = static_cast<const char *>(list);
    // No branch around a load (each would get a wait of its own): a lane with nothing left to ask for asks for the
    // list's last line again.
    const uint32_t last = bytes ? (bytes - 1) & ~63u : 0u;  // (an empty list is followed by readable entries)
    Touched t;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t at = (rank + k * workers) * 64;
        t.v[k] = *reinterpret_cast<const uint32_t *>(base + (at < last ? at : last));
    }
    return t;
}
__device__ __forceinline__ uint32_t touch_sum(const Touched &t)
{
    uint32_t sum = (t.v[0] + t.v[1]) + (t.v[2] + t.v[3]);
    asm volatile("" : "+v"(sum));
    return sum;
}
__device__ __forceinline__ void touch_done(uint32_t sum) { asm volatile("" ::"v"(sum)); }

// ---------------------------------------------------------------- leaf visit (shader.cl:115-139, ray-tracer.cl:149-203)

// LEAN (free schedule): `orig` is not kept up, and a ray whose queue has emptied is not given its end group
// as head -- the caller runs that group's shader right away.
template <bool LEAN, bool MAD>
__device__ __forceinline__ uint32_t visit_mesh(uint32_t g, RaySlot *ray, const ProcessArgs &a)
{
    typedef Arith<MAD> A;
    MISC_STAMP(t_start);
    const u32x4 grp = *const_ptr<u32x4>(a.scene.grp + (g & kIdMask));  // vtx_offs, tri_offs, tri_count, material_id
    // What the scan needs from the slot (its first 64 bytes: ray and head of the queue) is requested at once.
    const HotRay h = load_hot(ray);
    const uint32_t queue_len = h.queue_len;
    const RayHit head = ray->queue[0];
    const LocalRay lr = make_local<MAD>(g, h, head.local0, a.scene.mat);
    // (asked for before the instance matrix the leaf kernel is 1 % faster alone, but the four registers the requests
    // hold across the transformation cost 12 bytes of scratch and the step comes out slower)
    const uint32_t touched = touch_sum(touch_list(a.scene.tri_rec + grp.y, (grp.z + 1) / 2 * (2 * (uint32_t)sizeof(TriRecord))));

    count_lanes(a.frame, kTriTests, true, grp.z);
    // Triangle scan.  Leaf payload is wave-uniform and comes through the scalar cache, two triangles at a time:
    // a pair's record is 96 bytes, the twelve values of the two triangles (r0, p, q, n) interleaved, so that each
    // value of the first triangle shares an aligned register pair with the same value of the second and the
    // subtractions, products and sums of the test run for both triangles at once (packed FP32).  The device copy
    // of a leaf's list starts on an even index, is filled up to an even count with null triangles and is followed
    // by readable entries, so the next pair is always requested before the current pair is tested.
    ConstPtr<f32x8> rec = const_ptr<f32x8>(a.scene.tri_rec + grp.y);
#if defined(RT_STAMPS) && RT_STAMPS == 2
    asm volatile("" ::"v"(lr.start.x), "v"(lr.dir.x), "v"(lr.tmin));
#endif
    MISC_STAMP(t_loaded);
    float best = lr.tmax, hit_u = 0, hit_v = 0;
    uint32_t hit_index = 0xFFFFFFFFu;
    const uint32_t count = grp.z;
    const f32x2 sx = splat(lr.start.x), sy = splat(lr.start.y), sz = splat(lr.start.z);
    const f32x2 dx = splat(lr.dir.x), dy = splat(lr.dir.y), dz = splat(lr.dir.z);
    // mesh_shader's test (shader.cl:126-136) for triangles i and i + 1: every operation as there, in the same
    // order; the first triangle's hit is committed before the second is looked at.
    auto test_two = [&](const f32x8 x, const f32x8 y, const f32x8 z, uint32_t i) {
        const f32x2 r0x = {x.s0, x.s1}, r0y = {x.s2, x.s3}, r0z = {x.s4, x.s5};
        const f32x2 nx = {z.s2, z.s3}, ny = {z.s4, z.s5}, nz = {z.s6, z.s7};
        const f32x2 rx = r0x - sx, ry = r0y - sy, rz = r0z - sz;
        const f32x2 w = reciprocal2(A::dot2(dx, dy, dz, nx, ny, nz));
        const f32x2 tt = A::dot2(rx, ry, rz, nx, ny, nz) * w;
        const bool near0 = tt.x > lr.tmin && tt.x < best, near1 = tt.y > lr.tmin && tt.y < best;
        if (!(near0 || near1)) return;
        const f32x2 px = {x.s6, x.s7}, py = {y.s0, y.s1}, pz = {y.s2, y.s3};
        const f32x2 qx = {y.s4, y.s5}, qy = {y.s6, y.s7}, qz = {z.s0, z.s1};
        const f32x2 cx = A::cross2(dy, rz, dz, ry), cy = A::cross2(dz, rx, dx, rz), cz = A::cross2(dx, ry, dy, rx);  // cross(dir, r)
        const f32x2 u = -A::dot2(qx, qy, qz, cx, cy, cz) * w, v = A::dot2(px, py, pz, cx, cy, cz) * w;
        const f32x2 uv = u + v;
        if (near0 && u.x >= 0 && v.x >= 0 && uv.x <= 1) {
            best = tt.x;
            hit_u = u.x;
            hit_v = v.x;
            hit_index = i;
        }
        if (near1 && tt.y < best && u.y >= 0 && v.y >= 0 && uv.y <= 1) {
            best = tt.y;
            hit_u = u.y;
            hit_v = v.y;
            hit_index = i + 1;
        }
    };
    if (count) {
        f32x8 a0 = rec[0], a1 = rec[1], a2 = rec[2];
        for (uint32_t i = 0;; i += 4) {
            const uint32_t at = 3 * (i / 2);
            scalar_loads_landed();  // a0..a2 are in; only then ask for the next pair
            const f32x8 b0 = rec[at + 3], b1 = rec[at + 4], b2 = rec[at + 5];
            test_two(a0, a1, a2, i);
            if (i + 2 >= count) break;
            scalar_loads_landed();
            a0 = rec[at + 6];
            a1 = rec[at + 7];
            a2 = rec[at + 8];
            test_two(b0, b1, b2, i + 2);
            if (i + 4 >= count) break;
        }
    }

    touch_done(touched);
    MISC_STAMP(t_scanned);
    MISC_STAMP_ADD(a.frame, 0, t_start, t_loaded);
    MISC_STAMP_ADD(a.frame, 1, t_loaded, t_scanned);
    // Miss: pop the head (ray-tracer.cl:154-165).  Hit: insert_stop (ray-tracer.cl:173-203).  What a
    // hit still needs from memory (shading normal, sample id, or the shadow weight) is loaded before
    // the first store of this visit.
    const bool hit = hit_index != 0xFFFFFFFFu, shadow = (h.type & 0xFFu) == RT_RAY_SHADOW;  // (bounce count above the low byte)
    uint32_t end_group = h.material_id, sid = 0xFFFFFFFFu;
    float shadow_weight = 0;
    V3 nw = v3(0, 0, 0);
    if (hit) {
        if (shadow) {
            shadow_weight = ray->weight.w;
            end_group = RT_SPAWN_GROUP;
        } else {
            const uint32_t word = a.scene.tri[grp.y + hit_index];
            sid = h.sample_id;
            const rt_vertex *vtx = a.scene.vtx + grp.x;
            const float4 n0 = *reinterpret_cast<const float4 *>(vtx[word & 0x3FF].norm);
            const float4 n1 = *reinterpret_cast<const float4 *>(vtx[(word >> 10) & 0x3FF].norm);
            const float4 n2 = *reinterpret_cast<const float4 *>(vtx[(word >> 20) & 0x3FF].norm);
            const V3 nn = A::lin3(xyz(n0), 1 - hit_u - hit_v, xyz(n1), hit_u, xyz(n2), hit_v);
            nw = A::lin3(lr.m0, nn.x, lr.m1, nn.y, lr.m2, nn.z);
            end_group = grp.w;
        }
    }
    const float new_max = hit && !shadow ? best : h.t_max;
    auto record_hit = [&]() {  // the record is the camera ray's hit: a sample that has bounced keeps it
        if (h.type == RT_RAY_PRIMARY && sid < step_constant(&a.frame.glob->hit_capacity)) a.frame.hits[sid] = rt_hit_record{best, head.group_id, head.local0, hit_index};
    };
    auto commit_hit = [&](uint32_t len, float t_min) {  // what a hit leaves in the slot beside the queue
        if (shadow) {
            unsafeAtomicAdd(reinterpret_cast<float *>(&a.frame.area[h.pixel]) + 3, shadow_weight);
        } else {
            ray->norm = make_float4(nw.x, nw.y, nw.z, 0.0f);
            if (!LEAN) ray->orig = RayHit{best, head.group_id, head.local0, head.local1};
            record_hit();
        }
        store_commit(ray, end_group, h.type, len, t_min, new_max);
    };

    if (LEAN) {
        // Stack of pending entries.  The nearest one becomes the head unless it starts at or beyond the ray's
        // reach: then so do all the others, and the queue is empty.  A hit shortens the reach and leaves the stack
        // alone (the reference drops the entries beyond the hit on the spot).  Not so for a shadow ray: its hits do
        // not shorten its reach -- boxes beyond them are still entered afterwards -- so its entries beyond the hit
        // are taken out now, and the nearer ones move down to the bottom of the stack.
        uint32_t len = queue_len - 1;  // entries after this visit, the new head included
        RayHit next = RayHit{0.0f, 0u, 0u, 0u};
        bool moved = false;
        if (hit && shadow && len > 1) {
            uint32_t keep = len;
            RayHit e[RT_MAX_QUEUE_LEN];
#pragma unroll
            for (int i = 1; i < RT_MAX_QUEUE_LEN; i++) {
                e[i] = RayHit{0.0f, 0u, 0u, 0u};
                if ((uint32_t)i <= len) e[i] = ray->queue[queue_len - i];
            }
#pragma unroll
            for (int i = 1; i < RT_MAX_QUEUE_LEN; i++)
                if ((uint32_t)i <= keep && e[i].pos >= best) keep = i - 1;
            if (keep != len) {
#pragma unroll
                for (int i = 2; i < RT_MAX_QUEUE_LEN; i++)
                    if ((uint32_t)i <= keep) ray->queue[keep + 1 - i] = e[i];
                len = keep;
                next = e[1];
                moved = true;
            }
        }
        if (len && !moved) next = ray->queue[len];
        if (len && next.pos >= (hit && shadow ? best : new_max)) len = 0;
        // All that the stores below need from memory has been requested; waited for here, once, they go out back to
        // back (otherwise the compiler waits in front of a store's operands for everything under way, stores included).
        vector_loads_landed();
        if (len) {
            if (hit) commit_hit(len, head.pos);
            else store_progress(ray, h.type, len, head.pos);
            ray->queue[0] = next;
            return next.group_id;
        }
        if (hit) commit_hit(queue_len, h.t_min);
        else if (finish_in_sky_or_light(a, ray, end_group, h.pixel, h.type)) return RT_SPAWN_GROUP;
        return end_group;  // (the caller runs that group's shader right away: the head is not touched)
    }

    // The pending entries are fetched only now: held across the scan they would cost 28 registers.
    RayHit q[RT_MAX_QUEUE_LEN];
    q[0] = head;
#pragma unroll
    for (int i = 1; i < RT_MAX_QUEUE_LEN; i++) {
        q[i] = RayHit{0.0f, 0u, 0u, 0u};
        if ((uint32_t)i < queue_len) q[i] = ray->queue[i];
    }
    uint32_t len = queue_len - 1;
    if (hit) {
        // pending nodes that start at or beyond the hit are dropped, and everything after them
#pragma unroll
        for (int i = 0; i < RT_MAX_QUEUE_LEN - 1; i++)
            if ((uint32_t)i < len && q[i + 1].pos >= best) len = i;
    }
    // save_queue + copy_queue (ray-tracer.cl:157-168, 195-203).  An emptied queue leaves queue_len and t_min
    // alone; with a hit, the end group and the new t_max go out in the same request as those two.
    if (hit) commit_hit(len ? len : queue_len, len ? head.pos : h.t_min);
    else if (len) store_progress(ray, h.type, len, head.pos);
#pragma unroll
    for (int i = 0; i < RT_MAX_QUEUE_LEN - 1; i++)
        if ((uint32_t)i < len) ray->queue[i] = q[i + 1];
    if (len) return q[1].group_id;
    // Empty: the head becomes the ray's end group; queue[0].pos keeps its value (the reference leaves
    // it indeterminate, ray-tracer.cl:164,201).
    ray->queue[0].group_id = end_group;
    ray->queue[0].local0 = 0;
    ray->queue[0].local1 = 0;
    return end_group;
}

// ---------------------------------------------------------------- process, box groups
// aabb_shader (shader.cl:76-101) + insert_hits (ray-tracer.cl:90-117, 205-231)
//
// More than 16 accepted boxes in a visit are rare (3 visits in 10,000 on the benchmark scene, 5 in a
// million more than 24, the maximum seen is 36), so the kernel keeps 16 candidates per lane in LDS
// (20 KB per block of 256, plus 2 KB for the heads' local ids) and holds the pending queue entries in registers only after the scan: seven
// waves per SIMD instead of the two that 64 candidates per lane would allow.  A lane that accepts a
// 17th box leaves its ray untouched; k_process_aabb_big redoes that visit with a whole wave.

#ifndef RT_AABB_THREADS
#define RT_AABB_THREADS 256
#endif
#ifndef RT_MESH_THREADS
#define RT_MESH_THREADS 256
#endif
constexpr int kAabbThreads = RT_AABB_THREADS;  // main kernel
constexpr int kMeshThreads = RT_MESH_THREADS;
#ifndef RT_AABB_CAP
#define RT_AABB_CAP 16
#endif
#ifndef RT_AABB_WAVES
#define RT_AABB_WAVES 7  // waves per SIMD the box kernel is compiled for (72 registers)
#endif
#ifndef RT_MESH_WAVES
#define RT_MESH_WAVES 7  // the same for the leaf kernel
#endif
constexpr int kAabbCap = RT_AABB_CAP;

// What the slab test needs of a ray (shader.cl:79-91): the reciprocal direction, and the start -- or, when the
// multiply-adds are fused, -(start * reciprocal direction), which turns (corner - start) * inv_dir into one fused
// operation per corner coordinate.  x and y of each share a register pair; the two z values share a third
// (the packed operation on the corners' z reads one half of it for both corners).
template <bool MAD>
struct Slab {
    f32x2 a_xy, i_xy, z;  // z = {1 / dir.z, start.z or -(start.z / dir.z)}
    float ray_min, ray_max;
};
template <bool MAD>
__device__ __forceinline__ Slab<MAD> make_slab(const V3 start, const V3 dir, float ray_min, float ray_max)
{
    const V3 inv = v3(1 / dir.x, 1 / dir.y, 1 / dir.z);
    const V3 a = MAD ? v3(-(start.x * inv.x), -(start.y * inv.y), -(start.z * inv.z)) : start;
    return Slab<MAD>{f32x2{a.x, a.y}, f32x2{inv.x, inv.y}, f32x2{inv.z, a.z}, ray_min, ray_max};
}

// One box (a DevBox: lo.xy, hi.xy, lo.z, hi.z, two ids) against one ray: entry and exit parameter of the slab test
// (shader.cl:86-91), three packed operations for the six corner coordinates.  UNIFORM: the box is the same for
// every lane and sits in scalar registers.  The reference accepts a box iff
// t_max > t_min && t_max > ray.min && t_min < ray.max; with t_entry = max(t_min, ray.min) the first two are
// t_max > t_entry for every ray whose ray.min is a number (max() then returns one of its operands, and a t_min
// that is not a number fails the third comparison).
template <bool MAD, bool UNIFORM = true>
__device__ __forceinline__ bool box_test(const f32x8 b, const Slab<MAD> &r, float &t_entry)
{
    const f32x2 lo_xy = {b.s0, b.s1}, hi_xy = {b.s2, b.s3}, z = {b.s4, b.s5};
    f32x2 p1, p2, pz;
    // The z of both corners against {1 / dir.z, a.z}: both halves of the operation read the first half of that pair
    // as factor and the second as the term (op_sel), where the compiler would hold each value twice.
    if (MAD) {
        p1 = __builtin_elementwise_fma(lo_xy, r.i_xy, r.a_xy);
        p2 = __builtin_elementwise_fma(hi_xy, r.i_xy, r.a_xy);
        if (UNIFORM) asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(pz) : "s"(z), "v"(r.z));
        else asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(pz) : "v"(z), "v"(r.z));
    } else {
        p1 = (lo_xy - r.a_xy) * r.i_xy;
        p2 = (hi_xy - r.a_xy) * r.i_xy;
        // (z - start.z) * inv.z: the subtraction as an addition of the negated second half
        if (UNIFORM)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_mul_f32 %0, %0, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(pz) : "s"(z), "v"(r.z));
        else
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_mul_f32 %0, %0, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(pz) : "v"(z), "v"(r.z));
    }
    const float t_min = fmaxf(fmaxf(fminf(p1.x, p2.x), fminf(p1.y, p2.y)), fminf(pz.x, pz.y));
    const float t_max = fminf(fminf(fmaxf(p1.x, p2.x), fmaxf(p1.y, p2.y)), fmaxf(pz.x, pz.y));
    t_entry = fmaxf(t_min, r.ray_min);
    return t_max > t_entry && t_min < r.ray_max;
}

// The two ids of box `bi` of a list, as two 4-byte loads: taken as one 8-byte load they would each need an aligned
// register pair, and the box kernel, which has no register to spare, would spill eleven more values.
__device__ __forceinline__ void box_ids(const DevBox *list, uint32_t bi, uint32_t &group_id, uint32_t &local_id)
{
    group_id = list[bi].group_id;
    uint32_t again = bi;
    asm("" : "+v"(again));  // (keeps the compiler from seeing that the two addresses are neighbours)
    local_id = list[again].local_id;
}

__device__ __forceinline__ float lane_value(float v, uint32_t lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)lane));
}

// The rare visit that accepts more than the per-lane columns hold (the reference allows 64 candidates,
// shader.cl:93): the whole wave works on ONE ray.  Every active lane tests a share of the group's
// boxes, accepted boxes are appended in list order (ballot + prefix count, as the reference's loop
// appends them), and the reference's compare-exchange network (ray-tracer.cl:90-117) runs with one
// lane per exchange: all exchanges of one inner loop of the network touch disjoint pairs.  c_pos/c_idx
// are this wave's 64-entry scratch lists in LDS.  Returns the candidate count (0 when more than 64
// boxes are accepted: the reference gives up on the whole node), the list is left sorted.
template <bool MAD>
__device__ __forceinline__ uint32_t wave_visit_one(const DevBox *list, uint32_t count, const Slab<MAD> &slab, float *c_pos, uint16_t *c_idx)
{
    const unsigned long long active = __ballot(true);
    const unsigned long long below = (1ull << lane_id()) - 1;
    const uint32_t rank = __popcll(active & below), workers = __popcll(active);
    uint32_t total = 0;
    for (uint32_t base = 0; base < count; base += workers) {
        const uint32_t i = base + rank;
        bool hit = false;
        float t = 0;
        if (i < count) {
            const float4 lo = *reinterpret_cast<const float4 *>(&list[i].lo_x), hi = *reinterpret_cast<const float4 *>(&list[i].lo_z);
            f32x8 b;
            b.s0 = lo.x;  b.s1 = lo.y;  b.s2 = lo.z;  b.s3 = lo.w;
            b.s4 = hi.x;  b.s5 = hi.y;  b.s6 = hi.z;  b.s7 = hi.w;
            hit = box_test<MAD, false>(b, slab, t);
        }
        const unsigned long long m = __ballot(hit);
        const uint32_t at = total + __popcll(m & below);
        if (hit && at < RT_MAX_HITS) {
            c_pos[at] = t;
            c_idx[at] = (uint16_t)i;
        }
        total += __popcll(m);
    }
    if (total > RT_MAX_HITS) return 0;
    auto exchange = [&](uint32_t j, uint32_t i) {
        const float pj = c_pos[j], pi = c_pos[i];
        if (pj > pi) {
            const uint16_t ij = c_idx[j], ii = c_idx[i];
            c_pos[j] = pi;  c_pos[i] = pj;
            c_idx[j] = ii;  c_idx[i] = ij;
        }
    };
    for (uint32_t base = 1; base < total; base *= 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = rank; i < total; i += workers)
            if (i >= base && (i & base)) exchange(i - 2 * (i & (base - 1)) - 1, i);
        for (uint32_t offs = base / 2; offs; offs /= 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t i = rank; i < total; i += workers)
                if (i >= offs && (i & offs)) exchange(i - offs, i);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return total;
}

// One visit of box group g (wave-uniform) by this lane's ray.  CAP = candidates kept per lane in the
// LDS columns s_pos/s_idx.  With COOP false, a lane that accepts more than CAP boxes returns false
// without having changed its ray.  With COOP true the whole wave performs the visit of ONE ray (the
// caller hands every lane that ray's group and slot): the scan and the candidate sort are done by
// wave_visit_one in the wave's lists c_pos/c_idx with the reference's limit of 64 candidates, and
// only the lane `owner` merges and stores.
template <int CAP, int THREADS, bool COOP, bool LEAN, bool MAD, bool FRESH = false, typename PLACE = uint8_t>
__device__ __forceinline__ bool visit_aabb(uint32_t g, RaySlot *ray, const ProcessArgs &a, float (*s_pos)[THREADS],
                                           PLACE (*s_idx)[THREADS], float *c_pos, uint16_t *c_idx, uint32_t owner,
                                           uint32_t tid, uint32_t &new_key, uint32_t &group_flags, uint32_t sample = 0,
                                           uint32_t head_lds = 0xFFFFFFFFu)
{
    AABB_STAMP(t_start);
    MERGE_STAMP(t_m_start);
    const u32x4 grp = *const_ptr<u32x4>(a.scene.grp + (g & kIdMask));  // aabb_offs, aabb_count, flags
    const uint32_t count = grp.y;
    group_flags = grp.z;
    // What the scan needs from the slot (ray and head of the queue: its first 64 bytes) is requested
    // now.  No load is issued once the stores of the merge begin.
    // FRESH: the visit that starts sample number `sample`.  Its slot has not been written: the ray is worked
    // out here, there are no pending entries, and the slot's hot fields are stored whole at the end.
    HotRay h = FRESH ? camera_ray<MAD>(a.cam, a.frame.glob, sample) : load_hot(ray);
    const RayHit head = FRESH ? RayHit{0.001f, a.cam.root_group, a.cam.root_local, 0u} : ray->queue[0];
    const uint32_t old_len = FRESH ? 1u : h.queue_len;
    const float reach = h.t_max;
    // The head's local ids go into every entry this visit adds.  Held in registers across the scan they get spilled,
    // and a reload from scratch in front of each of those stores waits for the store before it (one counter for
    // both).  So they wait in LDS (head_lds: the byte address of this lane's eight bytes there) and are read back
    // entry by entry -- in assembly, which keeps the compiler from holding on to what was read.
    const bool head_in_lds = !COOP && !FRESH && head_lds != 0xFFFFFFFFu;
    if (head_in_lds) {
        const unsigned long long both = (unsigned long long)head.local1 << 32 | head.local0;
        asm volatile("ds_write_b64 %0, %1" ::"v"(head_lds), "v"(both) : "memory");
    }
    auto put_fresh_sample_in_its_slot = [&]() {  // for the kernel that redoes a visit this one gives up
        store_hot(ray, h);
        ray->queue[0] = head;
    };
    const LocalRay lr = make_local<MAD>(g, h, head.local0, a.scene.mat);
    const Slab<MAD> slab = make_slab<MAD>(lr.start, lr.dir, lr.tmin, lr.tmax);
    if (!COOP) count_lanes(a.frame, kBoxTests, true, count);  // a redone visit was counted the first time

    uint32_t n = 0;
    if (COOP) {
        n = wave_visit_one<MAD>(a.scene.aabb + grp.x, count, slab, c_pos, c_idx);
        if (lane_id() != owner) return true;
    } else {
    // Box scan.  The list is wave-uniform and comes through the scalar cache.  First the bounds of the
    // list's clusters (up to eight boxes each) are tested: a cluster no lane of the wave enters is
    // skipped, which cannot change any lane's result (see build_clusters in rt_api.hip).  Then the
    // boxes of the entered clusters are tested in list order, so every lane collects its candidates
    // in the order the reference's loop does (shader.cl:84-99).  The device copies of both lists are
    // filled up with entries no ray can enter and followed by readable spare entries: the loops have
    // no tail handling and always have the next 128 bytes in flight while the current ones are tested.
    // A list of more than 256 boxes (kClusterLong) is scanned 256 boxes -- 32 clusters of eight consecutive
    // boxes -- at a time; any other list in one go.  Places in such a list take two bytes: a kernel whose
    // columns hold one leaves the visit to the one-ray-per-wave kernel.
    if (sizeof(PLACE) == 1 && count > kMaxClusteredBoxes) {
        if (FRESH) put_fresh_sample_in_its_slot();
        return false;
    }
    asm volatile("" ::"v"(slab.i_xy.x), "v"(slab.i_xy.y), "v"(slab.z.x), "v"(slab.a_xy.x));
    AABB_STAMP(t_loaded);
    bool spill = false;
    const bool long_list = (grp.z & kClusterLong) != 0;
#ifdef RT_COUNT_SCANNED
    uint32_t scanned = 0;
#endif
    auto test_box = [&](const f32x8 b, uint32_t i) {
        float t;
        if (box_test<MAD>(b, slab, t)) {
            if (n < CAP) {
                s_pos[n][tid] = t;
                s_idx[n][tid] = (PLACE)i;
                n++;
            } else {
                spill = true;
            }
        }
    };
    for (uint32_t seg = 0;;) {  // seg: first box of the part being scanned
    uint32_t entered = 0;
    uint32_t clusters = (grp.z >> kClusterCountShift) & kClusterCountMask;
    if (long_list) clusters = count - seg > kMaxClusteredBoxes ? kMaxClusteredBoxes / 8 : (count - seg + 7) / 8;
    if (clusters) {
        ConstPtr<f32x16> two = const_ptr<f32x16>(a.scene.cluster + grp.w + seg / 8);  // bounds of two clusters per load
        auto test_two = [&](const f32x16 p, uint32_t c) {
            float t;
            if (__ballot(box_test<MAD>(p.lo, slab, t))) entered |= 1u << c;
            if (__ballot(box_test<MAD>(p.hi, slab, t))) entered |= 2u << c;
        };
        // four clusters per round trip (the bounds are filled up to a multiple of four with clusters no ray enters)
        f32x16 c0 = two[0], c1 = two[1];
        for (uint32_t c = 0;; c += 8) {
            scalar_loads_landed();
            const f32x16 d0 = two[c / 2 + 2], d1 = two[c / 2 + 3];
            test_two(c0, c);
            test_two(c1, c + 2);
            if (c + 4 >= clusters) break;
            scalar_loads_landed();
            c0 = two[c / 2 + 4];
            c1 = two[c / 2 + 5];
            test_two(d0, c + 4);
            test_two(d1, c + 6);
            if (c + 8 >= clusters) break;
        }
    }
#ifdef RT_COUNT_SCANNED
    scanned += clusters;
#endif
    if (!(grp.z & kClusterMasked)) {
        // cluster c = boxes [8c, 8c + 8): two loads of 128 bytes per half cluster, ping-pong
        ConstPtr<f32x16> pair = const_ptr<f32x16>(a.scene.aabb + grp.x + seg);  // per box: min.xyz, group id, max.xyz, local id
        if (entered) {
            uint32_t c = (uint32_t)__builtin_ctz(entered);
            entered &= entered - 1;
            f32x16 a0 = pair[4 * c], a1 = pair[4 * c + 1];
            for (;;) {
                scalar_loads_landed();  // a0, a1 are in; only then ask for the next four boxes
                const f32x16 b0 = pair[4 * c + 2], b1 = pair[4 * c + 3];
                test_box(a0.lo, seg + 8 * c);
                test_box(a0.hi, seg + 8 * c + 1);
                test_box(a1.lo, seg + 8 * c + 2);
                test_box(a1.hi, seg + 8 * c + 3);
                const bool more = entered != 0;
                const uint32_t next = more ? (uint32_t)__builtin_ctz(entered) : c;  // (a harmless reload at the end)
                entered &= entered - 1;
                scalar_loads_landed();
                a0 = pair[4 * next];
                a1 = pair[4 * next + 1];
                test_box(b0.lo, seg + 8 * c + 4);
                test_box(b0.hi, seg + 8 * c + 5);
                test_box(b1.lo, seg + 8 * c + 6);
                test_box(b1.hi, seg + 8 * c + 7);
#ifdef RT_COUNT_SCANNED
                scanned += 8;
#endif
                if (!more) break;
                c = next;
            }
        }
    } else {
        // clusters with membership masks: the boxes of the entered clusters, in list order
        ConstPtr<u64x4> masks = const_ptr<u64x4>(a.scene.cluster + grp.w + ((clusters + 3) & ~3u));
        unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        while (entered) {
            const uint32_t c = (uint32_t)__builtin_ctz(entered);
            entered &= entered - 1;
            const uint32_t c2 = entered ? (uint32_t)__builtin_ctz(entered) : c;
            entered &= entered - 1;
            const u64x4 x = masks[c], y = masks[c2];
            scalar_loads_landed();
            m0 |= x.x | y.x;
            m1 |= x.y | y.y;
            m2 |= x.z | y.z;
            m3 |= x.w | y.w;
        }
        ConstPtr<f32x8> box = const_ptr<f32x8>(a.scene.aabb + grp.x);
        auto scan_word = [&](unsigned long long m, uint32_t first) {
            while (m) {
                // up to four boxes per round: their loads are in flight together
                uint32_t i0 = first + (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const bool h1 = m != 0;
                const uint32_t i1 = h1 ? first + (uint32_t)__builtin_ctzll(m) : i0;
                m &= m - 1;
                const bool h2 = m != 0;
                const uint32_t i2 = h2 ? first + (uint32_t)__builtin_ctzll(m) : i0;
                m &= m - 1;
                const bool h3 = m != 0;
                const uint32_t i3 = h3 ? first + (uint32_t)__builtin_ctzll(m) : i0;
                m &= m - 1;
                const f32x8 b0 = box[i0], b1 = box[i1], b2 = box[i2], b3 = box[i3];
                scalar_loads_landed();
                test_box(b0, i0);
                if (h1) test_box(b1, i1);
                if (h2) test_box(b2, i2);
                if (h3) test_box(b3, i3);
#ifdef RT_COUNT_SCANNED
                scanned += 1 + h1 + h2 + h3;
#endif
            }
        };
        scan_word(m0, 0);
        scan_word(m1, 64);
        scan_word(m2, 128);
        scan_word(m3, 192);
    }
    seg += kMaxClusteredBoxes;
    if (!long_list || seg >= count) break;
    }
#ifdef RT_COUNT_SCANNED
    count_lanes(a.frame, kBoxScanned, true, scanned);
#endif
    AABB_STAMP(t_scanned);
    AABB_STAMP_ADD(a.frame, 0, t_start, t_loaded);
    AABB_STAMP_ADD(a.frame, 1, t_loaded, t_scanned);
    if (spill) {
        if (FRESH) put_fresh_sample_in_its_slot();
        return false;
    }
    }
    // The pending entries are fetched only now (two visits in three have none): held across the scan
    // they would cost 28 registers, i.e. two of the seven waves a SIMD holds of this kernel.  The hit
    // network below covers most of their latency.
    RayHit q[RT_MAX_QUEUE_LEN];
    q[0] = head;
#pragma unroll
    for (int i = 1; i < RT_MAX_QUEUE_LEN; i++) {
        q[i] = RayHit{0.0f, 0u, 0u, 0u};
        if ((uint32_t)i < old_len) q[i] = ray->queue[LEAN ? old_len - i : i];
    }

    // sort_hits (ray-tracer.cl:90-117): the reference's compare-exchange network, run with
    // wave-uniform (i, j) so every LDS access has a uniform row; lanes with fewer entries idle.
    AABB_STAMP(t_queue);
    if (!COOP) {
    const uint32_t n_max = wave_max_upto127(n);
    auto exchange = [&](uint32_t j, uint32_t i) {
        if (i < n) {
            const float pj = s_pos[j][tid], pi = s_pos[i][tid];
            if (pj > pi) {
                const PLACE ij = s_idx[j][tid], ii = s_idx[i][tid];
                s_pos[j][tid] = pi;  s_pos[i][tid] = pj;
                s_idx[j][tid] = ii;  s_idx[i][tid] = ij;
            }
        }
    };
    for (uint32_t base = 1; base < n_max; base *= 2) {
        for (uint32_t i = base; i < n_max; i++)
            if (i & base) exchange(i - 2 * (i & (base - 1)) - 1, i);
        for (uint32_t offs = base / 2; offs; offs /= 2)
            for (uint32_t i = offs; i < n_max; i++)
                if (i & offs) exchange(i - offs, i);
    }
    }

    AABB_STAMP(t_sorted);
    AABB_STAMP_ADD(a.frame, 2, t_queue, t_sorted);
    // insert_hits (ray-tracer.cl:205-231): merge the sorted candidates with the pending entries, old
    // entry first on equal position, at most 8 entries, the 8th turned into a root restart.  Both
    // lists are sorted, so every entry's place in the merged queue is a count: a pending entry i goes
    // behind the earlier pending entries and the candidates strictly before it, a candidate j behind
    // the earlier candidates and the pending entries not after it.  Only the first 8 candidates can
    // land inside the queue.  Their child ids are fetched together, then every surviving entry is
    // stored to its place from registers -- independent stores, no load in between.
    const bool take0 = (grp.z & RT_F_LOCAL0) != 0, take1 = (grp.z & RT_F_LOCAL1) != 0;
    const DevBox *list = a.scene.aabb + grp.x;
    float cand_pos[RT_MAX_QUEUE_LEN];
    uint32_t cand_group[RT_MAX_QUEUE_LEN], cand_local[RT_MAX_QUEUE_LEN];
#pragma unroll
    for (int j = 0; j < RT_MAX_QUEUE_LEN; j++) {
        cand_pos[j] = INFINITY;
        cand_group[j] = cand_local[j] = 0;
        if ((uint32_t)j < n) {
            const uint32_t bi = COOP ? c_idx[j] : s_idx[j][tid];
            cand_pos[j] = COOP ? c_pos[j] : s_pos[j][tid];
            if (j < 4) {  // (the later candidates' ids are fetched below, and only by waves that have such candidates)
                box_ids(list, bi, cand_group[j], cand_local[j]);
            }
        }
    }
#if defined(RT_STAMPS) && RT_STAMPS == 4
    MERGE_STAMP(t_m0);
    asm volatile("" ::"v"(cand_group[0]), "v"(cand_group[1]), "v"(cand_group[2]), "v"(cand_group[3]), "v"(cand_local[0]), "v"(cand_local[3]), "v"(cand_group[7]), "v"(cand_local[7]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MERGE_STAMP(t_m1);
    MERGE_STAMP_ADD(a.frame, 0, t_m0, t_m1);
#endif
    // Free schedule: pending entries that start at or beyond the ray's reach -- a hit since they were queued has
    // made them unreachable, and with them all that follow (see RaySlot) -- are counted out first.
    uint32_t live_len = old_len;
    if (LEAN && !FRESH) {
#pragma unroll
        for (int i = 1; i < RT_MAX_QUEUE_LEN; i++)
            if ((uint32_t)i < live_len && q[i].pos >= reach) live_len = i;
    }
    const uint32_t merged = live_len - 1 + n;
    const bool overflow = merged >= RT_MAX_QUEUE_LEN;
    const uint32_t k = overflow ? RT_MAX_QUEUE_LEN : merged;
    RayHit root = RayHit{0.0f, 0u, 0u, 0u};
    if (overflow) root = ray->root;  // ray-tracer.cl:228-231: the 8th entry keeps its position, restarts from the root later
    uint32_t first_group = 0;
    // Stack: queue[at] of the merged queue lies in queue[k - at]; `was` is where the entry lay before (kNowhere: a new one).
    // A pending entry stays where it is unless a candidate lands behind it or entries beyond it were counted out:
    // such entries are not stored again.
    constexpr uint32_t kNowhere = 0xFFFFFFFFu;
    auto place = [&](uint32_t at, RayHit e, uint32_t was) {
        if (at >= RT_MAX_QUEUE_LEN) return;
        if (at == RT_MAX_QUEUE_LEN - 1) {
            e.group_id = root.group_id;
            e.local0 = root.local0;
            e.local1 = root.local1;
            was = kNowhere;
        }
        if (at == 0) first_group = e.group_id;
#ifndef RT_ABLATE_STORES
        if (LEAN) {
            const uint32_t to = at ? k - at : 0u;
            if (to != was) ray->queue[to] = e;
        } else {
            ray->queue[at] = e;
        }
#else
        asm volatile("" ::"v"(e.pos), "v"(e.group_id), "v"(e.local0), "v"(e.local1), "v"(at));
#endif
    };
    // The counts run over 7 x 8 pairs in full; most visits have few pending entries and few candidates, so the
    // loops are cut, wave-uniformly, at the largest numbers any lane of the wave has (in halves of four).
    const bool many_old = __ballot(live_len > 4) != 0, many_new = __ballot(n > 4) != 0;
    const bool some_old = __ballot(live_len > 1) != 0;
    auto candidates_before = [&](float pos) {  // candidates strictly before a pending entry
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) c += cand_pos[j] < pos ? 1u : 0u;
        if (many_new) {
#pragma unroll
            for (int j = 4; j < RT_MAX_QUEUE_LEN; j++) c += cand_pos[j] < pos ? 1u : 0u;
        }
        return c;
    };
    auto pending_not_after = [&](float pos) {  // pending entries a candidate goes behind
        uint32_t c = 0;
#pragma unroll
        for (int i = 1; i < 4; i++) c += ((uint32_t)i < live_len && !(pos < q[i].pos)) ? 1u : 0u;
        if (many_old) {
#pragma unroll
            for (int i = 4; i < RT_MAX_QUEUE_LEN; i++) c += ((uint32_t)i < live_len && !(pos < q[i].pos)) ? 1u : 0u;
        }
        return c;
    };
    auto place_pending = [&](int i) {
        if ((uint32_t)i < live_len) place(i - 1 + candidates_before(q[i].pos), q[i], old_len - i);
    };
    // Everything the placement needs from memory (pending entries, child ids, the root entry) has been requested;
    // one wait here, and the stores below go out back to back.  Without it the compiler, which cannot tell across the
    // branches which loads are still under way, waits for all vector memory operations before each store's operands,
    // i.e. also for the store before.
    // The later candidates' ids: where registers are not short (the kernel with two-byte places, the one-ray-per-wave
    // kernel) they are asked for now, by the waves that need them; the main box kernel fetches them after the first four
    // candidates have been placed (below).
    constexpr bool ids_first = sizeof(PLACE) > 1 || COOP;
    if (ids_first && many_new) {
#pragma unroll
        for (int j = 4; j < RT_MAX_QUEUE_LEN; j++) {
            if ((uint32_t)j < n) {
                const uint32_t bi = COOP ? c_idx[j] : s_idx[j][tid];
                box_ids(list, bi, cand_group[j], cand_local[j]);
            }
        }
    }
    vector_loads_landed();
    if (some_old) {
#pragma unroll
        for (int i = 1; i < 4; i++) place_pending(i);
        if (many_old) {
#pragma unroll
            for (int i = 4; i < RT_MAX_QUEUE_LEN; i++) place_pending(i);
        }
    }
    auto place_candidate = [&](int j) {
        if ((uint32_t)j < n) {
            uint32_t inherited0 = head.local0, inherited1 = head.local1;
            if (head_in_lds) {
                unsigned long long both;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(both) : "v"(head_lds) : "memory");
                inherited0 = (uint32_t)both;
                inherited1 = (uint32_t)(both >> 32);
            }
            place(j + (some_old ? pending_not_after(cand_pos[j]) : 0u),
                  RayHit{cand_pos[j], cand_group[j], take0 ? cand_local[j] : inherited0, take1 ? cand_local[j] : inherited1}, kNowhere);
        }
    };
#pragma unroll
    for (int j = 0; j < 4; j++) place_candidate(j);
    if (many_new) {
        // Rarely needed, and sixteen registers the common case does without (held from the start they are spilled, and
        // every reload from scratch in front of a store waits for the stores before it).
        if (!ids_first) {
#pragma unroll
            for (int j = 4; j < RT_MAX_QUEUE_LEN; j++) {
                if ((uint32_t)j < n) {
                    const uint32_t bi = COOP ? c_idx[j] : s_idx[j][tid];
                    cand_group[j] = list[bi].group_id;
                    cand_local[j] = list[bi].local_id;
                }
            }
            vector_loads_landed();
        }
#pragma unroll
        for (int j = 4; j < RT_MAX_QUEUE_LEN; j++) place_candidate(j);
    }
    // save_queue (ray-tracer.cl:157-165)
    if (FRESH) {
        if (k) {
            h.queue_len = k;
            h.t_min = head.pos;
        } else if (LEAN && h.material_id == RT_SKY_GROUP && !(step_constant(&a.frame.glob->mode) & kModeTraceOnly)) {
            // A camera ray that enters nothing: the sky's share goes to its pixel here and now (sky_shader,
            // shader.cl:6-11, with the weight of one an unbounced camera ray has).  The slot is not written at all
            // and nothing is read back from it: the free schedule keeps no state for a finished slot.
            area_add(a.frame.area, h.pixel, 1.0f * 0.5f, 1.0f * 1.0f, 1.0f * 1.0f, 1.0f * 1.0f);
            new_key = RT_SPAWN_GROUP;
            return true;
        }
        // (a sample that enters nothing keeps queue_len and t_min as reset_ray set them, as below)
        store_hot(ray, h);
    }
    if (k) {
        if (!FRESH) store_progress(ray, h.type, k, head.pos);
        new_key = first_group;
    } else {
        const uint32_t end_group = h.material_id;
        if (!LEAN) {
            ray->queue[0].group_id = end_group;
            ray->queue[0].local0 = 0;
            ray->queue[0].local1 = 0;
        }
        new_key = end_group;
        // (sky and light are dealt with here, where pixel and type are at hand: the caller's run_end_group would read them back)
        if (LEAN && finish_in_sky_or_light(a, ray, end_group, h.pixel, h.type)) new_key = RT_SPAWN_GROUP;
    }
    AABB_STAMP(t_merged);
    AABB_STAMP_ADD(a.frame, 3, t_sorted, t_merged);
#if defined(RT_STAMPS) && RT_STAMPS == 4
    MERGE_STAMP(t_m2);
    MERGE_STAMP_ADD(a.frame, 1, t_m1, t_m2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MERGE_STAMP(t_m3);
    MERGE_STAMP_ADD(a.frame, 2, t_m2, t_m3);
    MERGE_STAMP_ADD(a.frame, 3, t_m_start, t_m3);
#endif
    return true;
}

// FREE (free schedule): the kernel also re-initialises finished slots (key 0) as new samples and
// sends them through the scene root in the same launch, runs end-group shaders on the spot
// (run_end_group) and carries the parked entries over, so no shading launch is needed.
// PLACE: one byte per candidate's place in its list, or two for scenes with lists of more than 256 boxes
// (24 instead of 20 KB of LDS per workgroup for the columns, i.e. six instead of seven workgroups per CU).
template <bool FREE, bool MAD, typename PLACE, int THREADS = kAabbThreads>
__device__ __forceinline__ void process_aabb(const ProcessArgs &a, uint32_t *spill_list, float (*s_pos)[THREADS],
                                             PLACE (*s_idx)[THREADS], uint32_t head_lds, uint32_t first_index, uint32_t tid)
{
    const uint32_t index = first_index + tid;
    DevGlobal *glob = a.frame.glob;
#if defined(RT_STAMPS) && RT_STAMPS == 3
    unsigned long long trace_t0;
    uint32_t trace_hw, trace_xcc;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trace_t0)::"memory");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(trace_hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(trace_xcc));
    unsigned long long *trace = a.frame.counters + size_t(kCounterShards) * kCounterKinds + (size_t(blockIdx.x) * (kAabbThreads / 64) + tid / 64) * 4;
    if ((tid & 63) == 0) { trace[0] = trace_t0; trace[1] = 0; trace[2] = ((unsigned long long)trace_xcc << 32) | trace_hw; trace[3] = 0; }
#endif
    // the grid covers the pool exactly: the entry is requested before the live count has arrived
    const rt_index_pair entry = a.frame.index[index];
    uint32_t key = entry.key;
    if (index >= step_constant(&glob->g.ray_count)) {
        if (FREE) a.frame.next[index] = entry;  // parked entry: not processed, carried over
        return;
    }
    uint32_t shader = (key >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK;
    if (FREE ? shader == RT_SH_MESH : shader != RT_SH_AABB) return;
    RaySlot *ray = &a.frame.rays[entry.slot];
    bool fresh = false;  // this lane starts a sample whose slot has not been written yet
    uint32_t sample = 0;
    if (FREE) {
        if (key == RT_SPAWN_GROUP) {  // a finished slot: the next sample starts here and visits the root below
            sample = blocked_sample(a.cam, step_constant(&glob->g.pixel_offset), step_constant(&glob->g.pixel_count), index);
            const uint32_t limit = step_constant(&glob->sample_limit);
            fresh = !(limit && sample >= limit);
            count_lanes(a.frame, kPrimarySpawned, fresh);
            key = fresh ? a.cam.root_group : a.group_count - 1;
            if (fresh && sample < step_constant(&glob->hit_capacity)) a.frame.hits[sample] = rt_hit_record{INFINITY, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            shader = (key >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK;
        }
        if (shader != RT_SH_AABB) {
            // a sample beyond the limit (dead-ray bin), a scene whose root is a leaf, or a shader key left by
            // rt_init_frame/rt_trace_closest: no box visit
            if (fresh) reset_ray<true>(ray, camera_ray<MAD>(a.cam, glob, sample), a.cam.root_group, a.cam.root_local, 0u, RT_SKY_GROUP);
            a.frame.next[index] = rt_index_pair{run_end_group<MAD>(key, ray, a), entry.slot};
            return;
        }
    }

    uint32_t new_key = key;
    bool done = true, visit = true;
    if (FREE && fresh) {
        // The samples that start here all enter the camera's root: a visit of its own, with the ray worked out in
        // registers and no pending entries.  A ray that comes out of an instance list at a model's root goes on below.
        uint32_t group_flags = 0;
        done = visit_aabb<kAabbCap, THREADS, false, true, MAD, true, PLACE>(uniform_value(key), ray, a, s_pos, s_idx, nullptr, nullptr, 0, tid,
                                                                     new_key, group_flags, sample);
        visit = done && (group_flags & RT_F_LOCAL0) && ((new_key >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK) == RT_SH_AABB;
        asm volatile("" ::: "memory");  // the next visit reads what this one has stored to the slot
        if (visit) key = new_key;
    }
    if (visit) for (int round = 0;; round++) {
        uint32_t group_flags = 0;
        bool todo = true;
        while (todo) {
            if (key != __builtin_amdgcn_readfirstlane(key)) continue;
            todo = false;
            const uint32_t g = uniform_value(key);  // every lane still here holds the same key
            done = visit_aabb<kAabbCap, THREADS, false, FREE, MAD, false, PLACE>(g, ray, a, s_pos, s_idx, nullptr, nullptr, 0, tid, new_key,
                                                                                  group_flags, 0, head_lds);
        }
        // Free schedule: a ray that has just been through a list of instances (the scene root) and whose
        // nearest node is a model's root goes through that root now.  A wave's rays enter few distinct
        // models, so the second visit is nearly as coherent as the first, and the ray saves a whole step.
        if (!FREE || round || !done || !(group_flags & RT_F_LOCAL0)) break;
        if (((new_key >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK) != RT_SH_AABB) break;
        asm volatile("" ::: "memory");  // the next visit reads what this one has stored to the slot
        key = new_key;
    }
    // Rare (3 visits in 10,000 on the benchmark scene): lanes whose columns overflowed and whose rays are
    // therefore untouched.  Their positions go to a list (one atomic per wave); k_process_aabb_big
    // performs those visits with one wave per ray.
    // (the position is worked out again: held across the visits it is spilled, and its reload from scratch waits for
    // every store of the visit, one counter for loads and stores)
    uint32_t tid_again = tid;
    asm volatile("" : "+v"(tid_again));
    const uint32_t index_again = first_index + tid_again;
    const unsigned long long spilled = __ballot(!done);
    if (spilled) {
        uint32_t base = 0;
        const uint32_t leader = (uint32_t)(__ffsll((long long)spilled) - 1);
        if (lane_id() == leader) base = atomicAdd(&glob->spill_count, (uint32_t)__popcll(spilled));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
        if (!done) {
            spill_list[base + __popcll(spilled & ((1ull << lane_id()) - 1))] = index_again;
            return;
        }
    }
    if (FREE) new_key = run_end_group<MAD>(new_key, ray, a);
    a.frame.next[index_again] = rt_index_pair{new_key, entry.slot};  // assign_index (ray-tracer.cl:171)
#if defined(RT_STAMPS) && RT_STAMPS == 3
    {
        unsigned long long trace_t1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trace_t1)::"memory");
        const unsigned long long m_ = __ballot(true);
        if (lane_id() == (uint32_t)(__ffsll((long long)m_) - 1)) { trace[1] = trace_t1; trace[3] = key; }
    }
#endif
}

template <bool FREE, bool MAD>
__attribute__((amdgpu_waves_per_eu(RT_AABB_WAVES, 8)))
__global__ __launch_bounds__(kAabbThreads) void k_process_aabb(ProcessArgs a, uint32_t *spill_list)
{
    __shared__ float s_pos[kAabbCap][kAabbThreads];
    __shared__ uint8_t s_idx[kAabbCap][kAabbThreads];
    __shared__ unsigned long long s_head[kAabbThreads];  // per lane: the head's two local ids (see visit_aabb)
    process_aabb<FREE, MAD, uint8_t>(a, spill_list, s_pos, s_idx, (uint32_t)(uintptr_t)&s_head[threadIdx.x], blockIdx.x * blockDim.x, threadIdx.x);
}

// The same with two-byte places, for scenes with box lists of more than 256 boxes.
template <bool FREE, bool MAD>
__global__ __launch_bounds__(kAabbThreads) void k_process_aabb_wide(ProcessArgs a, uint32_t *spill_list)
{
    __shared__ float s_pos[kAabbCap][kAabbThreads];
    __shared__ uint16_t s_idx[kAabbCap][kAabbThreads];
    __shared__ unsigned long long s_head[kAabbThreads];  // per lane: the head's two local ids (see visit_aabb)
    process_aabb<FREE, MAD, uint16_t>(a, spill_list, s_pos, s_idx, (uint32_t)(uintptr_t)&s_head[threadIdx.x], blockIdx.x * blockDim.x, threadIdx.x);
}

// The box visits the main kernel left undone, one wave per ray: a lone lane scanning a 256-box list and
// sorting 30 candidates in its own columns takes about 50 us, a wave sharing the work a few.
constexpr int kBigThreads = 256, kBigBlocks = 256;

template <bool FREE, bool MAD>
__global__ __launch_bounds__(kBigThreads) void k_process_aabb_big(ProcessArgs a, const uint32_t *spill_list)
{
    __shared__ float s_wave_pos[kBigThreads / 64][RT_MAX_HITS];   // per wave: the lists of wave_visit_one
    __shared__ uint16_t s_wave_idx[kBigThreads / 64][RT_MAX_HITS];
    const uint32_t tid = threadIdx.x, wave = tid / 64;
    const uint32_t total = a.frame.glob->spill_count;
    for (uint32_t at = blockIdx.x * (kBigThreads / 64) + wave; at < total; at += kBigBlocks * (kBigThreads / 64)) {
        const uint32_t index = spill_list[at];
        const uint32_t slot = uniform_value(a.frame.index[index].slot);
        RaySlot *ray = &a.frame.rays[slot];
        // the group the ray is in: the key of its index entry, except for a sample the box kernel has just started
        const uint32_t g = uniform_value(ray->queue[0].group_id);
        uint32_t new_key = g;
        uint32_t group_flags;
        visit_aabb<RT_MAX_QUEUE_LEN, kBigThreads, true, FREE, MAD, false, uint8_t>(g, ray, a, nullptr, nullptr, s_wave_pos[wave], s_wave_idx[wave], 0, tid,
                                                        new_key, group_flags);
        if (lane_id() == 0) {
            if (FREE) new_key = run_end_group<MAD>(new_key, ray, a);
            a.frame.next[index] = rt_index_pair{new_key, slot};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the lists are reused by the wave's next ray
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------- process, everything but box groups

// Leaf groups: triangle scan + hit commit.
template <bool FREE, bool MAD>
__device__ __forceinline__ void process_mesh(const ProcessArgs &a, uint32_t index);

template <bool FREE, bool MAD>
__attribute__((amdgpu_waves_per_eu(RT_MESH_WAVES, 8)))
__global__ __launch_bounds__(kMeshThreads) void k_process_mesh(ProcessArgs a)
{
    process_mesh<FREE, MAD>(a, blockIdx.x * blockDim.x + threadIdx.x);
}

template <bool FREE, bool MAD>
__device__ __forceinline__ void process_mesh(const ProcessArgs &a, uint32_t index)
{
    const rt_index_pair entry = a.frame.index[index];  // requested before the live count has arrived
    const uint32_t key = entry.key;
    if (index >= step_constant(&a.frame.glob->g.ray_count)) return;
    if (((key >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK) != RT_SH_MESH) return;
    RaySlot *ray = &a.frame.rays[entry.slot];

    uint32_t new_key = key;
    bool todo = true;
    while (todo) {
        if (key != __builtin_amdgcn_readfirstlane(key)) continue;
        todo = false;
        const uint32_t g = uniform_value(key);  // every lane still here holds the same key
        MISC_STAMP(t_mesh0);
        new_key = visit_mesh<FREE, MAD>(g, ray, a);
        MISC_STAMP(t_mesh1);
        MISC_STAMP_ADD(a.frame, 2, t_mesh0, t_mesh1);
    }
    if (FREE) new_key = run_end_group<MAD>(new_key, ray, a);
    a.frame.next[index] = rt_index_pair{new_key, entry.slot};  // assign_index (ray-tracer.cl:171)
}

// Shader groups (spawn, sky, light, material) and parked entries.
template <bool MAD>
__global__ __launch_bounds__(256) void k_process_shade(ProcessArgs a, uint32_t pool)
{
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    const DevGlobal *glob = a.frame.glob;
    if (index >= pool) return;
    const rt_index_pair entry = a.frame.index[index];
    if (index >= glob->g.ray_count) {  // parked entry: not processed, carried over
        a.frame.next[index] = entry;
        return;
    }
    const uint32_t key = entry.key;
    const uint32_t shader = (key >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK;
    if (shader == RT_SH_AABB || shader == RT_SH_MESH) return;
    RaySlot *ray = &a.frame.rays[entry.slot];
    const uint32_t dead = a.group_count - 1;
    const bool trace_only = (glob->mode & kModeTraceOnly) != 0;

    uint32_t new_key = key;
    bool todo = true;
    while (todo) {
        if (key != __builtin_amdgcn_readfirstlane(key)) continue;
        todo = false;
        const uint32_t g = uniform_value(key);  // every lane still here holds the same key
        MISC_STAMP(t_case0);
        switch ((g >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK) {
        case RT_SH_SPAWN: {
            bool started;
            new_key = init_ray<false, MAD>(a.cam, glob, ray, a.frame.hits, a.group_count, index + glob->g.pixel_offset, started);
            count_lanes(a.frame, kPrimarySpawned, started);
            break;
        }
        case RT_SH_SKY:
        case RT_SH_LIGHT:
            if (trace_only) {
                ray->queue[0].group_id = new_key = dead;
                break;
            }
            new_key = sky_or_light_shader<false>(a, ray, ((g >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK) == RT_SH_LIGHT);
            break;
        case RT_SH_MATERIAL: {
            if (trace_only) {
                ray->queue[0].group_id = new_key = dead;
                break;
            }
            const f32x4 color = *const_ptr<f32x4>(a.scene.grp + (g & kIdMask));
            new_key = mat_shader<false, MAD>(ray, make_float4(color.x, color.y, color.z, color.w), glob->shading);
            count_lanes(a.frame, kShadowSpawned, true);
            break;
        }
        default: {  // unknown shader type: the reference falls through to the pop (ray-tracer.cl:152-165)
            const uint32_t type_len = ray->type_len, queue_len = type_len >> 24;
            const float head_pos = ray->queue[0].pos;
            OldQueue old;
            load_pending(old, ray, queue_len);
            new_key = store_shifted(ray, old, type_len & 0xFFFFFFu, queue_len - 1, head_pos, ray->material_id);
            break;
        }
        }
        MISC_STAMP(t_case1);
        MISC_STAMP_ADD(a.frame, 3, t_case0, t_case1);
    }
    a.frame.next[index] = rt_index_pair{new_key, entry.slot};  // assign_index (ray-tracer.cl:171)
}

// ---------------------------------------------------------------- queue compaction (ray-tracer.cl:235-302)
//
// Two launches.  k_count_groups: one thread per group finds the group's run in the sorted list, splits
// it (plus the rays parked earlier) into whole warps and a parked remainder, and takes part in a
// block-wide prefix sum; the last block to finish turns the block totals into block offsets and
// updates the frame counters.  k_set_ray_index: one thread per entry moves it to its place, adding
// its group's block offset on the fly; its first blocks write the finished offsets into the group
// table.  No workgroup waits for another one.

constexpr uint32_t kCoarseKeys = 1024;

// count_groups (ray-tracer.cl:235-253) + update_groups (:255-290)
__global__ __launch_bounds__(kGroupThreads) void k_count_groups(DevGlobal *glob, rt_group_data *grp_data,
                                                                 const rt_index_pair *sorted, CompactWorkspace ws)
{
    __shared__ uint2 s_part[kGroupThreads];
    __shared__ uint32_t s_coarse[kCoarseKeys];
    __shared__ bool s_last;
    const uint32_t tid = threadIdx.x;
    const uint32_t g = blockIdx.x * kGroupThreads + tid;
    const uint32_t groups = glob->g.group_count, n = glob->g.ray_count;

    // E[id] = number of sorted entries whose group id is <= id, by binary search (the reference marks
    // boundaries per ray and fills the gaps serially).  The first ten steps of every search run on a
    // table of every stride-th key held in LDS, the rest on the list itself; both ends of the group's
    // run are searched in the same loop, so that their memory latencies overlap.
    uint32_t stride = 1;
    while ((n + stride - 1) / stride > kCoarseKeys) stride *= 2;
    const uint32_t samples = (n + stride - 1) / stride;  // s_coarse[j] = id of entry j * stride
    for (uint32_t j = tid; j < samples; j += kGroupThreads) s_coarse[j] = sorted[j * stride].key & kIdMask;
    __syncthreads();
    uint2 mine = make_uint2(0, 0);
    GroupMove m = GroupMove{0, 0, 0, 0, 0, {0, 0, 0}};
    if (g < groups) {
        // upper_bound on the samples: entries before sample number c are known to be <= id, the answer
        // lies in ((c - 1) * stride, c * stride]
        auto narrow = [&](bool inclusive, uint32_t id, uint32_t &lo, uint32_t &hi) {
            uint32_t a = 0, b = samples;
            while (a < b) {
                const uint32_t mid = a + (b - a) / 2;
                const uint32_t key = s_coarse[mid];
                if (inclusive ? key <= id : key < id) a = mid + 1;
                else b = mid;
            }
            lo = a ? (a - 1) * stride + 1 : 0;
            hi = a * stride < n ? a * stride : n;
            if (!a) hi = 0;
        };
        uint32_t lo0 = 0, hi0 = 0, lo1, hi1;
        if (g) narrow(false, g, lo0, hi0);
        narrow(true, g, lo1, hi1);
        while (lo0 < hi0 || lo1 < hi1) {
            const uint32_t mid0 = lo0 + (hi0 - lo0) / 2, mid1 = lo1 + (hi1 - lo1) / 2;
            const uint32_t k0 = lo0 < hi0 ? sorted[mid0].key & kIdMask : 0;
            const uint32_t k1 = lo1 < hi1 ? sorted[mid1].key & kIdMask : 0;
            if (lo0 < hi0) {
                if (k0 < g) lo0 = mid0 + 1;
                else hi0 = mid0;
            }
            if (lo1 < hi1) {
                if (k1 <= g) lo1 = mid1 + 1;
                else hi1 = mid1;
            }
        }
        const uint32_t main_count = lo1 - lo0;
        const rt_group_data grp = grp_data[g];
        const uint32_t total = main_count + grp.count[1];
        const uint32_t tail = g != groups - 1 ? total % glob->warp_width : total;  // the last group is the dead-ray bin
        m.base0 = lo0;
        m.base1 = grp.offset[1] - main_count;
        m.count1 = tail;
        mine = make_uint2(total - tail, tail);
        *reinterpret_cast<uint2 *>(grp_data[g].base) = make_uint2(m.base0, m.base1);
        *reinterpret_cast<uint2 *>(grp_data[g].count) = mine;
    }

    // Exclusive prefix sums of (whole-warp, parked) counts inside the block (ray-tracer.cl:276-290).
    uint2 incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ox = (uint32_t)__shfl_up((int)incl.x, off, 64), oy = (uint32_t)__shfl_up((int)incl.y, off, 64);
        if ((tid & 63) >= (uint32_t)off) {
            incl.x += ox;
            incl.y += oy;
        }
    }
    if ((tid & 63) == 63) s_part[tid >> 6] = incl;
    __syncthreads();
    uint2 before = make_uint2(0, 0), block_total = make_uint2(0, 0);
#pragma unroll
    for (int w = 0; w < kGroupThreads / 64; w++) {
        const uint2 t = s_part[w];
        if ((uint32_t)w < (tid >> 6)) {
            before.x += t.x;
            before.y += t.y;
        }
        block_total.x += t.x;
        block_total.y += t.y;
    }
    if (g < groups) {
        m.off0 = before.x + incl.x - mine.x;
        m.off1 = before.y + incl.y - mine.y;
        ws.move[g] = m;
    }
    if (g == 0) ws.block_sums[gridDim.x] = make_uint2(mine.x, 0);  // whole warps of group 0: the next step's new pixels

    // The last block to get here owns the block totals.
    if (tid == 0) {
        ws.block_sums[blockIdx.x] = block_total;
        __threadfence();
        s_last = atomicAdd(ws.ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const uint32_t blocks = gridDim.x;
    uint2 carry = make_uint2(0, 0);
    for (uint32_t first = 0; first < blocks; first += kGroupThreads) {  // one round up to 65,536 groups
        const uint2 sum = first + tid < blocks ? ws.block_sums[first + tid] : make_uint2(0, 0);
        s_part[tid] = sum;
        __syncthreads();
        for (uint32_t off = 1; off < kGroupThreads; off <<= 1) {
            const uint2 other = tid >= off ? s_part[tid - off] : make_uint2(0, 0);
            __syncthreads();
            s_part[tid].x += other.x;
            s_part[tid].y += other.y;
            __syncthreads();
        }
        const uint2 upto = s_part[tid], all = s_part[kGroupThreads - 1];
        if (first + tid < blocks) ws.block_sums[first + tid] = make_uint2(carry.x + upto.x - sum.x, carry.y + upto.y - sum.y);
        carry.x += all.x;
        carry.y += all.y;
        __syncthreads();
    }
    if (tid == 0) {
        glob->g.pixel_offset += glob->g.pixel_count;
        glob->g.pixel_count = ws.block_sums[blocks].x;
        glob->g.old_count = n;
        glob->g.ray_count = carry.x;  // all whole warps; the parked rays follow them
        glob->spill_count = 0;        // the next step's list of overflowed box visits starts empty
        *ws.ticket = 0;
    }
}

// set_ray_index (ray-tracer.cl:293-302)
__global__ __launch_bounds__(kGroupThreads) void k_set_ray_index(const DevGlobal *glob, rt_group_data *grp_data,
                                                                  CompactWorkspace ws, const rt_index_pair *src,
                                                                  rt_index_pair *dst, uint32_t count, uint32_t group_blocks)
{
    const uint32_t all_aligned = glob->g.ray_count;
    if (blockIdx.x < group_blocks) {  // finished offsets of this block's groups (ray-tracer.cl:283-289)
        const uint32_t g = blockIdx.x * kGroupThreads + threadIdx.x;
        if (g >= glob->g.group_count) return;
        const GroupMove m = ws.move[g];
        const uint2 block = ws.block_sums[blockIdx.x];
        *reinterpret_cast<uint2 *>(grp_data[g].offset) = make_uint2(block.x + m.off0, all_aligned + block.y + m.off1);
        return;
    }
    uint32_t index = (blockIdx.x - group_blocks) * kGroupThreads + threadIdx.x;
    if (index >= count) return;
    const rt_index_pair e = src[index];
    const uint32_t g = e.key & kIdMask;
    const GroupMove m = ws.move[g];
    const uint2 block = ws.block_sums[g / kGroupThreads];
    index -= index < glob->g.old_count ? m.base0 : m.base1;
    if (index < m.count1) index += all_aligned + block.y + m.off1;
    else index += block.x + m.off0 - m.count1;
    if (index < count) dst[index] = e;  // always true for consistent group data; keeps bad caller data from faulting
}

// Converts ray slots to the reference's RayQueue layout (ray-tracer.h:164-170) for rt_dump_state.  `stack`: the
// slots keep their pending entries the free schedule's way (entries behind queue_len read as zero).
__global__ void k_export_rays(const RaySlot *rays, rt_ray_queue *out, uint32_t count, bool stack)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const RaySlot &s = rays[i];
    rt_ray_queue q;
    q.weight[0] = s.weight.x;  q.weight[1] = s.weight.y;  q.weight[2] = s.weight.z;  q.weight[3] = s.weight.w;
    q.pixel = s.pixel;  q.type = s.type_len & 0xFFFFFFu;  q.material_id = s.material_id;  q.queue_len = s.type_len >> 24;
    q.start_min[0] = s.start[0];  q.start_min[1] = s.start[1];  q.start_min[2] = s.start[2];  q.start_min[3] = s.t_min;
    q.dir_max[0] = s.dir[0];  q.dir_max[1] = s.dir[1];  q.dir_max[2] = s.dir[2];  q.dir_max[3] = s.t_max;
    q.norm[0] = s.norm.x;  q.norm[1] = s.norm.y;  q.norm[2] = s.norm.z;  q.norm[3] = s.norm.w;
    q.root = rt_ray_hit{s.root.pos, s.root.group_id, {s.root.local0, s.root.local1}};
    q.orig = rt_ray_hit{s.orig.pos, s.orig.group_id, {s.orig.local0, s.orig.local1}};
    for (int k = 0; k < RT_MAX_QUEUE_LEN; k++) {
        RayHit e = s.queue[k];
        if (stack && k) e = (uint32_t)k < q.queue_len ? s.queue[q.queue_len - k] : RayHit{0.0f, 0u, 0u, 0u};
        q.queue[k] = rt_ray_hit{e.pos, e.group_id, {e.local0, e.local1}};
    }
    out[i] = q;
}

}  // namespace rtd

// ---------------------------------------------------------------- launchers (used by rt_api.hip)

#include "rt_launch.h"

namespace rtd {

static inline dim3 grid_for(uint32_t count, uint32_t block) { return dim3((count + block - 1) / block); }

void launch_init_frame(const LaunchCtx &c)
{
    ProcessArgs a{c.scene, c.frame, c.cam, c.group_count};
    k_init_groups<<<grid_for(c.group_count, 256), 256, 0, c.stream>>>(c.grp_data, c.group_count);
    if (c.mad) k_init_rays<true><<<grid_for(c.pool, 256), 256, 0, c.stream>>>(a, c.pool);
    else k_init_rays<false><<<grid_for(c.pool, 256), 256, 0, c.stream>>>(a, c.pool);
    (void)hipMemsetAsync(c.frame.area, 0, sizeof(float4) * size_t(c.area_size), c.stream);
}

void launch_load_rays(const LaunchCtx &c, const rt_ray *rays, uint32_t n_rays)
{
    ProcessArgs a{c.scene, c.frame, c.cam, c.group_count};
    k_init_groups<<<grid_for(c.group_count, 256), 256, 0, c.stream>>>(c.grp_data, c.group_count);
    k_load_rays<<<grid_for(c.pool, 256), 256, 0, c.stream>>>(a, c.pool, rays, n_rays);
}

template <bool FREE, bool MAD>
static void process_boxes(const LaunchCtx &c, const ProcessArgs &a)
{
    const dim3 grid = grid_for(c.pool, kAabbThreads);
    if (c.long_lists) k_process_aabb_wide<FREE, MAD><<<grid, kAabbThreads, 0, c.stream>>>(a, c.spill_list);
    else k_process_aabb<FREE, MAD><<<grid, kAabbThreads, 0, c.stream>>>(a, c.spill_list);
    k_process_aabb_big<FREE, MAD><<<kBigBlocks, kBigThreads, 0, c.stream>>>(a, c.spill_list);
}

void launch_process_boxes(const LaunchCtx &c)
{
    const ProcessArgs a{c.scene, c.frame, c.cam, c.group_count};
    if (c.free_schedule) {
        if (c.mad) process_boxes<true, true>(c, a);
        else process_boxes<true, false>(c, a);
    } else {
        if (c.mad) process_boxes<false, true>(c, a);
        else process_boxes<false, false>(c, a);
    }
}

void launch_process_leaves(const LaunchCtx &c, hipStream_t stream)
{
    ProcessArgs a{c.scene, c.frame, c.cam, c.group_count};
    const dim3 grid = grid_for(c.pool, kMeshThreads);
    if (c.free_schedule) {
        if (c.mad) k_process_mesh<true, true><<<grid, kMeshThreads, 0, stream>>>(a);
        else k_process_mesh<true, false><<<grid, kMeshThreads, 0, stream>>>(a);
    } else {
        if (c.mad) k_process_mesh<false, true><<<grid, kMeshThreads, 0, stream>>>(a);
        else k_process_mesh<false, false><<<grid, kMeshThreads, 0, stream>>>(a);
    }
}

void launch_process_shade(const LaunchCtx &c)
{
    ProcessArgs a{c.scene, c.frame, c.cam, c.group_count};
    if (c.mad) k_process_shade<true><<<grid_for(c.pool, 256), 256, 0, c.stream>>>(a, c.pool);
    else k_process_shade<false><<<grid_for(c.pool, 256), 256, 0, c.stream>>>(a, c.pool);
}

void launch_compact(const LaunchCtx &c, const rt_index_pair *sorted, rt_index_pair *dst)
{
    launch_compact_raw(c.stream, c.frame.glob, c.grp_data, c.compact, sorted, dst, c.pool, c.group_count);
}

uint32_t compact_group_blocks(uint32_t group_count) { return grid_for(group_count, kGroupThreads).x; }

void launch_compact_raw(hipStream_t stream, DevGlobal *glob, rt_group_data *grp_data, const CompactWorkspace &ws,
                        const rt_index_pair *sorted, rt_index_pair *dst, uint32_t count, uint32_t group_count)
{
    const uint32_t group_blocks = compact_group_blocks(group_count);
    k_count_groups<<<group_blocks, kGroupThreads, 0, stream>>>(glob, grp_data, sorted, ws);
    k_set_ray_index<<<group_blocks + grid_for(count, kGroupThreads).x, kGroupThreads, 0, stream>>>(glob, grp_data, ws, sorted, dst,
                                                                                                 count, group_blocks);
}

void launch_resolve(const LaunchCtx &c, void *out)
{
    k_resolve_rgba8<<<grid_for(c.area_size, 256), 256, 0, c.stream>>>(c.frame.area, static_cast<uchar4 *>(out), c.area_size);
}

void launch_export_rays(const LaunchCtx &c, rt_ray_queue *out)
{
    k_export_rays<<<grid_for(c.pool, 256), 256, 0, c.stream>>>(c.frame.rays, out, c.pool, c.free_schedule);
}

}  // namespace rtd
