// rt_device.h -- device-side data layout and small math helpers shared by the gfx950 kernels.
//
// Floating point contract (the parity tests rely on it): the library is built with
// -ffp-contract=off, every expression below is evaluated in float exactly as written, division and
// sqrt are the correctly rounded ones.  dot(a,b) = (ax*bx + ay*by) + az*bz,
// normalize(v) = v * (1 / sqrt(dot(v,v))), pow(x,5) and pow(x,100) are products of repeated squares.
// Kernels instantiated with MAD (contexts created with RT_FLAG_ARITH_MAD) fuse the multiply-adds that
// struct Arith<true> names, and only those; see include/rt_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_hip.h"

namespace rtd {

constexpr uint32_t kIdMask = RT_GROUP_ID_MASK;
constexpr uint32_t kWave = 64;
constexpr uint32_t kModeTraceOnly = 1u;  // shader groups park the ray instead of shading (rt_trace_closest)

struct RayHit {
    float pos;
    uint32_t group_id;
    uint32_t local0, local1;
};

// One ray slot, 256 bytes, 256-byte aligned: what the reference keeps per ray in its RayQueue
// (ray-tracer.h:164-170) plus the sample id.  The kernels are bound by the memory traffic of these slots --
// 128-byte lines fetched, lines dirtied, store requests -- so the layout follows the accesses:
//   * bytes 0-47 are what every visit reads (three 16-byte loads), byte 48 is the head of the queue; the fields
//     a visit changes lie next to each other so that one store covers them -- (type_len, t_min) after a visit,
//     (material_id, type_len, t_min, t_max) after a hit;
//   * the first 128-byte line also holds four pending entries: most visits touch this line only.
// The reference schedule keeps the pending entries as the reference does, queue[1..queue_len) nearest first,
// stale entries included.  The free schedule keeps them as a stack, nearest on top: queue[i], i >= 1, lies in
// queue[queue_len - i].  Popping the head then moves one entry, a merge stores only the entries that change
// place, and entries a hit has made unreachable (pos >= t_max: always the far end) stay until the next merge or
// pop counts them out.  A shadow ray's t_max, infinite in the reference for all its life, is that reach only.
struct alignas(256) RaySlot {
    uint32_t pixel, sample_id, material_id;        //   0
    uint32_t type_len;                             //  12  type (ray kind, bounces << 8) | queue_len << 24
    float t_min, t_max;                            //  16
    float start[3], dir[3];                        //  24, 36
    RayHit queue[RT_MAX_QUEUE_LEN];                //  48
    float4 weight;                                 // 176
    float4 norm;                                   // 192
    RayHit orig;                                   // 208
    RayHit root;                                   // 224
    uint32_t pad[4];                               // 240
};
static_assert(sizeof(RaySlot) == 256, "ray slot must be 256 bytes");

// Triangle of a leaf, expanded when the scene is uploaded: first corner, two edges and their cross
// product (the values mesh_shader derives per triangle, shader.cl:126-127), 48 bytes, fetched with
// scalar loads because every lane of a wave is in the same leaf.  On the device the records of triangles
// 2k and 2k + 1 are interleaved value by value (r0.x of both, r0.y of both, ...): see visit_mesh.
struct alignas(16) TriRecord {
    float r0[3], p[3], q[3], n[3];
};
static_assert(sizeof(TriRecord) == 48, "triangle record must be 48 bytes");

// Device-resident frame state: the reference's GlobalData (ray-tracer.h:106-111) followed by what the
// reference fixes at build time or does not have.
struct alignas(16) DevGlobal {
    rt_global_data g;
    uint32_t warp_width, sample_limit, hit_capacity, row_offset;
    uint32_t mode, sort_count, spill_count, shading;  // sort_count: element count for stand-alone sort calls;
                                                       // spill_count: box visits left to k_process_aabb_big this step;
                                                       // shading: bits 8-31 of rt_config.flags shifted down (mode, bounces, seed)
};

// Queue compaction: what moving an entry of group g needs, and the per-block prefix sums.
constexpr int kGroupThreads = 256;  // groups per workgroup of the counting kernel
struct GroupMove {
    uint32_t base0, base1;  // first sorted position of the group's new entries / of its parked entries, as in rt_group_data
    uint32_t count1;        // rays parked after this step
    uint32_t off0, off1;    // whole-warp and parked rays of the earlier groups of the same block
    uint32_t pad[3];
};
static_assert(sizeof(GroupMove) == 32, "GroupMove is read as one 32-byte record");
struct CompactWorkspace {
    GroupMove *move;     // [group_count]
    uint2 *block_sums;   // [blocks + 1]: totals per block of kGroupThreads groups, then their exclusive prefix sums; last: group 0's whole warps
    uint32_t *ticket;    // blocks finished; zero between launches
};

// One 32-byte record of the culling data of a box list: the bounds of a cluster of boxes
// (laid out like a DevBox) or the 256-bit membership mask of a cluster.
union alignas(32) ClusterRecord {
    float f[8];
    uint32_t u[8];
};
// Device copy of a box group's entry: {first box, boxes, flags | clusters << 8 | masked << 16, first cluster record}.
constexpr uint32_t kClusterCountShift = 8, kClusterCountMask = 0xFF, kClusterMasked = 1u << 16, kClusterLong = 1u << 17;
// A list's clusters sit in a 32-bit mask while its boxes are scanned, 256 boxes at a time; a longer list
// (kClusterLong; up to kMaxListBoxes boxes, two bytes per place in the candidate columns) has one cluster per
// eight consecutive boxes and is scanned part by part.
constexpr uint32_t kMaxClusteredBoxes = 256, kMaxListBoxes = 65536;

// Device copy of a box (rt_aabb re-ordered): the x and y of either corner and the two z share aligned register
// pairs, so that the slab test's three subtract-multiply (or fused multiply-add) steps each do two values at once.
struct alignas(32) DevBox {
    float lo_x, lo_y, hi_x, hi_y, lo_z, hi_z;
    uint32_t group_id, local_id;
};
static_assert(sizeof(DevBox) == 32, "a box is one 32-byte record");

struct Scene {
    const rt_group *grp;
    const rt_matrix *mat;
    const DevBox *aabb;
    const ClusterRecord *cluster;
    const rt_vertex *vtx;
    const uint32_t *tri;
    const TriRecord *tri_rec;
};

// Event counters are sharded: every wave adds to the shard of its workgroup, one 64-byte line each, and
// the host sums the shards.  Thousands of waves adding to ONE word serialise at about 12 ns per atomic,
// which is as long as a whole kernel here, and every wave would sit out its own atomic at its next
// wait for a load (vmcnt counts atomics too).
constexpr uint32_t kCounterShards = 1024;
enum Counter : uint32_t {
    kPrimarySpawned = 0, kShadowSpawned = 1,
    kBoxTests = 2, kTriTests = 3,  // what the reference's loops would test (every box / triangle of a visited group, per ray)
    kBoxScanned = 4,               // cluster and box tests actually executed per ray (builds with RT_COUNT_SCANNED only; shares its place with the phase stamps of diagnostic builds)
    kCounterKinds = 8
};

struct Frame {
    DevGlobal *glob;
    unsigned long long *counters;  // [kCounterShards][kCounterKinds]
    float4 *area;
    RaySlot *rays;
    rt_index_pair *index;        // (group key, slot) in processing order, input of a step
    rt_index_pair *next;         // the same pairs with the keys after the step
    rt_hit_record *hits;
};

// ---------------------------------------------------------------- vector helpers

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 xyz(const float4 &v) { return V3{v.x, v.y, v.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Two floats in a register pair: subtract, add, multiply and fused multiply-add work on both at once (v_pk_*_f32).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat(float x) { return f32x2{x, x}; }

// 1.0f / w for both halves, correctly rounded: the sequence the compiler emits for one float division (scale,
// estimate, one Newton step, three residual corrections, un-scale, special cases), with the six multiply-adds of
// its core done for both halves at once.  Same bits as 1.0f / w.x and 1.0f / w.y for every float
// (tools/micro/rcp_exact.hip runs all 2^32 of them).
__device__ __forceinline__ f32x2 reciprocal2(f32x2 w)
{
    bool unused0, unused1, scaled0, scaled1;
    const f32x2 d = {__builtin_amdgcn_div_scalef(1.0f, w.x, false, &unused0), __builtin_amdgcn_div_scalef(1.0f, w.y, false, &unused1)};
    const f32x2 n = {__builtin_amdgcn_div_scalef(1.0f, w.x, true, &scaled0), __builtin_amdgcn_div_scalef(1.0f, w.y, true, &scaled1)};
    const f32x2 y0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 e = __builtin_elementwise_fma(-d, y0, splat(1.0f));
    const f32x2 y1 = __builtin_elementwise_fma(e, y0, y0);
    const f32x2 q0 = n * y1;
    const f32x2 r0 = __builtin_elementwise_fma(-d, q0, n);
    const f32x2 q1 = __builtin_elementwise_fma(r0, y1, q0);
    const f32x2 r1 = __builtin_elementwise_fma(-d, q1, n);
    return f32x2{__builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r1.x, y1.x, q1.x, scaled0), w.x, 1.0f),
                 __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r1.y, y1.y, q1.y, scaled1), w.y, 1.0f)};
}

// The fields of a slot every visit reads, in registers: bytes 0-47 of the slot, three 16-byte requests.
struct HotRay {
    uint32_t pixel, sample_id, material_id;
    uint32_t type, queue_len;  // (one word in the slot)
    float t_min, t_max;
    V3 start, dir;
};
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ HotRay load_hot(const RaySlot *ray)
{
    const uint4 a = *reinterpret_cast<const uint4 *>(ray);
    const float4 b = *reinterpret_cast<const float4 *>(&ray->t_min), c = *reinterpret_cast<const float4 *>(&ray->start[2]);
    return HotRay{a.x, a.y, a.z, a.w & 0xFFFFFFu, a.w >> 24, b.x, b.y, V3{b.z, b.w, c.x}, V3{c.y, c.z, c.w}};
}
__device__ __forceinline__ void store_hot(RaySlot *ray, const HotRay &h)
{
    *reinterpret_cast<uint4 *>(ray) = make_uint4(h.pixel, h.sample_id, h.material_id, h.type | h.queue_len << 24);
    *reinterpret_cast<float4 *>(&ray->t_min) = make_float4(h.t_min, h.t_max, h.start.x, h.start.y);
    *reinterpret_cast<float4 *>(&ray->start[2]) = make_float4(h.start.z, h.dir.x, h.dir.y, h.dir.z);
}
// queue_len and t_min, one 8-byte request (save_queue, ray-tracer.cl:157-165); `type` is the ray's unchanged type
__device__ __forceinline__ void store_progress(RaySlot *ray, uint32_t type, uint32_t queue_len, float t_min)
{
    *reinterpret_cast<u32x2_a4 *>(&ray->type_len) = u32x2_a4{type | queue_len << 24, __float_as_uint(t_min)};
}
// material_id, queue_len, t_min and t_max, one 16-byte request (insert_stop, ray-tracer.cl:173-203)
__device__ __forceinline__ void store_commit(RaySlot *ray, uint32_t material_id, uint32_t type, uint32_t queue_len, float t_min,
                                             float t_max)
{
    *reinterpret_cast<u32x4_a4 *>(&ray->material_id) =
        u32x4_a4{material_id, type | queue_len << 24, __float_as_uint(t_min), __float_as_uint(t_max)};
}

__device__ __forceinline__ V3 normalize(V3 a) { return a * (1.0f / sqrtf(dot(a, a))); }

// The arithmetic of the two modes (rt_config.flags, RT_FLAG_ARITH_MAD).  Arith<false>: every product and sum is
// rounded on its own.  Arith<true>: a product that feeds a sum is fused with it at exactly these places.
template <bool MAD>
struct Arith {
    static __device__ __forceinline__ float dot(V3 a, V3 b)
    {
        if (MAD) return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
        return a.x * b.x + a.y * b.y + a.z * b.z;
    }
    static __device__ __forceinline__ V3 cross(V3 a, V3 b)
    {
        if (MAD)
            return V3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
        return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    }
    // a * s + b
    static __device__ __forceinline__ V3 madd(V3 a, float s, V3 b)
    {
        if (MAD) return V3{__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z)};
        return V3{a.x * s + b.x, a.y * s + b.y, a.z * s + b.z};
    }
    // (a * s + b * t) + c * u
    static __device__ __forceinline__ V3 lin3(V3 a, float s, V3 b, float t, V3 c, float u)
    {
        if (MAD)
            return V3{__builtin_fmaf(c.x, u, __builtin_fmaf(b.x, t, a.x * s)), __builtin_fmaf(c.y, u, __builtin_fmaf(b.y, t, a.y * s)),
                      __builtin_fmaf(c.z, u, __builtin_fmaf(b.z, t, a.z * s))};
        return (a * s + b * t) + c * u;
    }
    static __device__ __forceinline__ V3 normalize(V3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
    // the same for two values per operand (packed FP32)
    static __device__ __forceinline__ f32x2 dot2(f32x2 ax, f32x2 ay, f32x2 az, f32x2 bx, f32x2 by, f32x2 bz)
    {
        if (MAD) return __builtin_elementwise_fma(az, bz, __builtin_elementwise_fma(ay, by, ax * bx));
        return (ax * bx + ay * by) + az * bz;
    }
    // p * q - r * s
    static __device__ __forceinline__ f32x2 cross2(f32x2 p, f32x2 q, f32x2 r, f32x2 s)
    {
        if (MAD) return __builtin_elementwise_fma(p, q, -(r * s));
        return p * q - r * s;
    }
};

__device__ __forceinline__ float pow5(float x)
{
    const float x2 = x * x;
    return (x2 * x2) * x;
}
__device__ __forceinline__ float pow100(float x)
{
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16, x64 = x32 * x32;
    return (x64 * x32) * x4;
}

// Counter-based random numbers for the diffuse bounce: a 32-bit integer hash of (seed, sample, bounce, draw).
__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;  x *= 0x7FEB352Du;
    x ^= x >> 15;  x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float uniform01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// Pointers into the constant address space: loads through them with a wave-uniform address are
// scalar loads (s_load_dwordx4/x8), which is how node payload reaches a wave whose lanes all sit in
// the same group.  Only used on scene arrays, which no kernel writes.
template <typename T>
using ConstPtr = const __attribute__((address_space(4))) T *;

template <typename T>
__device__ __forceinline__ ConstPtr<T> const_ptr(const void *p)
{
    return (ConstPtr<T>)(uintptr_t)p;
}

// Built-in vector types: unlike HIP's float4/uint4 structs they can be loaded through ConstPtr.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Phase timing for diagnostic builds: RT_STAMP(t) reads the shader clock into t, RT_STAMP_ADD adds a
// difference to counter 4 + (slot & 3) of the workgroup's shard once per wave.  Both compile to nothing in the shipped library.
#ifdef RT_STAMPS
#define RT_STAMP(var) \
    unsigned long long var; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define RT_STAMP_ADD(frame, slot, from, to) \
    do { \
        const unsigned long long m_ = __ballot(true); \
        if (lane_id() == (uint32_t)(__ffsll((long long)m_) - 1)) \
            atomicAdd(&(frame).counters[(blockIdx.x % kCounterShards) * kCounterKinds + 4 + ((slot) & 3)], (to) - (from)); \
    } while (0)
#else
#define RT_STAMP(var)
#define RT_STAMP_ADD(frame, slot, from, to)
#endif

// The value of v in the first active lane, in a scalar register.  Call it where all active lanes are
// known to hold the same v: written as an asm statement so that the optimiser can neither fold it into
// an earlier readfirstlane nor replace it by the per-lane v it was proven equal to -- either would turn
// the scalar loads that depend on it back into per-lane loads.
__device__ __forceinline__ uint32_t uniform_value(uint32_t v)
{
    uint32_t s;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(v));
    return s;
}

// A field of the frame state that no kernel of the process stage writes (limits, modes, the counts the grouping left),
// read through the scalar path: one request for the wave, and not among the vector requests -- a vector load between
// two stores makes the wave wait for the first store as well.
template <typename T>
__device__ __forceinline__ T step_constant(const T *field)
{
    return *const_ptr<T>(field);
}

// Waits for every vector load issued so far (and every store: they share the counter).
__device__ __forceinline__ void vector_loads_landed()
{
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt untouched
}

// Waits for every scalar load (and LDS access) issued so far.  Scalar loads return out of order, so a
// wave can only wait for all of them; in a software-pipelined scan this goes BEFORE the request for the
// next batch, otherwise the first use of the current batch waits for the batch just requested as well.
__device__ __forceinline__ void scalar_loads_landed()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt and expcnt untouched
}

// Largest v (0..127) among the ACTIVE lanes of the wave, by bisection with ballots; unlike a shuffle
// reduction it never looks at the registers of lanes that are switched off.
__device__ __forceinline__ uint32_t wave_max_upto127(uint32_t v)
{
    uint32_t best = 0;
#pragma unroll
    for (int bit = 6; bit >= 0; bit--) {
        const uint32_t trial = best | (1u << bit);
        if (__ballot(v >= trial)) best = trial;
    }
    return best;
}

}  // namespace rtd
