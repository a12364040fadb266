// rt_api.hip -- the C ABI of include/rt_hip.h: context, buffers, launch sequence, readbacks.
//
// Host-side successor of the reference's OpenCL plumbing (cl-helper.h; main.cpp:87-127, 219-265,
// 345-477).  One context = one GPU = one stream.  No OpenCL, no run-time compilation: the kernels are
// gfx950 code objects inside this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "rt_launch.h"

using namespace rtd;

struct rt_ctx {
    rt_config cfg{};
    hipStream_t stream = nullptr, side = nullptr;  // side: leaf kernel, concurrent with the box and shading kernels
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint32_t pool = 0, area_size = 0, group_count = 0;
    bool scene_ready = false, frame_ready = false;

    DevGlobal *glob = nullptr;
    unsigned long long *counters = nullptr;
    float4 *area = nullptr;
    RaySlot *rays = nullptr;
    rt_index_pair *index = nullptr, *next = nullptr, *sorted = nullptr;
    rt_hit_record *hits = nullptr;
    uint32_t *spill_list = nullptr;
    rt_group_data *grp_data = nullptr;
    CompactWorkspace compact = {nullptr, nullptr, nullptr};
    FusedWorkspace fused = {nullptr, nullptr};  // free schedule with at most kFusedMaxGroups groups: sort + compaction as one sequence
    uchar4 *rgba = nullptr;
    SortWorkspace ws{};
    hipGraphExec_t step_graph = nullptr;  // one step (both streams), captured on first use; dropped when its arguments change
    bool use_graph = true;
    bool free_schedule = false;  // RT_FLAG_FREE_SCHEDULE
    bool long_lists = false;     // the uploaded scene has a box list of more than kMaxClusteredBoxes boxes
    bool mad = false;            // RT_FLAG_ARITH_MAD

    rt_group *grp = nullptr;
    rt_matrix *mat = nullptr;
    DevBox *aabb = nullptr;
    ClusterRecord *cluster = nullptr;
    rt_vertex *vtx = nullptr;
    uint32_t *tri = nullptr;
    TriRecord *tri_rec = nullptr;
    rt_camera cam{};

    uint64_t steps = 0;
    bool profiling = false;
    bool serial = false;  // stage timing with the process kernels one after another on one stream
    struct Span {
        int stage;
        hipEvent_t from, to;
    };
    std::vector<Span> pending;  // one per stage and timed step
    std::vector<hipEvent_t> spare;
    double stage_ms[RT_STAGE_COUNT] = {0, 0, 0, 0, 0};
    uint64_t timed_steps = 0;

    std::string last_error;
};

namespace {

int fail(rt_ctx *ctx, int status, const char *fmt, ...)
{
    char text[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(text, sizeof(text), fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = text;
    return status;
}

#define RT_HIP(ctx, call)                                                                                   \
    do {                                                                                                    \
        hipError_t err_ = (call);                                                                           \
        if (err_ != hipSuccess)                                                                             \
            return fail(ctx, err_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "%s: %s", #call, \
                        hipGetErrorString(err_));                                                           \
    } while (0)

template <typename T>
hipError_t dev_alloc(T *&ptr, size_t count)
{
    return hipMalloc(reinterpret_cast<void **>(&ptr), (count ? count : 1) * sizeof(T));
}

template <typename T>
void dev_free(T *&ptr)
{
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
}

LaunchCtx launch_ctx(const rt_ctx *c)
{
    LaunchCtx l{};
    l.stream = c->stream;
    l.scene = Scene{c->grp, c->mat, c->aabb, c->cluster, c->vtx, c->tri, c->tri_rec};
    l.frame = Frame{c->glob, c->counters, c->area, c->rays, c->index, c->next, c->hits};
    l.cam = c->cam;
    l.grp_data = c->grp_data;
    l.spill_list = c->spill_list;
    l.compact = c->compact;
    l.pool = c->pool;
    l.area_size = c->area_size;
    l.group_count = c->group_count;
    l.free_schedule = c->free_schedule;
    l.long_lists = c->long_lists;
    l.mad = c->mad;
    return l;
}

int check_launch(rt_ctx *ctx, const char *what)
{
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(ctx, RT_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(err));
    return RT_OK;
}

void free_compact(CompactWorkspace &ws)
{
    dev_free(ws.move);
    dev_free(ws.block_sums);
    dev_free(ws.ticket);
}

hipError_t alloc_compact(CompactWorkspace &ws, uint32_t group_count)
{
    hipError_t err = dev_alloc(ws.move, group_count);
    if (err == hipSuccess) err = dev_alloc(ws.block_sums, compact_group_blocks(group_count) + 1);
    if (err == hipSuccess) err = dev_alloc(ws.ticket, 1);
    if (err == hipSuccess) err = hipMemset(ws.ticket, 0, sizeof(uint32_t));
    return err;
}

void drop_step_graph(rt_ctx *c)
{
    if (c->step_graph) (void)hipGraphExecDestroy(c->step_graph);
    c->step_graph = nullptr;
}

void free_scene(rt_ctx *c)
{
    drop_step_graph(c);
    dev_free(c->grp);
    dev_free(c->mat);
    dev_free(c->aabb);
    dev_free(c->cluster);
    dev_free(c->vtx);
    dev_free(c->tri);
    dev_free(c->tri_rec);
    dev_free(c->grp_data);
    free_compact(c->compact);
    dev_free(c->fused.ghist);
    dev_free(c->fused.block_main);
    c->scene_ready = c->frame_ready = false;
}

// Expands one leaf's packed triangles into TriRecords with the float operations mesh_shader performs
// per triangle (shader.cl:126-127), so the kernel's results do not depend on where they are done.
// (`mad`: the cross product's components as fused multiply-adds, as Arith<true>::cross has them.)
void expand_leaf(const rt_mesh_shader &mesh, const rt_vertex *vtx, const uint32_t *tri, TriRecord *out, bool mad)
{
    for (uint32_t i = 0; i < mesh.tri_count; i++) {
        const uint32_t word = tri[mesh.tri_offs + i];
        const float *a = vtx[mesh.vtx_offs + (word & 0x3FF)].pos;
        const float *b = vtx[mesh.vtx_offs + ((word >> 10) & 0x3FF)].pos;
        const float *c = vtx[mesh.vtx_offs + ((word >> 20) & 0x3FF)].pos;
        TriRecord &r = out[i];
        for (int k = 0; k < 3; k++) {
            r.r0[k] = a[k];
            r.p[k] = b[k] - a[k];
            r.q[k] = c[k] - a[k];
        }
        if (mad) {
            r.n[0] = std::fmaf(r.p[1], r.q[2], -(r.p[2] * r.q[1]));
            r.n[1] = std::fmaf(r.p[2], r.q[0], -(r.p[0] * r.q[2]));
            r.n[2] = std::fmaf(r.p[0], r.q[1], -(r.p[1] * r.q[0]));
        } else {
            r.n[0] = r.p[1] * r.q[2] - r.p[2] * r.q[1];
            r.n[1] = r.p[2] * r.q[0] - r.p[0] * r.q[2];
            r.n[2] = r.p[0] * r.q[1] - r.p[1] * r.q[0];
        }
    }
}

// Wave-level culling data of one box list (see visit_aabb): the boxes are grouped into clusters of at
// most eight, each with bounds that enclose its boxes strictly (one float step outwards on every
// side, so that a ray parallel to a face of a box still meets the cluster's open slab).  A wave tests
// the cluster bounds first and scans only the boxes of clusters some lane enters; a lane that misses
// a cluster misses every box inside it (the slab test is monotonic in the bounds, rounding included),
// so the accepted boxes, their order and their entry parameters are exactly those of a full scan.
// Clusters are runs of eight consecutive boxes when the list is spatially ordered (the lists the
// hierarchy builder emits are), otherwise median-split groups with a 256-bit membership mask each.
struct ClusterSet {
    std::vector<ClusterRecord> bounds, masks;  // masks empty: cluster c holds boxes [8c, 8c + 8)
};

ClusterRecord cluster_bounds(const rt_aabb *boxes, const std::vector<uint32_t> &members)
{
    ClusterRecord r;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {1e30f, 1e30f, 1e30f};  // empty: no ray enters
    bool any = false;
    for (uint32_t i : members) {
        const rt_aabb &b = boxes[i];
        for (int k = 0; k < 3; k++) {
            lo[k] = any ? std::min(lo[k], b.min[k]) : b.min[k];
            hi[k] = any ? std::max(hi[k], b.max[k]) : b.max[k];
        }
        any = true;
    }
    static const int lo_at[3] = {0, 1, 4}, hi_at[3] = {2, 3, 5};  // the layout of a DevBox
    for (int k = 0; k < 3; k++) {
        r.f[lo_at[k]] = any ? std::nextafterf(lo[k], -INFINITY) : lo[k];
        r.f[hi_at[k]] = any ? std::nextafterf(hi[k], INFINITY) : hi[k];
    }
    r.f[6] = r.f[7] = 0;
    return r;
}

double surface(const ClusterRecord &r)
{
    const double x = double(r.f[2]) - r.f[0], y = double(r.f[3]) - r.f[1], z = double(r.f[5]) - r.f[4];
    return x * y + y * z + z * x;
}

void split_spatially(const rt_aabb *boxes, std::vector<uint32_t> ids, std::vector<std::vector<uint32_t>> &out)
{
    if (ids.size() <= 8) {
        std::sort(ids.begin(), ids.end());
        out.push_back(std::move(ids));
        return;
    }
    float lo[3], hi[3];
    auto centre = [&](uint32_t i, int k) { return 0.5f * (boxes[i].min[k] + boxes[i].max[k]); };
    for (int k = 0; k < 3; k++) {
        lo[k] = hi[k] = centre(ids[0], k);
        for (uint32_t i : ids) {
            lo[k] = std::min(lo[k], centre(i, k));
            hi[k] = std::max(hi[k], centre(i, k));
        }
    }
    int axis = 0;
    for (int k = 1; k < 3; k++)
        if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
    std::stable_sort(ids.begin(), ids.end(), [&](uint32_t x, uint32_t y) { return centre(x, axis) < centre(y, axis); });
    const size_t half = (ids.size() + 1) / 2;
    split_spatially(boxes, std::vector<uint32_t>(ids.begin(), ids.begin() + half), out);
    split_spatially(boxes, std::vector<uint32_t>(ids.begin() + half, ids.end()), out);
}

// One cluster per eight consecutive boxes.
ClusterSet run_clusters(const rt_aabb *boxes, uint32_t count)
{
    ClusterSet runs;
    for (uint32_t first = 0; first < count; first += 8) {
        std::vector<uint32_t> members;
        for (uint32_t i = first; i < count && i < first + 8; i++) members.push_back(i);
        runs.bounds.push_back(cluster_bounds(boxes, members));
    }
    return runs;
}

ClusterSet build_clusters(const rt_aabb *boxes, uint32_t count)
{
    ClusterSet runs = run_clusters(boxes, count), spatial;
    double cost_runs = 0, cost_spatial = 0;
    for (const ClusterRecord &r : runs.bounds) cost_runs += surface(r);
    if (count <= 16) return runs;
    std::vector<uint32_t> all(count);
    for (uint32_t i = 0; i < count; i++) all[i] = i;
    std::vector<std::vector<uint32_t>> groups;
    split_spatially(boxes, all, groups);
    for (const std::vector<uint32_t> &members : groups) {
        spatial.bounds.push_back(cluster_bounds(boxes, members));
        cost_spatial += surface(spatial.bounds.back());
        ClusterRecord mask;
        std::memset(&mask, 0, sizeof(mask));
        for (uint32_t i : members) mask.u[i / 32] |= 1u << (i % 32);
        spatial.masks.push_back(mask);
    }
    // membership masks cost a load per box in the scan: only for lists that are far from spatial order
    return cost_spatial < 0.5 * cost_runs && spatial.bounds.size() <= 32 ? spatial : runs;
}

hipError_t set_global_field(rt_ctx *c, size_t offset, const void *src, size_t size)
{
    return hipMemcpyAsync(reinterpret_cast<char *>(c->glob) + offset, src, size, hipMemcpyHostToDevice, c->stream);
}

int resolve_timers(rt_ctx *c)
{
    if (c->pending.empty()) return RT_OK;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->side) RT_HIP(c, hipStreamSynchronize(c->side));
    for (const rt_ctx::Span &span : c->pending) {
        float ms = 0;
        RT_HIP(c, hipEventElapsedTime(&ms, span.from, span.to));
        c->stage_ms[span.stage] += ms;
        if (span.stage == RT_STAGE_SORT) c->timed_steps++;
        c->spare.push_back(span.from);
        c->spare.push_back(span.to);
    }
    c->pending.clear();
    return RT_OK;
}

int new_event(rt_ctx *c, hipEvent_t &ev, hipStream_t stream)
{
    if (!c->spare.empty()) {
        ev = c->spare.back();
        c->spare.pop_back();
    } else {
        RT_HIP(c, hipEventCreate(&ev));
    }
    RT_HIP(c, hipEventRecord(ev, stream));
    return RT_OK;
}

// Brackets what `launch` enqueues on `stream` with two events when stage timing is on.
template <typename Launch>
int timed(rt_ctx *c, int stage, hipStream_t stream, Launch launch)
{
    if (!c->profiling) {
        launch();
        return RT_OK;
    }
    rt_ctx::Span span{stage, nullptr, nullptr};
    int rc = new_event(c, span.from, stream);
    if (rc) return rc;
    launch();
    if ((rc = new_event(c, span.to, stream))) return rc;
    c->pending.push_back(span);
    return RT_OK;
}

// One make_step (main.cpp:439-477): process, sort by group, re-pack.  The step starts and ends with
// the live index in c->index, so the launch sequence is the same every step.
int one_step(rt_ctx *c, bool run_process)
{
    const LaunchCtx l = launch_ctx(c);
    int rc;
    if (run_process) {
        // Box, leaf and shader groups touch disjoint rays within a step.  The leaf kernel runs on the
        // side stream beside the other two; the sort waits for both streams.
        const bool fork = c->side && !c->serial;
        hipStream_t leaf_stream = fork ? c->side : c->stream;
        if (fork) {
            RT_HIP(c, hipEventRecord(c->ev_fork, c->stream));
            RT_HIP(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        }
        if ((rc = timed(c, RT_STAGE_TRAVERSE, c->stream, [&] { launch_process_boxes(l); }))) return rc;
        if ((rc = timed(c, RT_STAGE_LEAVES, leaf_stream, [&] { launch_process_leaves(l, leaf_stream); }))) return rc;
        if (fork) RT_HIP(c, hipEventRecord(c->ev_join, c->side));
        // free schedule: the visiting kernels shade, spawn and carry the parked entries over themselves
        if (!c->free_schedule && (rc = timed(c, RT_STAGE_SHADE, c->stream, [&] { launch_process_shade(l); }))) return rc;
        if (fork) RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    if (c->fused.ghist) {
        // free schedule: histogram, group scan and scatter passes as one sequence; the last pass writes the re-packed index
        // (the process kernels are done with the old index by then, so it is written in place of it)
        if (!run_process)
            RT_HIP(c, hipMemcpyAsync(c->next, c->index, sizeof(rt_index_pair) * size_t(c->pool), hipMemcpyDeviceToDevice, c->stream));
        rc = timed(c, RT_STAGE_SORT, c->stream, [&] {
            launch_fused(c->stream, c->ws, c->fused, c->compact, c->glob, c->grp_data, c->next, c->index, c->pool, c->group_count);
        });
        if (rc) return rc;
    } else {
        rc = timed(c, RT_STAGE_SORT, c->stream, [&] {
            launch_sort(c->stream, c->ws, run_process ? c->next : c->index, c->sorted, c->pool, &c->glob->g.ray_count, c->group_count,
                        c->free_schedule);
        });
        if (rc) return rc;
        if ((rc = timed(c, RT_STAGE_COMPACT, c->stream, [&] { launch_compact(l, c->sorted, c->index); }))) return rc;
    }
    if (c->profiling && c->pending.size() >= 5 * 2048 && (rc = resolve_timers(c))) return rc;
    return check_launch(c, "make_step");
}

}  // namespace

extern "C" {

const char *rt_status_string(int status)
{
    switch (status) {
    case RT_OK: return "RT_OK";
    case RT_ERR_INVALID_VALUE: return "RT_ERR_INVALID_VALUE";
    case RT_ERR_NO_DEVICE: return "RT_ERR_NO_DEVICE";
    case RT_ERR_OUT_OF_MEMORY: return "RT_ERR_OUT_OF_MEMORY";
    case RT_ERR_LAUNCH: return "RT_ERR_LAUNCH";
    case RT_ERR_NOT_READY: return "RT_ERR_NOT_READY";
    case RT_ERR_UNSUPPORTED: return "RT_ERR_UNSUPPORTED";
    case RT_ERR_HIP: return "RT_ERR_HIP";
    default: return "RT_ERR_UNKNOWN";
    }
}

const char *rt_last_error(rt_ctx *ctx)
{
    return ctx ? ctx->last_error.c_str() : "";
}

int rt_create(const rt_config *cfg, rt_ctx **out)
{
    if (!cfg || !out) return RT_ERR_INVALID_VALUE;
    *out = nullptr;
    if (!cfg->width || !cfg->height || !cfg->ray_count || (cfg->flags & ~(RT_FLAG_FREE_SCHEDULE | RT_FLAG_ARITH_MAD | RT_FLAG_SHADING_MASK)))
        return RT_ERR_INVALID_VALUE;
    if (((cfg->flags >> 8) & 15u) > RT_SHADE_DIFFUSE) return RT_ERR_INVALID_VALUE;
    if (uint64_t(cfg->width) * cfg->height > 0x7FFFFFFFull) return RT_ERR_INVALID_VALUE;
    if (uint64_t(cfg->ray_count) + RT_SORT_TILE > 0x7FFFFFFFull) return RT_ERR_INVALID_VALUE;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return RT_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= n_dev) return RT_ERR_NO_DEVICE;

    rt_ctx *c = new rt_ctx;
    c->cfg = *cfg;
    c->free_schedule = (cfg->flags & RT_FLAG_FREE_SCHEDULE) != 0;
    c->mad = (cfg->flags & RT_FLAG_ARITH_MAD) != 0;
    if (!c->cfg.warp_width) c->cfg.warp_width = 64;
    c->pool = (cfg->ray_count + RT_SORT_TILE - 1) / RT_SORT_TILE * RT_SORT_TILE;
    c->area_size = cfg->width * cfg->height;
    *out = c;  // so that rt_last_error / rt_destroy work on a half-built context

    RT_HIP(c, hipSetDevice(cfg->device));
    (void)sort_static_setup(nullptr);  // rt_sort_pairs_static's workspace on this device, looked up while no stream is being
                                       // captured; that entry point reports a failure itself, a context does not need it
    (void)hipGetLastError();
    RT_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->use_graph = getenv("RT_NO_GRAPH") == nullptr;
    if (!getenv("RT_NO_OVERLAP")) {
        RT_HIP(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        RT_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        RT_HIP(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    RT_HIP(c, dev_alloc(c->glob, 1));
#if defined(RT_STAMPS) && RT_STAMPS == 3
    RT_HIP(c, dev_alloc(c->counters, size_t(kCounterShards) * kCounterKinds + size_t(c->pool / 64) * 4));  // + per-wave trace
#else
    RT_HIP(c, dev_alloc(c->counters, size_t(kCounterShards) * kCounterKinds));
#endif
    RT_HIP(c, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long) * kCounterShards * kCounterKinds, c->stream));
    RT_HIP(c, dev_alloc(c->area, c->area_size));
    RT_HIP(c, dev_alloc(c->rays, c->pool));
    RT_HIP(c, dev_alloc(c->index, c->pool));
    RT_HIP(c, dev_alloc(c->next, c->pool));
    RT_HIP(c, dev_alloc(c->sorted, c->pool));
    RT_HIP(c, dev_alloc(c->ws.tmp, c->pool));
    RT_HIP(c, dev_alloc(c->hits, cfg->hit_capacity));
    RT_HIP(c, dev_alloc(c->rgba, c->area_size));
    RT_HIP(c, dev_alloc(c->spill_list, c->pool));
    c->ws.tiles = c->pool / RT_SORT_TILE;
    RT_HIP(c, dev_alloc(c->ws.counts, size_t(256) * c->ws.tiles * kSortCountsPerTile));
    RT_HIP(c, dev_alloc(c->ws.totals, 256));
    RT_HIP(c, hipMemsetAsync(c->glob, 0, sizeof(DevGlobal), c->stream));
    RT_HIP(c, hipMemsetAsync(c->hits, 0, sizeof(rt_hit_record) * size_t(cfg->hit_capacity ? cfg->hit_capacity : 1), c->stream));
    RT_HIP(c, hipMemsetAsync(c->rays, 0, sizeof(RaySlot) * size_t(c->pool), c->stream));
    RT_HIP(c, hipMemsetAsync(c->area, 0, sizeof(float4) * size_t(c->area_size), c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_destroy(rt_ctx *c)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (const rt_ctx::Span &span : c->pending) {
        (void)hipEventDestroy(span.from);
        (void)hipEventDestroy(span.to);
    }
    for (hipEvent_t ev : c->spare) (void)hipEventDestroy(ev);
    free_scene(c);
    dev_free(c->glob);
    dev_free(c->counters);
    dev_free(c->area);
    dev_free(c->rays);
    dev_free(c->index);
    dev_free(c->next);
    dev_free(c->sorted);
    dev_free(c->ws.tmp);
    dev_free(c->ws.counts);
    dev_free(c->ws.totals);
    dev_free(c->hits);
    dev_free(c->rgba);
    dev_free(c->spill_list);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

int rt_upload_scene(rt_ctx *c, const rt_global_data *data, const rt_group *grp, size_t n_groups, const rt_matrix *mat,
                    size_t n_mat, const rt_aabb *aabb, size_t n_aabb, const rt_vertex *vtx, size_t n_vtx,
                    const uint32_t *tri, size_t n_tri)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    if (!data || !grp || !aabb) return fail(c, RT_ERR_INVALID_VALUE, "null scene array");
    if ((n_mat && !mat) || (n_vtx && !vtx) || (n_tri && !tri)) return fail(c, RT_ERR_INVALID_VALUE, "null scene array");
    const uint32_t G = data->group_count;
    if (uint64_t(data->cam.width) * data->cam.height != c->area_size)
        return fail(c, RT_ERR_INVALID_VALUE, "camera %ux%u does not match the context's image", data->cam.width, data->cam.height);
    if (G < 8 || G > RT_GROUP_ID_MASK || n_groups > G - 1)
        return fail(c, RT_ERR_INVALID_VALUE, "group_count %u must exceed the %zu groups given (last group is the dead-ray bin)", G, n_groups);
    if (n_groups < 3) return fail(c, RT_ERR_INVALID_VALUE, "the three predefined groups are missing");

    // Shape checks: every index a kernel will follow must stay inside the arrays given.
    auto shader_of = [](uint32_t id) { return (id >> RT_GROUP_SH_SHIFT) & RT_GROUP_SH_MASK; };
    std::vector<uint8_t> kind(n_groups, 0xFF);
    std::vector<uint32_t> work;
    bool any_ortho = false, any_long = false;
    auto visit = [&](uint32_t id, const char *where) -> int {
        const uint32_t gi = id & RT_GROUP_ID_MASK, sh = shader_of(id);
        if (((id >> RT_GROUP_TR_SHIFT) & RT_GROUP_TR_MASK) == RT_TR_ORTHO) any_ortho = true;
        if (gi >= n_groups) return fail(c, RT_ERR_INVALID_VALUE, "%s refers to group %u of %zu", where, gi, n_groups);
        if (sh != RT_SH_AABB && sh != RT_SH_MESH) return fail(c, RT_ERR_INVALID_VALUE, "%s: group %u is not a box or mesh group", where, gi);
        // The reference's transform() sets the local ray for tr_identity and tr_ortho only; tr_affine is an empty "TODO" case
        // and tr_none has none (shader.cl:45-72), so a ray entering such a group there is scanned with an unset ray.
        const uint32_t tr = (id >> RT_GROUP_TR_SHIFT) & RT_GROUP_TR_MASK;
        if (tr != RT_TR_IDENTITY && tr != RT_TR_ORTHO)
            return fail(c, RT_ERR_UNSUPPORTED, "%s: group %u has transform type %u; the reference defines a result for tr_identity and tr_ortho only", where, gi, tr);
        if (kind[gi] == 0xFF) {
            kind[gi] = uint8_t(sh);
            work.push_back(id);
        } else if (kind[gi] != sh) {
            return fail(c, RT_ERR_INVALID_VALUE, "group %u is referenced with two shader types", gi);
        }
        return RT_OK;
    };
    int rc = visit(data->cam.root_group, "camera root");
    if (rc) return rc;
    // Device copies of the lists the kernels scan, re-laid out: every box list starts on a multiple of
    // eight boxes and is filled up to one with boxes no ray can enter, every leaf's triangles start on a
    // multiple of two and are filled up with null triangles no ray can hit, and both arrays end with
    // spare entries.  The scan loops then need no tail handling and may read ahead past a list's end.
    // The device group table carries the re-laid-out offsets.
    std::vector<rt_group> dev_grp(grp, grp + n_groups);
    std::vector<DevBox> dev_boxes;
    std::vector<ClusterRecord> dev_clusters;
    ClusterRecord no_cluster;
    for (int k = 0; k < 8; k++) no_cluster.f[k] = k >= 6 ? 0.0f : 1e30f;
    std::vector<uint32_t> dev_tri;
    std::vector<TriRecord> dev_rec;
    const DevBox never = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 0u, 0u};  // zero-size: exit parameter never exceeds entry
    TriRecord null_tri;
    std::memset(&null_tri, 0, sizeof(null_tri));  // zero normal: t is NaN, every comparison fails
    while (!work.empty()) {
        const uint32_t id = work.back();
        work.pop_back();
        const rt_group &g = grp[id & RT_GROUP_ID_MASK];
        if (shader_of(id) == RT_SH_AABB) {
            const rt_aabb_shader &s = g.aabb;
            if (uint64_t(s.aabb_offs) + s.aabb_count > n_aabb)
                return fail(c, RT_ERR_INVALID_VALUE, "group %u: boxes [%u, +%u) exceed %zu", id & RT_GROUP_ID_MASK, s.aabb_offs, s.aabb_count, n_aabb);
            if (s.aabb_count > kMaxListBoxes)
                return fail(c, RT_ERR_UNSUPPORTED, "group %u has %u boxes; at most %u per group are supported", id & RT_GROUP_ID_MASK, s.aabb_count, kMaxListBoxes);
            for (uint32_t i = 0; i < s.aabb_count; i++) {
                const rt_aabb &b = aabb[s.aabb_offs + i];
                if ((rc = visit(b.group_id, "a box"))) return rc;
                if ((s.flags & RT_F_LOCAL0) && b.local_id >= n_mat)
                    return fail(c, RT_ERR_INVALID_VALUE, "box %u: instance %u of %zu matrices", s.aabb_offs + i, b.local_id, n_mat);
            }
            uint32_t *entry = reinterpret_cast<uint32_t *>(&dev_grp[id & RT_GROUP_ID_MASK]);
            entry[0] = uint32_t(dev_boxes.size());
            for (uint32_t i = 0; i < s.aabb_count; i++) {
                const rt_aabb &b = aabb[s.aabb_offs + i];
                dev_boxes.push_back(DevBox{b.min[0], b.min[1], b.max[0], b.max[1], b.min[2], b.max[2], b.group_id, b.local_id});
            }
            while (dev_boxes.size() % 8) dev_boxes.push_back(never);
            // clusters of the list: [bounds, filled up to a multiple of four][membership masks, if any]
            // (a list of more than kMaxClusteredBoxes boxes gets one cluster per eight consecutive boxes; the kernel works out
            // how many of them each part of 256 boxes has)
            const bool long_list = s.aabb_count > kMaxClusteredBoxes;
            any_long = any_long || long_list;
            const ClusterSet set = long_list ? run_clusters(aabb + s.aabb_offs, s.aabb_count) : build_clusters(aabb + s.aabb_offs, s.aabb_count);
            const uint32_t clusters = long_list ? 0u : uint32_t(set.bounds.size());
            entry[2] = (s.flags & 0xFFu) | clusters << kClusterCountShift | (set.masks.empty() ? 0u : kClusterMasked) | (long_list ? kClusterLong : 0u);
            entry[3] = uint32_t(dev_clusters.size());
            dev_clusters.insert(dev_clusters.end(), set.bounds.begin(), set.bounds.end());
            while (dev_clusters.size() % 4) dev_clusters.push_back(no_cluster);
            dev_clusters.insert(dev_clusters.end(), set.masks.begin(), set.masks.end());
            while (dev_clusters.size() % 4) dev_clusters.push_back(no_cluster);
        } else {
            const rt_mesh_shader &s = g.mesh;
            if (uint64_t(s.tri_offs) + s.tri_count > n_tri)
                return fail(c, RT_ERR_INVALID_VALUE, "group %u: triangles [%u, +%u) exceed %zu", id & RT_GROUP_ID_MASK, s.tri_offs, s.tri_count, n_tri);
            for (uint32_t i = 0; i < s.tri_count; i++) {
                const uint32_t w = tri[s.tri_offs + i];
                const uint32_t top = std::max(std::max(w & 0x3FF, (w >> 10) & 0x3FF), (w >> 20) & 0x3FF);
                if (uint64_t(s.vtx_offs) + top >= n_vtx)
                    return fail(c, RT_ERR_INVALID_VALUE, "group %u: vertex %u exceeds %zu", id & RT_GROUP_ID_MASK, s.vtx_offs + top, n_vtx);
            }
            const uint32_t mid = s.material_id & RT_GROUP_ID_MASK;
            if (mid >= n_groups) return fail(c, RT_ERR_INVALID_VALUE, "group %u: material group %u of %zu", id & RT_GROUP_ID_MASK, mid, n_groups);
            // The group a ray goes to after a hit is run as a shader.  A box or leaf group there would be entered
            // without having been checked and re-laid out above, so it is not accepted.
            if (shader_of(s.material_id) > RT_SH_MATERIAL)
                return fail(c, RT_ERR_INVALID_VALUE, "group %u: its material group %u is not a spawn, sky, light or material group", id & RT_GROUP_ID_MASK, mid);
            const size_t at = dev_tri.size();
            dev_grp[id & RT_GROUP_ID_MASK].mesh.tri_offs = uint32_t(at);
            dev_tri.insert(dev_tri.end(), tri + s.tri_offs, tri + s.tri_offs + s.tri_count);
            dev_rec.resize(at + s.tri_count);
            expand_leaf(s, vtx, tri, dev_rec.data() + at, c->mad);
            if (dev_tri.size() % 2) {
                dev_tri.push_back(0);
                dev_rec.push_back(null_tri);
            }
        }
    }
    dev_boxes.insert(dev_boxes.end(), 8, never);
    dev_clusters.insert(dev_clusters.end(), 8, no_cluster);
    dev_tri.insert(dev_tri.end(), 6, 0u);
    dev_rec.insert(dev_rec.end(), 6, null_tri);
    if (dev_boxes.size() > 0x7FFFFFFFu || dev_tri.size() > 0x7FFFFFFFu) return fail(c, RT_ERR_UNSUPPORTED, "scene too large");
    // A group entered with an instance transform takes its matrix from the ray's first local id: the number a box
    // of an instance list gave it (checked above), or else the camera's.
    if (any_ortho && data->cam.root_local >= n_mat)
        return fail(c, RT_ERR_INVALID_VALUE, "camera root instance %u of %zu matrices", data->cam.root_local, n_mat);

    RT_HIP(c, hipSetDevice(c->cfg.device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    c->group_count = G;
    c->long_lists = any_long;
    c->cam = data->cam;
    RT_HIP(c, dev_alloc(c->grp, G));
    RT_HIP(c, dev_alloc(c->mat, n_mat));
    RT_HIP(c, dev_alloc(c->aabb, dev_boxes.size()));
    RT_HIP(c, dev_alloc(c->cluster, dev_clusters.size()));
    RT_HIP(c, dev_alloc(c->vtx, n_vtx));
    RT_HIP(c, dev_alloc(c->tri, dev_tri.size()));
    RT_HIP(c, dev_alloc(c->tri_rec, dev_rec.size()));
    RT_HIP(c, dev_alloc(c->grp_data, G));
    RT_HIP(c, alloc_compact(c->compact, G));
    // One counter per group in the rank kernel's LDS: where the device grants a workgroup that much (160 KB on gfx950, so
    // up to kFusedMaxGroups groups); otherwise, and for larger scenes, the plain-order sort and the two compaction kernels.
    if (c->free_schedule && G <= kFusedMaxGroups && fused_setup(G) == hipSuccess) {
        RT_HIP(c, dev_alloc(c->fused.ghist, size_t(G) * fused_hist_slices()));
        RT_HIP(c, dev_alloc(c->fused.block_main, compact_group_blocks(G)));
        RT_HIP(c, hipMemset(c->fused.ghist, 0, sizeof(uint32_t) * size_t(G) * fused_hist_slices()));
    } else {
        (void)hipGetLastError();
    }
    RT_HIP(c, hipMemset(c->grp, 0, sizeof(rt_group) * size_t(G)));
    RT_HIP(c, hipMemset(c->grp_data, 0, sizeof(rt_group_data) * size_t(G)));
    RT_HIP(c, hipMemcpy(c->grp, dev_grp.data(), sizeof(rt_group) * n_groups, hipMemcpyHostToDevice));
    if (n_mat) RT_HIP(c, hipMemcpy(c->mat, mat, sizeof(rt_matrix) * n_mat, hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->aabb, dev_boxes.data(), sizeof(DevBox) * dev_boxes.size(), hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->cluster, dev_clusters.data(), sizeof(ClusterRecord) * dev_clusters.size(), hipMemcpyHostToDevice));
    if (n_vtx) RT_HIP(c, hipMemcpy(c->vtx, vtx, sizeof(rt_vertex) * n_vtx, hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->tri, dev_tri.data(), sizeof(uint32_t) * dev_tri.size(), hipMemcpyHostToDevice));
    // The leaf kernel tests triangles two at a time with both triangles' values side by side: records 2k and
    // 2k + 1 (leaves start on even indices and are filled up to even counts) are interleaved value by value.
    for (size_t k = 0; k + 1 < dev_rec.size(); k += 2) {
        float both[2][12], mixed[24];
        std::memcpy(both, &dev_rec[k], sizeof(both));
        for (int v = 0; v < 12; v++) {
            mixed[2 * v] = both[0][v];
            mixed[2 * v + 1] = both[1][v];
        }
        std::memcpy(&dev_rec[k], mixed, sizeof(mixed));
    }
    RT_HIP(c, hipMemcpy(c->tri_rec, dev_rec.data(), sizeof(TriRecord) * dev_rec.size(), hipMemcpyHostToDevice));

    DevGlobal init;
    std::memset(&init, 0, sizeof(init));
    init.g = *data;
    init.g.ray_count = c->pool;
    init.g.pixel_offset = init.g.pixel_count = init.g.old_count = 0;
    init.warp_width = c->cfg.warp_width;
    init.sample_limit = c->cfg.sample_limit;
    init.hit_capacity = c->cfg.hit_capacity;
    init.row_offset = c->cfg.row_offset;
    init.shading = c->cfg.flags >> 8;
    RT_HIP(c, hipMemcpy(c->glob, &init, sizeof(init), hipMemcpyHostToDevice));
    c->scene_ready = true;
    return RT_OK;
}

int rt_init_frame(rt_ctx *c)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    if (!c->scene_ready) return fail(c, RT_ERR_NOT_READY, "rt_init_frame before rt_upload_scene");
    RT_HIP(c, hipSetDevice(c->cfg.device));
    RT_HIP(c, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long) * kCounterShards * kCounterKinds, c->stream));
    launch_init_frame(launch_ctx(c));
    c->steps = 0;
    c->frame_ready = true;
    return check_launch(c, "init_frame");
}

int rt_step(rt_ctx *c, uint32_t n_steps)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    if (!c->frame_ready) return fail(c, RT_ERR_NOT_READY, "rt_step before rt_init_frame");
    RT_HIP(c, hipSetDevice(c->cfg.device));
    for (uint32_t s = 0; s < n_steps; s++) {
        // Every launch of a step has the same arguments from step to step (the counts that change live
        // in device memory), so the step is captured once, fork and join of the second stream
        // included, and replayed: the dependencies between its kernels then cost less than events
        // recorded from the host.
        if (c->use_graph && !c->profiling) {
            if (!c->step_graph) {
                hipGraph_t graph = nullptr;
                RT_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                const int rc = one_step(c, true);
                const hipError_t ended = hipStreamEndCapture(c->stream, &graph);
                if (rc) {
                    if (graph) (void)hipGraphDestroy(graph);
                    return rc;
                }
                RT_HIP(c, ended);
                const hipError_t made = hipGraphInstantiate(&c->step_graph, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                RT_HIP(c, made);
            }
            RT_HIP(c, hipGraphLaunch(c->step_graph, c->stream));
        } else {
            const int rc = one_step(c, true);
            if (rc) return rc;
        }
        c->steps++;
    }
    return RT_OK;
}

int rt_set_warp_width(rt_ctx *c, uint32_t warp_width)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    if (!c->scene_ready) return fail(c, RT_ERR_NOT_READY, "rt_set_warp_width before rt_upload_scene");
    c->cfg.warp_width = warp_width ? warp_width : 64;
    RT_HIP(c, set_global_field(c, offsetof(DevGlobal, warp_width), &c->cfg.warp_width, sizeof(uint32_t)));
    RT_HIP(c, hipStreamSynchronize(c->stream));  // the source of the async copy is a context field
    return RT_OK;
}

int rt_sync(rt_ctx *c)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_resolve_rgba8(rt_ctx *c, void *dst, int dst_is_device)
{
    if (!c || (!dst && !dst_is_device)) return RT_ERR_INVALID_VALUE;
    RT_HIP(c, hipSetDevice(c->cfg.device));
    launch_resolve(launch_ctx(c), dst_is_device && dst ? dst : c->rgba);
    int rc = check_launch(c, "resolve");
    if (rc) return rc;
    if (!dst_is_device) RT_HIP(c, hipMemcpyAsync(dst, c->rgba, size_t(c->area_size) * 4, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_resolve_rgba8_enqueue(rt_ctx *c, void *dst_device)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    RT_HIP(c, hipSetDevice(c->cfg.device));
    launch_resolve(launch_ctx(c), dst_device ? dst_device : c->rgba);
    return check_launch(c, "resolve");
}

int rt_read_area(rt_ctx *c, float *dst)
{
    if (!c || !dst) return RT_ERR_INVALID_VALUE;
    RT_HIP(c, hipMemcpyAsync(dst, c->area, sizeof(float4) * size_t(c->area_size), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_get_global(rt_ctx *c, rt_global_data *dst)
{
    if (!c || !dst) return RT_ERR_INVALID_VALUE;
    RT_HIP(c, hipMemcpyAsync(dst, c->glob, sizeof(rt_global_data), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_current_ray(rt_ctx *c, uint32_t *pixel_offset)
{
    if (!c || !pixel_offset) return RT_ERR_INVALID_VALUE;
    rt_global_data g;
    const int rc = rt_get_global(c, &g);
    if (rc) return rc;
    *pixel_offset = g.pixel_offset;
    return RT_OK;
}

namespace {

int read_counters(rt_ctx *c, unsigned long long sums[kCounterKinds], bool reset)
{
    std::vector<unsigned long long> shards(size_t(kCounterShards) * kCounterKinds);
    RT_HIP(c, hipMemcpyAsync(shards.data(), c->counters, shards.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (reset) RT_HIP(c, hipMemsetAsync(c->counters, 0, shards.size() * sizeof(unsigned long long), c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t k = 0; k < kCounterKinds; k++) sums[k] = 0;
    for (size_t i = 0; i < shards.size(); i++) sums[i % kCounterKinds] += shards[i];
    return RT_OK;
}

}  // namespace

int rt_get_counters(rt_ctx *c, rt_counters *dst)
{
    if (!c || !dst) return RT_ERR_INVALID_VALUE;
    unsigned long long sums[kCounterKinds];
    const int rc = read_counters(c, sums, false);
    if (rc) return rc;
    dst->primary_spawned = sums[kPrimarySpawned];
    dst->shadow_spawned = sums[kShadowSpawned];
    dst->steps = c->steps;
    dst->box_tests = sums[kBoxTests];
    dst->tri_tests = sums[kTriTests];
    return RT_OK;
}

int rt_sort_pairs(rt_ctx *c, const rt_index_pair *in, rt_index_pair *out, uint32_t count, uint32_t n, uint32_t group_count,
                  int on_device)
{
    if (!c || !in || !out) return RT_ERR_INVALID_VALUE;
    if (!count || count % RT_SORT_TILE || count > c->pool) return fail(c, RT_ERR_INVALID_VALUE, "count %u must be a multiple of %u within the pool of %u", count, RT_SORT_TILE, c->pool);
    if (n > count || !group_count || group_count > RT_GROUP_ID_MASK + 1) return fail(c, RT_ERR_INVALID_VALUE, "bad n or group_count");
    RT_HIP(c, hipSetDevice(c->cfg.device));
    const size_t bytes = sizeof(rt_index_pair) * size_t(count);
    rt_index_pair *d_in = nullptr, *d_out = nullptr;
    int rc = RT_OK;
    auto step = [&](hipError_t err, const char *what) {
        if (err != hipSuccess && !rc) rc = fail(c, err == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
    };
    if (!on_device) {
        step(dev_alloc(d_in, count), "alloc");
        step(dev_alloc(d_out, count), "alloc");
        if (!rc) step(hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, c->stream), "upload");
    }
    if (!rc) step(set_global_field(c, offsetof(DevGlobal, sort_count), &n, sizeof(n)), "count");
    if (!rc) {
        launch_sort(c->stream, c->ws, on_device ? in : d_in, on_device ? out : d_out, count, &c->glob->sort_count, group_count,
                    c->free_schedule);
        rc = check_launch(c, "sort");
    }
    if (!on_device && !rc) step(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream), "copy back");
    step(hipStreamSynchronize(c->stream), "sort");  // also: n is read from this call's frame until the copy has run
    dev_free(d_in);
    dev_free(d_out);
    return rc;
}

int rt_compact_groups(rt_ctx *c, const rt_index_pair *sorted, rt_index_pair *out, uint32_t count, rt_group_data *grp_data,
                      rt_global_data *global, uint32_t warp_width)
{
    if (!c || !sorted || !out || !grp_data || !global) return RT_ERR_INVALID_VALUE;
    if (!count || !warp_width || !global->group_count || global->ray_count > count)
        return fail(c, RT_ERR_INVALID_VALUE, "bad count, warp width, group_count or ray_count");
    for (uint32_t i = 0; i < count; i++)
        if ((sorted[i].key & RT_GROUP_ID_MASK) >= global->group_count)
            return fail(c, RT_ERR_INVALID_VALUE, "pair %u: group %u of %u", i, sorted[i].key & RT_GROUP_ID_MASK, global->group_count);
    RT_HIP(c, hipSetDevice(c->cfg.device));
    const uint32_t G = global->group_count;
    DevGlobal host;
    std::memset(&host, 0, sizeof(host));
    host.g = *global;
    host.warp_width = warp_width;
    DevGlobal *d_glob = nullptr;
    rt_group_data *d_grp = nullptr;
    rt_index_pair *d_in = nullptr, *d_out = nullptr;
    CompactWorkspace ws = {nullptr, nullptr, nullptr};
    int rc = RT_OK;
    auto step = [&](hipError_t err, const char *what) {
        if (err != hipSuccess && !rc) rc = fail(c, RT_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
    };
    step(alloc_compact(ws, G), "alloc");
    step(dev_alloc(d_glob, 1), "alloc");
    step(dev_alloc(d_grp, G), "alloc");
    step(dev_alloc(d_in, count), "alloc");
    step(dev_alloc(d_out, count), "alloc");
    if (!rc) {
        step(hipMemcpyAsync(d_glob, &host, sizeof(host), hipMemcpyHostToDevice, c->stream), "upload");
        step(hipMemcpyAsync(d_grp, grp_data, sizeof(rt_group_data) * size_t(G), hipMemcpyHostToDevice, c->stream), "upload");
        step(hipMemcpyAsync(d_in, sorted, sizeof(rt_index_pair) * size_t(count), hipMemcpyHostToDevice, c->stream), "upload");
        step(hipMemsetAsync(d_out, 0, sizeof(rt_index_pair) * size_t(count), c->stream), "clear");
    }
    if (!rc) {
        launch_compact_raw(c->stream, d_glob, d_grp, ws, d_in, d_out, count, G);
        rc = check_launch(c, "compact");
    }
    if (!rc) {
        step(hipMemcpyAsync(&host, d_glob, sizeof(host), hipMemcpyDeviceToHost, c->stream), "download");
        step(hipMemcpyAsync(grp_data, d_grp, sizeof(rt_group_data) * size_t(G), hipMemcpyDeviceToHost, c->stream), "download");
        step(hipMemcpyAsync(out, d_out, sizeof(rt_index_pair) * size_t(count), hipMemcpyDeviceToHost, c->stream), "download");
    }
    step(hipStreamSynchronize(c->stream), "compact");
    if (!rc) *global = host.g;
    free_compact(ws);
    dev_free(d_glob);
    dev_free(d_grp);
    dev_free(d_in);
    dev_free(d_out);
    return rc;
}

int rt_dump_state(rt_ctx *c, rt_ray_queue *ray_list, rt_index_pair *ray_index, rt_group_data *grp_data)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    if (!c->scene_ready) return fail(c, RT_ERR_NOT_READY, "no scene");
    RT_HIP(c, hipSetDevice(c->cfg.device));
    if (ray_list) {
        rt_ray_queue *d_out = nullptr;
        RT_HIP(c, hipStreamSynchronize(c->stream));
        RT_HIP(c, dev_alloc(d_out, c->pool));
        launch_export_rays(launch_ctx(c), d_out);
        const hipError_t err = hipMemcpyAsync(ray_list, d_out, sizeof(rt_ray_queue) * size_t(c->pool), hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);
        dev_free(d_out);
        if (err != hipSuccess) return fail(c, RT_ERR_HIP, "dump rays: %s", hipGetErrorString(err));
    }
    if (ray_index) RT_HIP(c, hipMemcpyAsync(ray_index, c->index, sizeof(rt_index_pair) * size_t(c->pool), hipMemcpyDeviceToHost, c->stream));
    if (grp_data) RT_HIP(c, hipMemcpyAsync(grp_data, c->grp_data, sizeof(rt_group_data) * size_t(c->group_count), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "dump");
}

int rt_read_hits(rt_ctx *c, rt_hit_record *dst, uint32_t first, uint32_t count)
{
    if (!c || !dst) return RT_ERR_INVALID_VALUE;
    if (uint64_t(first) + count > c->cfg.hit_capacity) return fail(c, RT_ERR_INVALID_VALUE, "hits [%u, +%u) exceed capacity %u", first, count, c->cfg.hit_capacity);
    RT_HIP(c, hipMemcpyAsync(dst, c->hits + first, sizeof(rt_hit_record) * size_t(count), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_trace_closest(rt_ctx *c, const rt_ray *rays, uint32_t n_rays, rt_hit_record *hits)
{
    if (!c || (n_rays && (!rays || !hits))) return RT_ERR_INVALID_VALUE;
    if (!c->scene_ready) return fail(c, RT_ERR_NOT_READY, "rt_trace_closest before rt_upload_scene");
    if (n_rays > c->pool) return fail(c, RT_ERR_INVALID_VALUE, "%u rays exceed the pool of %u", n_rays, c->pool);
    RT_HIP(c, hipSetDevice(c->cfg.device));
    c->frame_ready = false;

    rt_ray *d_rays = nullptr;
    rt_hit_record *d_hits = nullptr, *saved_hits = c->hits;
    int rc = RT_OK;
    auto step = [&](hipError_t err, const char *what) {
        if (err != hipSuccess && !rc) rc = fail(c, err == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
    };
    step(dev_alloc(d_rays, n_rays), "alloc");
    step(dev_alloc(d_hits, n_rays), "alloc");
    if (rc) {
        dev_free(d_rays);
        dev_free(d_hits);
        return rc;
    }
    // sample ids 0..n-1 get a hit record, later ids are parked; granularity 1; shaders park the ray
    const uint32_t settings[5] = {1u, n_rays ? n_rays : 1u, n_rays, c->cfg.row_offset, kModeTraceOnly};
    step(set_global_field(c, offsetof(DevGlobal, warp_width), settings, sizeof(settings)), "settings");
    if (n_rays) step(hipMemcpyAsync(d_rays, rays, sizeof(rt_ray) * size_t(n_rays), hipMemcpyHostToDevice, c->stream), "upload rays");
    c->hits = d_hits;
    if (!rc) {
        launch_load_rays(launch_ctx(c), d_rays, n_rays);
        rc = one_step(c, false);  // parks the unused slots
    }
    uint32_t live = 1;
    for (uint32_t guard = 0; !rc && live; guard++) {
        if (guard > 4096) {
            rc = fail(c, RT_ERR_NOT_READY, "rays still in flight after %u steps", guard * 8);
            break;
        }
        for (int s = 0; s < 8 && !rc; s++) rc = one_step(c, true);
        if (!rc) {
            step(hipMemcpyAsync(&live, &c->glob->g.ray_count, sizeof(live), hipMemcpyDeviceToHost, c->stream), "poll");
            step(hipStreamSynchronize(c->stream), "poll");
        }
    }
    if (!rc && n_rays) step(hipMemcpyAsync(hits, d_hits, sizeof(rt_hit_record) * size_t(n_rays), hipMemcpyDeviceToHost, c->stream), "download hits");
    const uint32_t restore[5] = {c->cfg.warp_width, c->cfg.sample_limit, c->cfg.hit_capacity, c->cfg.row_offset, 0u};
    step(set_global_field(c, offsetof(DevGlobal, warp_width), restore, sizeof(restore)), "restore");
    step(hipStreamSynchronize(c->stream), "trace");
    c->hits = saved_hits;
    dev_free(d_rays);
    dev_free(d_hits);
    return rc;
}

int rt_set_profiling(rt_ctx *c, int enabled)
{
    if (!c) return RT_ERR_INVALID_VALUE;
    const int rc = resolve_timers(c);
    if (rc) return rc;
    c->profiling = enabled != 0;
    c->serial = enabled == RT_PROFILE_SERIAL;
    if (enabled) {
        for (double &ms : c->stage_ms) ms = 0;
        c->timed_steps = 0;
    }
    return RT_OK;
}

int rt_get_stage_times(rt_ctx *c, double ms[RT_STAGE_COUNT], uint64_t *steps)
{
    if (!c || !ms) return RT_ERR_INVALID_VALUE;
    const int rc = resolve_timers(c);
    if (rc) return rc;
    for (int s = 0; s < RT_STAGE_COUNT; s++) ms[s] = c->stage_ms[s];
    if (steps) *steps = c->timed_steps;
    return RT_OK;
}

int rt_debug_stamps(rt_ctx *c, uint64_t dst[8], int reset)
{
    if (!c || !dst) return RT_ERR_INVALID_VALUE;
    unsigned long long sums[kCounterKinds];
    const int rc = read_counters(c, sums, false);
    if (rc) return rc;
    for (int k = 0; k < 8; k++) dst[k] = k < 4 ? sums[4 + k] : 0;
#if defined(RT_STAMPS) && RT_STAMPS == 3
    dst[7] = reinterpret_cast<uint64_t>(c->counters + size_t(kCounterShards) * kCounterKinds);  // device address of the wave trace
#endif
    if (reset) {
        // only the stamp half of every shard is cleared; the event counters keep running
        std::vector<unsigned long long> zero(4, 0);
        for (uint32_t s = 0; s < kCounterShards; s++)
            RT_HIP(c, hipMemcpyAsync(c->counters + size_t(s) * kCounterKinds + 4, zero.data(), 4 * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

void *rt_image_device(rt_ctx *c)
{
    return c ? static_cast<void *>(c->rgba) : nullptr;
}

void *rt_stream(rt_ctx *c)
{
    return c ? static_cast<void *>(c->stream) : nullptr;
}

}  // extern "C"
