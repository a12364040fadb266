// rt_launch.h -- host-callable launchers of the kernels in rt_kernels.hip and rt_sort.hip.
#pragma once

#include "rt_device.h"

namespace rtd {

struct LaunchCtx {
    hipStream_t stream;
    Scene scene;
    Frame frame;
    rt_camera cam;
    rt_group_data *grp_data;
    uint32_t *spill_list;  // pool entries: index positions of the box visits left to the one-ray-per-wave kernel
    CompactWorkspace compact;
    uint32_t pool, area_size, group_count;
    bool free_schedule;  // RT_FLAG_FREE_SCHEDULE: shading and spawn inside the visiting kernels, no shading launch
    bool long_lists;     // the scene has a box list of more than 256 boxes: the box kernel with two-byte places
    bool mad;            // RT_FLAG_ARITH_MAD: the kernels that fuse their multiply-adds
};

void launch_init_frame(const LaunchCtx &c);
void launch_load_rays(const LaunchCtx &c, const rt_ray *rays_dev, uint32_t n_rays);
// The three process kernels touch disjoint rays, so they may run on different streams.
void launch_process_boxes(const LaunchCtx &c);
void launch_process_leaves(const LaunchCtx &c, hipStream_t stream);
void launch_process_shade(const LaunchCtx &c);
void launch_compact(const LaunchCtx &c, const rt_index_pair *sorted, rt_index_pair *dst);
void launch_compact_raw(hipStream_t stream, DevGlobal *glob, rt_group_data *grp_data, const CompactWorkspace &ws,
                        const rt_index_pair *sorted, rt_index_pair *dst, uint32_t count, uint32_t group_count);
uint32_t compact_group_blocks(uint32_t group_count);  // block_sums holds this many entries + 1
void launch_resolve(const LaunchCtx &c, void *out_dev);
void launch_export_rays(const LaunchCtx &c, rt_ray_queue *out_dev);

// Radix sort of (key, slot) pairs, see rt_sort.hip.
constexpr uint32_t kSortCountsPerTile = 4;  // plain-order sorts cut every tile of RT_SORT_TILE pairs into four
struct SortWorkspace {
    uint32_t *counts;   // [256][tiles * kSortCountsPerTile] per-tile digit counts, scanned in place (the static
                        // workspace of rt_sort_pairs_static, which only sorts in the reference's order, holds [256][tiles])
    uint32_t *totals;   // [256] per-digit totals of a pass
    rt_index_pair *tmp; // one pool-sized ping-pong buffer
    uint32_t tiles;     // capacity in tiles of RT_SORT_TILE pairs
};
// Sorts the first *n_dev of `count` pairs of `in` into `out`; pairs beyond are copied through.
// plain_order: equal keys keep the order they have in `in`; otherwise the order of the reference's sort
// (every tile of RT_SORT_TILE pairs read thread-major, sort.cl:26-27).
void launch_sort(hipStream_t stream, const SortWorkspace &ws, const rt_index_pair *in, rt_index_pair *out,
                 uint32_t count, const uint32_t *n_dev, uint32_t group_count, bool plain_order = false);
uint32_t sort_key_bits(uint32_t group_count);

// Grouping + queue compaction of the free schedule as one sequence (rt_sort.hip): per-group histogram that also
// ranks every pair inside its group, group scan, one placement of every pair in the re-packed index; ws.tmp holds
// the ranks.  For at most kFusedMaxGroups groups.
constexpr uint32_t kFusedMaxGroups = 32768;  // the histogram of a workgroup lives in LDS
struct FusedWorkspace {
    uint32_t *ghist;       // [group_count] new pairs per group, zero between steps
    uint32_t *block_main;  // [blocks of kGroupThreads groups] new pairs per block, then their exclusive prefix sums
};
uint32_t fused_hist_slices();
hipError_t fused_setup(uint32_t group_count);  // once per scene, before the first launch_fused
void launch_fused(hipStream_t stream, const SortWorkspace &ws, const FusedWorkspace &fw, const CompactWorkspace &cw, DevGlobal *glob,
                  rt_group_data *grp_data, const rt_index_pair *in, rt_index_pair *out, uint32_t count, uint32_t group_count);
// Looks up the workspace of rt_sort_pairs_static on the current device (once per device; allocates nothing).
hipError_t sort_static_setup(void **out);

}  // namespace rtd
