// rt_sort.hip -- the ray-key sort as an LDS-staged LSD radix sort for gfx950.
//
// Output contract (what sort.cl:19-102 driven by main.cpp:444-463 produces): the first n pairs, read
// tile by tile (8,192 pairs) in the thread-major order of local_count (thread t of a tile owns pairs
// 512*i + t, sort.cl:26-27), stably sorted by the low key bits that cover group_count groups; pairs
// [n, count) unchanged.  The reference gets there with 5-bit digits and three launches per digit, a
// 512-wide scan per bin and a single-work-group global scan.  Here:
//   * 8-bit digits (two passes for up to 65,536 groups instead of three or four);
//   * per pass: tile histograms -> per-digit scan over tiles -> scatter;
//   * the scatter stages its tile in LDS (in logical, i.e. thread-major, order on the first pass),
//     ranks 64 keys at a time with wave ballots (match-any on the digit) against per-wave LDS
//     counters, orders the tile in LDS and writes each digit's run to HBM contiguously.
#include <mutex>

#include "rt_launch.h"

namespace rtd {

constexpr uint32_t kTile = RT_SORT_TILE;   // the reference's sort block: its thread-major order is defined on it
constexpr uint32_t kSortThreads = 512;     // ... by 512 work-items owning 16 pairs each (sort.cl:26-27)
constexpr uint32_t kRadixBits = 8;
constexpr uint32_t kRadix = 1u << kRadixBits;
// Plain-order sorts (free schedule) are free to use smaller tiles: four times as many workgroups, a
// quarter of the LDS each, so that several are resident per CU and the passes stop being latency-bound.
constexpr uint32_t kSmallTile = 2048, kSmallThreads = 256;
#ifndef RT_RANK_THREADS
#define RT_RANK_THREADS 1024
#endif
static_assert(kTile == kSmallTile * kSortCountsPerTile, "pools are whole reference tiles, hence whole small tiles");

// Key bits the reference's pass loop covers: 5 bits per pass while (group_count-1) >> 5p is non-zero,
// at most the 24 id bits (main.cpp:444-445).
uint32_t sort_key_bits(uint32_t group_count)
{
    uint32_t passes = 0;
    for (uint32_t max = group_count - 1; max; max >>= 5) passes++;
    const uint32_t bits = 5 * passes;
    return bits > 24 ? 24 : bits;
}

// Exclusive prefix of v over the block (all threads must call); total = sum over the block.
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= (uint32_t)off) incl += o;
    }
    __syncthreads();  // s_wave may still be in use by a previous call
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < THREADS / 64; k++) {
        const uint32_t t = s_wave[k];
        if (k < w) before += t;
        all += t;
    }
    total = all;
    return before + incl - v;
}

// The same for three values at once (one pair of barriers instead of three); s_wave3: [THREADS / 64][3].
template <int THREADS>
__device__ __forceinline__ void block_exclusive_scan3(const uint32_t (&v)[3], uint32_t (*s_wave3)[3], uint32_t (&before)[3],
                                                      uint32_t (&total)[3])
{
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl[3] = {v[0], v[1], v[2]};
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl[k], off, 64);
            if (lane >= (uint32_t)off) incl[k] += o;
        }
    }
    __syncthreads();  // s_wave3 may still be in use by a previous call
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 3; k++) s_wave3[w][k] = incl[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t sum_before = 0, all = 0;
#pragma unroll
        for (uint32_t j = 0; j < THREADS / 64; j++) {
            const uint32_t t = s_wave3[j][k];
            if (j < w) sum_before += t;
            all += t;
        }
        total[k] = all;
        before[k] = sum_before + incl[k] - v[k];
    }
}

// Per-tile digit histogram of one pass: counts[digit * tiles + tile].
template <uint32_t TILE, uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void k_sort_count(const rt_index_pair *src, const uint32_t *n_ptr,
                                                        uint32_t *counts, uint32_t tiles, uint32_t shift, uint32_t mask)
{
    constexpr uint32_t kPerThread = TILE / THREADS;
    __shared__ uint32_t hist[kRadix];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, base = tile * TILE, n = *n_ptr;
    if (base >= n) return;
    for (uint32_t d = tid; d < kRadix; d += THREADS) hist[d] = 0;
    __syncthreads();
    uint32_t keys[kPerThread];
#pragma unroll
    for (uint32_t i = 0; i < kPerThread; i++) {  // all loads first: they are in flight together
        const uint32_t p = base + i * THREADS + tid;
        keys[i] = p < n ? src[p].key : 0xFFFFFFFFu;
    }
#pragma unroll
    for (uint32_t i = 0; i < kPerThread; i++)
        if (base + i * THREADS + tid < n) atomicAdd(&hist[(keys[i] >> shift) & mask], 1u);
    __syncthreads();
    for (uint32_t d = tid; d < kRadix; d += THREADS) counts[d * tiles + tile] = hist[d];
}

// One block of 256 threads per digit: exclusive scan of that digit's counts over the active tiles, and its total.
__device__ __forceinline__ void scan_digit_row(uint32_t *counts, uint32_t *totals, uint32_t n, uint32_t tiles, uint32_t tile_size,
                                               uint32_t digit, uint32_t *s_wave)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t active = (n + tile_size - 1) / tile_size;
    uint32_t *row = counts + digit * tiles;
    const uint32_t per = (active + 255) / 256;
    const uint32_t first = tid * per < active ? tid * per : active;
    const uint32_t last = first + per < active ? first + per : active;
    uint32_t sum = 0;
    for (uint32_t i = first; i < last; i++) sum += row[i];
    uint32_t total;
    uint32_t run = block_exclusive_scan<256>(sum, s_wave, total);
    for (uint32_t i = first; i < last; i++) {
        const uint32_t t = row[i];
        row[i] = run;
        run += t;
    }
    if (tid == 0) totals[digit] = total;
}

__global__ __launch_bounds__(256) void k_sort_scan(uint32_t *counts, uint32_t *totals, const uint32_t *n_ptr, uint32_t tiles,
                                                   uint32_t tile_size)
{
    __shared__ uint32_t s_wave[4];
    scan_digit_row(counts, totals, *n_ptr, tiles, tile_size, blockIdx.x, s_wave);
}

// Scatter of one pass.  first_pass: read the tile in the reference's thread-major order.
// tail_src (final pass only): where pairs at or beyond n are copied from.
template <uint32_t TILE, uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void k_sort_scatter(const rt_index_pair *src, rt_index_pair *dst,
                                                          const rt_index_pair *tail_src, const uint32_t *n_ptr,
                                                          const uint32_t *counts, const uint32_t *totals,
                                                          uint32_t tiles, uint32_t shift, uint32_t mask,
                                                          uint32_t first_pass)
{
    constexpr uint32_t kTile = TILE, kSortThreads = THREADS, kSortWaves = THREADS / 64, kPerThread = TILE / THREADS;
    // The tile, one padding slot after every 16 pairs: the transposing store (stride 16 pairs per
    // thread) and the linear 8-byte reads are then both free of bank conflicts.
    __shared__ uint2 s_tile[kTile + kTile / 16];
    __shared__ uint32_t s_cnt[kSortWaves * kRadix];  // per wave and digit: running count, later the wave's base
    __shared__ uint32_t s_start[kRadix];             // first position of the digit inside the ordered tile
    __shared__ uint32_t s_gbase[kRadix];             // destination of the digit's run minus s_start
    __shared__ uint32_t s_wave[kSortWaves];

    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t tile = blockIdx.x, base = tile * kTile, n = *n_ptr;
    const uint32_t m = base >= n ? 0 : (n - base < kTile ? n - base : kTile);

    if (tail_src) {
#pragma unroll 4
        for (uint32_t i = 0; i < kPerThread; i++) {
            const uint32_t p = i * kSortThreads + tid;
            if (p >= m) {
                dst[base + p] = tail_src[base + p];
            }
        }
    }
    if (m == 0) return;

    for (uint32_t i = tid; i < kSortWaves * kRadix; i += kSortThreads) s_cnt[i] = 0;

    // Stage the tile.  Thread t of a (possibly ragged) tile owns `mine` pairs 512*i + t; in logical
    // order they are consecutive, after the pairs of lower threads (inverse of sort.cl:72-82).
    {
        const uint32_t each = m / kSortThreads, extra = m % kSortThreads;
        const uint32_t mine = each + (tid < extra ? 1 : 0);
        const uint32_t q0 = tid < extra ? tid * (each + 1) : extra * (each + 1) + (tid - extra) * each;
#pragma unroll 4
        for (uint32_t i = 0; i < kPerThread; i++) {
            if (i >= mine) break;
            const uint32_t p = i * kSortThreads + tid;
            const uint32_t q = first_pass ? q0 + i : p;
            const rt_index_pair e = src[base + p];
            s_tile[q + (q >> 4)] = make_uint2(e.key, e.slot);
        }
    }
    __syncthreads();

    // Rank: wave w owns logical positions [1024 w, 1024 w + 1024), 64 at a time.
    uint32_t keys[kPerThread], vals[kPerThread], ranks[kPerThread];
    const unsigned long long lanes_below = (1ull << lane) - 1;
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; r++) {
        const uint32_t q = w * (kTile / kSortWaves) + r * 64 + lane;
        const bool valid = q < m;
        uint2 e = make_uint2(0, 0);
        if (valid) e = s_tile[q + (q >> 4)];
        const uint32_t d = (e.x >> shift) & mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < kRadixBits; b++) {
            const bool bit = (d >> b) & 1;
            const unsigned long long vote = __ballot(bit);
            peers &= bit ? vote : ~vote;
        }
        const uint32_t before = s_cnt[w * kRadix + d];
        const uint32_t below = __popcll(peers & lanes_below);
        if (valid && below == 0) s_cnt[w * kRadix + d] = before + __popcll(peers);
        keys[r] = e.x;
        vals[r] = e.y;
        ranks[r] = before + below;
    }
    __syncthreads();

    // Per digit: turn the waves' counts into bases, get the tile's count.
    uint32_t digit_count = 0;
    if (tid < kRadix) {
#pragma unroll
        for (uint32_t k = 0; k < kSortWaves; k++) {
            const uint32_t t = s_cnt[k * kRadix + tid];
            s_cnt[k * kRadix + tid] = digit_count;
            digit_count += t;
        }
    }
    uint32_t unused;
    const uint32_t start = block_exclusive_scan<kSortThreads>(digit_count, s_wave, unused);
    const uint32_t digit_base = block_exclusive_scan<kSortThreads>(tid <= mask ? totals[tid] : 0u, s_wave, unused);
    if (tid < kRadix) {
        s_start[tid] = start;
        s_gbase[tid] = digit_base + counts[tid * tiles + tile] - start;
    }
    __syncthreads();

    // Order the tile in LDS, then write each digit's run contiguously.
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; r++) {
        const uint32_t q = w * (kTile / kSortWaves) + r * 64 + lane;
        if (q < m) {
            const uint32_t d = (keys[r] >> shift) & mask;
            s_tile[s_start[d] + s_cnt[w * kRadix + d] + ranks[r]] = make_uint2(keys[r], vals[r]);
        }
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t i = 0; i < kPerThread; i++) {
        const uint32_t j = i * kSortThreads + tid;
        if (j < m) {
            const uint2 e = s_tile[j];
            dst[s_gbase[(e.x >> shift) & mask] + j] = rt_index_pair{e.x, e.y};
        }
    }
}

__global__ void k_copy_pairs(const rt_index_pair *src, rt_index_pair *dst, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[i];
}

template <uint32_t TILE, uint32_t THREADS>
static void sort_passes(hipStream_t stream, const SortWorkspace &ws, const rt_index_pair *in, rt_index_pair *out, uint32_t count,
                        const uint32_t *n_dev, uint32_t bits, uint32_t passes, bool thread_major)
{
    const uint32_t tiles = count / TILE, stride = ws.tiles * (kTile / TILE);  // counts rows are sized for the context's pool
    const uint32_t digit_bits = (bits + passes - 1) / passes;
    const rt_index_pair *src = in;
    for (uint32_t k = 0; k < passes; k++) {
        const uint32_t shift = k * digit_bits;
        const uint32_t width = bits - shift < digit_bits ? bits - shift : digit_bits;
        const uint32_t mask = (1u << width) - 1;
        const bool final_pass = k + 1 == passes;
        rt_index_pair *dst = ((passes - 1 - k) % 2 == 0) ? out : ws.tmp;
        k_sort_count<TILE, THREADS><<<tiles, THREADS, 0, stream>>>(src, n_dev, ws.counts, stride, shift, mask);
        k_sort_scan<<<mask + 1, 256, 0, stream>>>(ws.counts, ws.totals, n_dev, stride, TILE);
        k_sort_scatter<TILE, THREADS><<<tiles, THREADS, 0, stream>>>(src, dst, final_pass ? in : nullptr, n_dev, ws.counts, ws.totals,
                                                                     stride, shift, mask, k == 0 && thread_major ? 1u : 0u);
        src = dst;
    }
}

void launch_sort(hipStream_t stream, const SortWorkspace &ws, const rt_index_pair *in, rt_index_pair *out, uint32_t count,
                 const uint32_t *n_dev, uint32_t group_count, bool plain_order)
{
    // The reference's pass loop covers 5 key bits per pass (sort_key_bits); sorting by exactly those bits
    // is part of its output.  In plain order only the grouping matters: the bits of group_count - 1.
    uint32_t bits = sort_key_bits(group_count);
    if (plain_order)
        for (bits = 0; bits < 24 && ((group_count - 1) >> bits); bits++) {}
    const uint32_t passes = (bits + kRadixBits - 1) / kRadixBits;
    if (passes == 0) {
        k_copy_pairs<<<(count + 255) / 256, 256, 0, stream>>>(in, out, count);
        return;
    }
    if (plain_order) sort_passes<kSmallTile, kSmallThreads>(stream, ws, in, out, count, n_dev, bits, passes, false);
    else sort_passes<kTile, kSortThreads>(stream, ws, in, out, count, n_dev, bits, passes, true);
}

// ---------------------------------------------------------------- grouping + compaction in one sequence (free schedule)
//
// The reference sorts (three launches per pass), then finds the group boundaries in the sorted list,
// scans the groups and moves every pair a second time (count_groups, update_groups, set_ray_index).
// With the order free, only the grouping matters, and no pair has to move twice: one read of the new keys
// counts the groups and gives every pair a rank inside its group (k_fused_rank), the group scan turns the
// counts into places (k_fused_groups), and every pair goes straight to its re-packed place (k_fused_place).
// Three launches instead of eight.  The order of the pairs inside a group depends on the order in which
// workgroups reach the group's counter, i.e. it may differ from run to run; which sample a finished slot
// starts next follows from it, what any sample computes does not.

// What k_fused_place needs to send a pair to its place in the re-packed index.
struct Placement {
    const DevGlobal *glob;
    rt_group_data *grp_data;
    const GroupMove *move;        // per group: first place inside its block of groups, parked count, offsets inside the block
    const uint2 *block_sums;      // per block of kGroupThreads groups: whole-warp and parked rays of the blocks before
    const uint32_t *block_main;   // ... and their new pairs
    uint32_t count, group_blocks, tile_blocks, groups;
};

// Rank of every one of the first n pairs inside its group, one tile of 8,192 pairs per workgroup of 1,024 threads
// (the dense per-group counters in LDS are cleared and walked by four times as many threads as with 256, and
// every workgroup is still resident at once): the tile's pairs are
// counted per group in LDS, handing each its rank among the tile's pairs of that group; one atomic per
// non-empty group then reserves the tile's range in the group and returns where it starts.  s_group: [groups].
constexpr uint32_t kRankThreads = RT_RANK_THREADS, kRankTile = 8 * kRankThreads;  // pairs per workgroup of the rank kernel
__global__ __launch_bounds__(kRankThreads) void k_fused_rank(const rt_index_pair *src, DevGlobal *glob, uint32_t *ghist,
                                                              uint32_t *rank, uint32_t groups)
{
    extern __shared__ uint32_t s_group[];
    constexpr uint32_t kPerThread = kRankTile / kRankThreads;
    const uint32_t tid = threadIdx.x, lane = tid & 63, base = blockIdx.x * kRankTile, n = glob->g.ray_count;
    if (blockIdx.x == 0 && tid == 0) glob->sort_count = n;  // the later kernels run after the live count has been replaced
    if (base >= n) return;
    uint32_t keys[kPerThread], local[kPerThread];
#pragma unroll
    for (uint32_t i = 0; i < kPerThread; i++) {
        const uint32_t p = base + i * kRankThreads + tid;
        // (read whether or not it counts -- the tile lies inside the index -- so that the eight requests go out together:
        // a load behind a branch gets a wait of its own)
        const uint32_t key = src[p].key;
        keys[i] = p < n ? key : 0xFFFFFFFFu;
    }
    for (uint32_t i = tid; i < groups; i += kRankThreads) s_group[i] = 0;
    __syncthreads();
    const unsigned long long lanes_below = (1ull << lane) - 1;
#pragma unroll
    for (uint32_t i = 0; i < kPerThread; i++) {
        bool todo = base + i * kRankThreads + tid < n;
        uint32_t id = keys[i] & kIdMask;
        id = id < groups ? id : groups - 1;
        keys[i] = id;
        local[i] = 0;
        // Neighbouring rays mostly move on to the same few nodes; 64 LDS atomics on one word would run one
        // after the other.  The two most likely values take one atomic per wave each, the rest one per lane.
#pragma unroll
        for (int round = 0; round < 2; round++) {
            const unsigned long long left = __ballot(todo);
            if (!left) break;
            const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)id, (int)(__ffsll((long long)left) - 1));
            const unsigned long long same = __ballot(todo && id == first);
            const uint32_t leader = (uint32_t)(__ffsll((long long)same) - 1);
            uint32_t before = 0;
            if (todo && id == first && lane == leader) before = atomicAdd(&s_group[first], (uint32_t)__popcll(same));
            before = (uint32_t)__builtin_amdgcn_readlane((int)before, (int)leader);
            if (todo && id == first) {
                todo = false;
                local[i] = before + (uint32_t)__popcll(same & lanes_below);
            }
        }
        if (todo) local[i] = atomicAdd(&s_group[id], 1u);
    }
    __syncthreads();
    // (eight reservations in flight per thread: one after the other they would take a memory round trip each)
    constexpr uint32_t kBatch = 8;
    for (uint32_t first = tid; first < groups; first += kBatch * kRankThreads) {
        uint32_t c[kBatch], start[kBatch];
#pragma unroll
        for (uint32_t k = 0; k < kBatch; k++) {
            const uint32_t g = first + k * kRankThreads;
            c[k] = g < groups ? s_group[g] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; k++) {
            start[k] = 0;
            if (c[k]) start[k] = atomicAdd(&ghist[first + k * kRankThreads], c[k]);
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; k++)
            if (c[k]) s_group[first + k * kRankThreads] = start[k];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kPerThread; i++) {
        const uint32_t p = base + i * kRankThreads + tid;
        if (p < n) rank[p] = local[i] + s_group[keys[i]];
    }
}

// update_groups (ray-tracer.cl:255-290) from the histogram: one thread per group, block-wide prefix sums,
// the last block to finish scans the block totals and updates the frame counters.
__global__ __launch_bounds__(kGroupThreads) void k_fused_groups(DevGlobal *glob, rt_group_data *grp_data, uint32_t *ghist,
                                                                 CompactWorkspace ws, uint32_t *block_main, uint32_t group_blocks)
{
    __shared__ uint32_t s_wave3[kGroupThreads / 64][3];
    __shared__ bool s_last;
    const uint32_t tid = threadIdx.x;
    const uint32_t g = blockIdx.x * kGroupThreads + tid;
    const uint32_t groups = step_constant(&glob->g.group_count), n = step_constant(&glob->sort_count);
    const uint32_t warp_width = step_constant(&glob->warp_width);
    uint32_t main_count = 0, aligned = 0, tail = 0;
    rt_group_data grp = rt_group_data{{0, 0}, {0, 0}, {0, 0}};
    if (g < groups) {
        main_count = ghist[g];
        grp = grp_data[g];
        ghist[g] = 0;  // ready for the next step (stored after both loads have been asked for)
        const uint32_t total = main_count + grp.count[1];
        tail = g != groups - 1 ? total % warp_width : total;  // the last group is the dead-ray bin
        aligned = total - tail;
    }
    const uint32_t mine[3] = {main_count, aligned, tail};
    uint32_t before[3], totals[3];
    block_exclusive_scan3<kGroupThreads>(mine, s_wave3, before, totals);
    const uint32_t before_main = before[0], before_aligned = before[1], before_tail = before[2];
    const uint32_t sum_main = totals[0], sum_aligned = totals[1], sum_tail = totals[2];
    if (g < groups) {
        ws.move[g] = GroupMove{before_main, grp.offset[1] - main_count, tail, before_aligned, before_tail, {0, 0, 0}};
        grp_data[g].base[1] = grp.offset[1] - main_count;
        *reinterpret_cast<uint2 *>(grp_data[g].count) = make_uint2(aligned, tail);
    }
    if (g == 0) ws.block_sums[group_blocks] = make_uint2(aligned, 0);  // whole warps of group 0: the next step's new pixels

    // The last block to get here owns the block totals.
    if (tid == 0) {
        ws.block_sums[blockIdx.x] = make_uint2(sum_aligned, sum_tail);
        block_main[blockIdx.x] = sum_main;
        __threadfence();
        s_last = atomicAdd(ws.ticket, 1u) == group_blocks - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    uint32_t carry[3] = {0, 0, 0};
    for (uint32_t first = 0; first < group_blocks; first += kGroupThreads) {
        const bool in = first + tid < group_blocks;
        const uint2 sums = in ? ws.block_sums[first + tid] : make_uint2(0, 0);
        const uint32_t v[3] = {in ? block_main[first + tid] : 0u, sums.x, sums.y};
        uint32_t excl[3], all[3];
        block_exclusive_scan3<kGroupThreads>(v, s_wave3, excl, all);
        if (in) {
            block_main[first + tid] = carry[0] + excl[0];
            ws.block_sums[first + tid] = make_uint2(carry[1] + excl[1], carry[2] + excl[2]);
        }
        for (int k = 0; k < 3; k++) carry[k] += all[k];
    }
    if (tid == 0) {
        glob->g.pixel_offset += glob->g.pixel_count;
        glob->g.pixel_count = ws.block_sums[group_blocks].x;
        glob->g.old_count = n;
        glob->g.ray_count = carry[1];  // all whole warps; the parked rays follow them
        glob->spill_count = 0;
        *ws.ticket = 0;
    }
}

// set_ray_index (ray-tracer.cl:293-302): every pair to its place in the re-packed index -- the whole warps of
// all groups first, group by group, then the parked remainders.  A pair of the first n goes by its rank among
// its group's new pairs, a pair parked earlier by its place among the parked ones, which follow the new ones.
// The first workgroups also finish the group table (ray-tracer.cl:262-289).
__global__ __launch_bounds__(kGroupThreads) void k_fused_place(const rt_index_pair *src, const uint32_t *rank, rt_index_pair *dst,
                                                                Placement pl)
{
    const uint32_t old_count = pl.glob->g.old_count, all_aligned = pl.glob->g.ray_count;
    if (blockIdx.x < pl.group_blocks) {
        const uint32_t g = blockIdx.x * kGroupThreads + threadIdx.x;
        if (g < pl.glob->g.group_count) {
            const GroupMove m = pl.move[g];
            const uint2 block = pl.block_sums[blockIdx.x];
            pl.grp_data[g].base[0] = pl.block_main[blockIdx.x] + m.base0;
            *reinterpret_cast<uint2 *>(pl.grp_data[g].offset) = make_uint2(block.x + m.off0, all_aligned + block.y + m.off1);
        }
    }
    if (blockIdx.x >= pl.tile_blocks) return;
    constexpr uint32_t kPerThread = kSmallTile / kGroupThreads;
    const uint32_t base = blockIdx.x * kSmallTile;
    rt_index_pair e[kPerThread];
    uint32_t r[kPerThread];
#pragma unroll
    for (uint32_t i = 0; i < kPerThread; i++) {
        const uint32_t p = base + i * kGroupThreads + threadIdx.x;
        e[i] = src[p];
        r[i] = p < old_count ? rank[p] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < kPerThread; i++) {
        const uint32_t p = base + i * kGroupThreads + threadIdx.x;
        uint32_t g = e[i].key & kIdMask;
        if (p < old_count && g >= pl.groups) g = pl.groups - 1;  // (as counted)
        const GroupMove m = pl.move[g];
        const uint2 block = pl.block_sums[g / kGroupThreads];
        const uint32_t index = p < old_count ? r[i] : p - m.base1;  // place among the group's pairs, new ones first
        const uint32_t to = index < m.count1 ? index + all_aligned + block.y + m.off1 : index + block.x + m.off0 - m.count1;
        if (to < pl.count) dst[to] = e[i];  // always true for consistent group data; keeps bad caller data from faulting
    }
}

uint32_t fused_hist_slices() { return 1; }

uint32_t fused_hist_lds_bytes(uint32_t group_count)
{
    return group_count * sizeof(uint32_t);
}

hipError_t fused_setup(uint32_t group_count)
{
    const uint32_t bytes = fused_hist_lds_bytes(group_count);
    if (bytes <= 48 * 1024) return hipSuccess;
    int device = 0, limit = 0;
    hipError_t err = hipGetDevice(&device);
    if (err == hipSuccess) err = hipDeviceGetAttribute(&limit, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
    if (err != hipSuccess) return err;
    if (bytes > uint32_t(limit)) return hipErrorInvalidValue;  // the caller takes the sort path
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_rank), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void launch_fused(hipStream_t stream, const SortWorkspace &ws, const FusedWorkspace &fw, const CompactWorkspace &cw, DevGlobal *glob,
                  rt_group_data *grp_data, const rt_index_pair *in, rt_index_pair *out, uint32_t count, uint32_t group_count)
{
    const uint32_t tiles = count / kSmallTile;
    const uint32_t group_blocks = compact_group_blocks(group_count);
    uint32_t *rank = reinterpret_cast<uint32_t *>(ws.tmp);  // one word per pair in the sort's ping-pong buffer
    const Placement pl{glob, grp_data, cw.move, cw.block_sums, fw.block_main, count, group_blocks, tiles, group_count};
    const uint32_t place_grid = tiles > group_blocks ? tiles : group_blocks;
    k_fused_rank<<<count / kRankTile, kRankThreads, fused_hist_lds_bytes(group_count), stream>>>(in, glob, fw.ghist, rank, group_count);
    k_fused_groups<<<group_blocks, kGroupThreads, 0, stream>>>(glob, grp_data, fw.ghist, cw, fw.block_main, group_blocks);
    k_fused_place<<<place_grid, kGroupThreads, 0, stream>>>(in, rank, out, pl);
}

}  // namespace rtd

// ---------------------------------------------------------------- the sort without any allocation
//
// rt_sort_pairs_static: the same kernels, with the pass workspace (live count, per-digit totals, per-tile
// digit counts, ping-pong buffer) in a file-scope device array instead of buffers a context allocated.
// Every GPU holds its own copy of the array at its own address; that address is looked up once per
// device and only the looked-up pointer is ever handed to a kernel or a copy.

namespace {

constexpr uint32_t kStaticMaxPairs = RT_SORT_STATIC_MAX_PAIRS;
constexpr uint32_t kStaticMaxTiles = kStaticMaxPairs / rtd::kTile;
static_assert(kStaticMaxPairs % rtd::kTile == 0, "the static limit is a whole number of tiles");

struct StaticSortScratch {
    uint32_t n;                                        // live count of the call in flight
    uint32_t pad[63];
    uint32_t totals[rtd::kRadix];
    uint32_t counts[rtd::kRadix * kStaticMaxTiles];
    rt_index_pair tmp[kStaticMaxPairs];
};

// 8.1 MB per GPU, held until the process ends.  In host code this name is only a handle for
// hipGetSymbolAddress; it is not a device address.
static __device__ __align__(256) unsigned char scratch[sizeof(StaticSortScratch)];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_lock;
StaticSortScratch *g_scratch_ptr[kMaxDevices];  // device address of `scratch` on each GPU, once looked up

__global__ void k_store_u32(uint32_t *dst, uint32_t value)
{
    *dst = value;
}

}  // namespace

namespace rtd {

// Looks up, once per device, where `scratch` lives on the CURRENT device.  Called by rt_create for the
// context's device and by rt_sort_pairs_static on first use.  Allocates nothing.
hipError_t sort_static_setup(void **out)
{
    int device = -1;
    hipError_t err = hipGetDevice(&device);
    if (err != hipSuccess) return err;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(g_scratch_lock);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        err = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (err != hipSuccess) return err;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = static_cast<StaticSortScratch *>(ptr);
    }
    if (out) *out = g_scratch_ptr[device];
    return hipSuccess;
}

}  // namespace rtd

extern "C" int rt_sort_pairs_static(const rt_index_pair *in_dev, rt_index_pair *out_dev, uint32_t count, uint32_t n,
                                    uint32_t group_count, void *stream)
{
    if (!in_dev || !out_dev || !count || count % RT_SORT_TILE) return RT_ERR_INVALID_VALUE;
    if (count > kStaticMaxPairs) return RT_ERR_UNSUPPORTED;  // beyond the static workspace: use rt_sort_pairs
    if (n > count || !group_count || group_count > RT_GROUP_ID_MASK + 1) return RT_ERR_INVALID_VALUE;
    const uintptr_t a = reinterpret_cast<uintptr_t>(in_dev), b = reinterpret_cast<uintptr_t>(out_dev);
    const uintptr_t bytes = uintptr_t(count) * sizeof(rt_index_pair);
    if (a < b + bytes && b < a + bytes) return RT_ERR_INVALID_VALUE;  // in and out overlap
    void *where = nullptr;
    if (rtd::sort_static_setup(&where) != hipSuccess) return RT_ERR_HIP;
    StaticSortScratch *ws = static_cast<StaticSortScratch *>(where);
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();  // an error left behind by an earlier, unrelated call is not this call's
    // Nothing in the workspace has to start at zero: every pass overwrites the counts of the tiles it
    // reads and all totals before using them, and tmp is written before it is read.
    k_store_u32<<<1, 1, 0, s>>>(&ws->n, n);
    const rtd::SortWorkspace sw{ws->counts, ws->totals, ws->tmp, count / RT_SORT_TILE};
    rtd::launch_sort(s, sw, in_dev, out_dev, count, &ws->n, group_count);
    return hipGetLastError() == hipSuccess ? RT_OK : RT_ERR_LAUNCH;
}

