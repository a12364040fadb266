// tiled.cpp -- see tiled.h.

#include "tiled.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <iostream>
#include <set>

namespace rt {

void band_of(uint32_t index, uint32_t bands, uint32_t height, uint32_t &first, uint32_t &rows)
{
    const uint32_t base = height / bands, extra = height % bands;
    first = index * base + (index < extra ? index : extra);
    rows = base + (index < extra ? 1 : 0);
}

TiledRenderer::TiledRenderer(size_t width, size_t height) : width_(width), height_(height) {}

TiledRenderer::~TiledRenderer()
{
    for (size_t i = 0; i < band_.size(); i++) {
        if (i < device_.size()) (void)hipSetDevice(device_[i]);
        if (i < comms_.size() && comms_[i]) (void)ncclCommDestroy(static_cast<ncclComm_t>(comms_[i]));
        if (i < band_image_.size() && band_image_[i]) (void)hipFree(band_image_[i]);
        if (i < frame_image_.size() && frame_image_[i]) (void)hipFree(frame_image_[i]);
    }
}

bool TiledRenderer::fail(const char *what, const char *detail)
{
    std::cout << what << ": " << detail << std::endl;
    return false;
}

bool TiledRenderer::init(const std::vector<int> &devices)
{
    const uint32_t n = uint32_t(devices.size());
    if (!n || n > height_) return fail("Cannot tile the frame", "need between 1 and `height` bands");
    int present = 0;
    if (hipGetDeviceCount(&present) != hipSuccess || present <= 0) return fail("Cannot tile the frame", "no GPU");
    const std::set<int> distinct(devices.begin(), devices.end());
    for (int d : distinct)
        if (d < 0 || d >= present) return fail("Cannot tile the frame", "device ordinal out of range");
    // one band per GPU (gathered with RCCL), or every band on ONE GPU (a rehearsal, gathered by copies on it)
    if (distinct.size() != n && distinct.size() != 1) return fail("Cannot tile the frame", "repeat one device ordinal or none");
    if (sample_limit_per_pixel && uint64_t(sample_limit_per_pixel) * width_ * ((height_ + n - 1) / n) > 0x7FFFFFFFull)
        return fail("Cannot tile the frame", "samples per band exceed 2^31 - 1");
    device_ = devices;
    max_rows_ = (height_ + n - 1) / n;
    const size_t band_bytes = max_rows_ * width_ * 4;
    band_image_.assign(n, nullptr);
    frame_image_.assign(n, nullptr);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t first, rows;
        band_of(i, n, uint32_t(height_), first, rows);
        // every GPU keeps a full-size pool, as a single GPU would for the whole frame (main.cpp:516)
        band_.emplace_back(new RayTracer(width_, height_, width_ * height_));
        RayTracer &r = *band_.back();
        r.model_files = model_files;
        r.warp_width = warp_width;
        r.flags = flags;
        r.grid = grid;
        r.band_first = first;
        r.band_rows = rows;
        r.sample_limit = sample_limit_per_pixel ? uint32_t(sample_limit_per_pixel * width_ * rows) : 0;
        if (!r.init(devices[i])) return false;
        if (hipSetDevice(devices[i]) != hipSuccess || hipMalloc(&band_image_[i], band_bytes) != hipSuccess ||
            hipMalloc(&frame_image_[i], band_bytes * n) != hipSuccess || hipMemset(band_image_[i], 0, band_bytes) != hipSuccess)
            return fail("Cannot allocate the band images", hipGetErrorString(hipGetLastError()));
    }
    // RCCL needs one rank per distinct device; bands that share a GPU (a rehearsal) are gathered by copies.
    if (distinct.size() == n) {
        std::vector<ncclComm_t> comms(n);
        const ncclResult_t rc = ncclCommInitAll(comms.data(), int(n), devices.data());
        if (rc != ncclSuccess) return fail("Cannot create the RCCL communicators", ncclGetErrorString(rc));
        comms_.assign(comms.begin(), comms.end());
    }
    image_.assign(width_ * height_ * 4, 0);
    padded_.assign(band_bytes * n, 0);
    return true;
}

bool TiledRenderer::init_frame()
{
    for (auto &b : band_)
        if (!b->init_frame()) return false;
    return true;
}

bool TiledRenderer::make_step()
{
    for (auto &b : band_)  // enqueues only: the GPUs run their steps side by side
        if (!b->make_step()) return false;
    return true;
}

bool TiledRenderer::set_warp_width(uint32_t w)
{
    for (auto &b : band_)
        if (!b->set_warp_width(w)) return false;
    return true;
}

bool TiledRenderer::draw_frame()
{
    const size_t n = band_.size(), band_bytes = max_rows_ * width_ * 4;
    for (size_t i = 0; i < n; i++)  // enqueued only: the GPUs resolve side by side, and the gather is ordered behind each band's resolve
        if (!band_[i]->enqueue_resolve_to_device(band_image_[i])) return false;
    if (!comms_.empty()) {
        // one collective per frame, on the resolved image; every rank receives the whole frame
        ncclResult_t rc = ncclGroupStart();
        for (size_t i = 0; i < n && rc == ncclSuccess; i++) {
            if (hipSetDevice(device_[i]) != hipSuccess) return fail("Cannot select device", hipGetErrorString(hipGetLastError()));
            rc = ncclAllGather(band_image_[i], frame_image_[i], band_bytes, ncclUint8, static_cast<ncclComm_t>(comms_[i]),
                               static_cast<hipStream_t>(band_[i]->stream()));
        }
        const ncclResult_t end = ncclGroupEnd();
        if (rc != ncclSuccess || end != ncclSuccess) return fail("RCCL all-gather failed", ncclGetErrorString(rc != ncclSuccess ? rc : end));
        if (hipSetDevice(device_[0]) != hipSuccess || hipStreamSynchronize(static_cast<hipStream_t>(band_[0]->stream())) != hipSuccess)
            return fail("Cannot finish the gather", hipGetErrorString(hipGetLastError()));
    } else {
        // bands that share a GPU (init() accepts nothing else without RCCL): copies on that GPU
        for (size_t i = 0; i < n; i++) {
            if (!band_[i]->sync()) return false;
            if (hipSetDevice(device_[0]) != hipSuccess ||
                hipMemcpy(static_cast<char *>(frame_image_[0]) + i * band_bytes, band_image_[i], band_bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                return fail("Cannot gather the bands", hipGetErrorString(hipGetLastError()));
        }
    }
    if (hipMemcpy(padded_.data(), frame_image_[0], band_bytes * n, hipMemcpyDeviceToHost) != hipSuccess)
        return fail("Cannot read the frame", hipGetErrorString(hipGetLastError()));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t first, rows;
        band_of(i, uint32_t(n), uint32_t(height_), first, rows);
        std::memcpy(image_.data() + size_t(first) * width_ * 4, padded_.data() + i * band_bytes, size_t(rows) * width_ * 4);
    }
    return true;
}

bool TiledRenderer::counters(rt_counters &sum)
{
    std::memset(&sum, 0, sizeof(sum));
    for (auto &b : band_) {
        rt_counters c;
        if (!b->counters(c)) return false;
        sum.primary_spawned += c.primary_spawned;
        sum.shadow_spawned += c.shadow_spawned;
        sum.box_tests += c.box_tests;
        sum.tri_tests += c.tri_tests;
        sum.steps = c.steps;
    }
    return true;
}

uint64_t TiledRenderer::rays_in_flight()
{
    uint64_t sum = 0;
    for (auto &b : band_) sum += b->rays_in_flight();
    return sum;
}

uint64_t TiledRenderer::current_ray()
{
    uint64_t sum = 0;
    for (auto &b : band_) sum += b->current_ray();
    return sum;
}

bool TiledRenderer::write_ppm(const char *path) const
{
    return write_ppm_file(path, image_.data(), width_, height_);
}

}  // namespace rt
