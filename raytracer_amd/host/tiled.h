// tiled.h -- one frame on several GPUs of a node: screen-tile data parallelism under the C++ host.
//
// The reference renders on one device (one cl_command_queue, main.cpp:240).  Here the frame is cut
// into row bands, one rt::RayTracer (one context, one GPU) per band, the scene replicated; a band is the
// full camera with fewer rows and a row offset, so a pixel gets the same samples whichever GPU owns it.
// Steps are only enqueued (rt_step is asynchronous), so all GPUs work at once; once per frame the
// resolved RGBA8 bands are gathered: with one context per distinct device by an RCCL all-gather on the
// contexts' streams (over xGMI), otherwise -- several bands rehearsed on one GPU -- by device copies.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "raytracer.h"

namespace rt {

// (first row, row count) of band `index` of `bands`: rows dealt as evenly as possible, earlier bands first.
void band_of(uint32_t index, uint32_t bands, uint32_t height, uint32_t &first, uint32_t &rows);

class TiledRenderer {
public:
    TiledRenderer(size_t width, size_t height);
    ~TiledRenderer();
    TiledRenderer(const TiledRenderer &) = delete;
    TiledRenderer &operator=(const TiledRenderer &) = delete;

    std::vector<std::string> model_files;
    uint32_t warp_width = 64, sample_limit_per_pixel = 0, flags = 0, grid = 0;

    // One band per entry of `devices` (HIP ordinals): all distinct (one band per GPU, gathered with RCCL) or all the
    // same (a rehearsal of the bands on one GPU, gathered by copies on it).
    bool init(const std::vector<int> &devices);
    bool init_frame();
    bool make_step();
    bool set_warp_width(uint32_t w);
    // Resolves every band, gathers them and copies the whole frame to image().
    bool draw_frame();
    bool counters(rt_counters &sum);
    uint64_t rays_in_flight();
    uint64_t current_ray();  // samples started, summed over the bands

    const std::vector<uint8_t> &image() const { return image_; }
    bool uses_rccl() const { return !comms_.empty(); }
    size_t bands() const { return band_.size(); }
    bool write_ppm(const char *path) const;

private:
    bool fail(const char *what, const char *detail);

    size_t width_, height_;
    std::vector<std::unique_ptr<RayTracer>> band_;
    std::vector<int> device_;
    std::vector<void *> band_image_, frame_image_;  // per band, on its device: padded band, all padded bands
    std::vector<void *> comms_;                     // ncclComm_t per band when every band has a device of its own
    size_t max_rows_ = 0;
    std::vector<uint8_t> image_, padded_;
};

}  // namespace rt
