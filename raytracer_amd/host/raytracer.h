// raytracer.h -- the host-side renderer object, successor of the reference's class RayTracer
// (main.cpp:50-207), driving the gfx950 layer through the C ABI of rt_hip.h instead of OpenCL.
// Same life cycle and method names: init(), init_frame(), make_step(), draw_frame(), current_ray();
// same error convention: methods return bool and print "<what>: <status>" to std::cout.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "rt_hip.h"
#include "scene.h"

namespace rt {

class RayTracer {
public:
    // width x height image, ray_count slots (rounded up to a multiple of 8,192 like main.cpp:188).
    RayTracer(size_t width, size_t height, size_t ray_count);
    ~RayTracer();
    RayTracer(const RayTracer &) = delete;
    RayTracer &operator=(const RayTracer &) = delete;

    // What the benchmark varies; defaults give the reference scene (main.cpp:268-343).
    std::vector<std::string> model_files = {"bun_zipper.ply", "dragon_vrip.ply"};
    uint32_t n_obj = 256, grid = 0;
    uint32_t warp_width = 64;            // queue-compaction granularity (the reference: 32)
    uint32_t band_first = 0, band_rows = 0;  // row band of this GPU; 0 rows = whole frame
    uint32_t sample_limit = 0;
    uint32_t flags = 0;                  // rt_config.flags: RT_FLAG_FREE_SCHEDULE, RT_FLAG_SHADING(...)

    // Builds the scene on the host, creates the device context and uploads (main.cpp:192-195).
    bool init(int device);
    bool init_frame();
    bool make_step();
    // Resolves the accumulator to RGBA8 (main.cpp:479-490); the image is kept in image().
    bool draw_frame();
    // The same resolve into a device buffer of width * height() * 4 bytes on this context's GPU; no host copy.
    bool resolve_to_device(void *device_image);
    bool enqueue_resolve_to_device(void *device_image);  // the same, not waited for
    bool sync();                                          // waits until the context's stream is idle
    void *stream();                      // the context's hipStream_t
    uint32_t rays_in_flight();           // GlobalData.ray_count
    uint32_t current_ray();
    bool counters(rt_counters &out);
    bool set_warp_width(uint32_t w);

    const std::vector<uint8_t> &image() const { return image_; }
    size_t width() const { return width_; }
    size_t height() const { return band_rows ? band_rows : height_; }
    size_t pool_size() const { return ray_count_; }
    uint32_t group_count() const { return scene_.data.group_count; }
    const Scene &scene() const { return scene_; }
    // Writes image() as a binary PPM, first row at the bottom as the reference's GL quad shows it.
    bool write_ppm(const char *path) const;

private:
    bool report(const char *what, int status);

    size_t width_, height_, ray_count_;
    rt_ctx *ctx_ = nullptr;
    Scene scene_;
    std::vector<uint8_t> image_;
};

// RGBA8 rows (first row first) as a binary PPM, first row at the bottom as the reference's GL quad shows it.
bool write_ppm_file(const char *path, const uint8_t *rgba, size_t width, size_t height);

}  // namespace rt
