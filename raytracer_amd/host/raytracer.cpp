// raytracer.cpp -- see raytracer.h.

#include "raytracer.h"

#include <cstdio>
#include <iostream>

namespace rt {

RayTracer::RayTracer(size_t width, size_t height, size_t ray_count)
    : width_(width), height_(height), ray_count_(align_up(ray_count, RT_SORT_TILE))
{
}

RayTracer::~RayTracer()
{
    if (ctx_) rt_destroy(ctx_);
}

bool RayTracer::report(const char *what, int status)
{
    if (status == RT_OK) return true;
    std::cout << what << ": " << rt_status_string(status);
    const char *detail = rt_last_error(ctx_);
    if (detail && *detail) std::cout << " (" << detail << ")";
    std::cout << std::endl;
    return false;
}

bool RayTracer::init(int device)
{
    SceneSpec spec;
    spec.model_files = model_files;
    spec.n_obj = n_obj;
    spec.grid = grid;
    spec.width = uint32_t(width_);
    spec.height = uint32_t(height_);
    spec.ray_count = uint32_t(ray_count_);
    spec.band_first = band_first;
    spec.band_rows = band_rows;
    std::string error;
    std::cout << "Building scene..." << std::endl;
    if (!build_scene(spec, scene_, error)) {
        std::cout << "Scene build failed: " << error << std::endl;
        return false;
    }
    std::cout << "Group count: " << scene_.data.group_count << std::endl;

    rt_config cfg = {};
    cfg.device = device;
    cfg.width = scene_.data.cam.width;
    cfg.height = scene_.data.cam.height;
    cfg.ray_count = scene_.pool_size;
    cfg.warp_width = warp_width;
    cfg.sample_limit = sample_limit;
    cfg.row_offset = scene_.row_offset;
    cfg.flags = flags;
    if (!report("Cannot create context", rt_create(&cfg, &ctx_))) return false;
    image_.assign(size_t(cfg.width) * cfg.height * 4, 0);
    return report("Cannot upload scene",
                  rt_upload_scene(ctx_, &scene_.data, scene_.groups.data(), scene_.groups.size(), scene_.matrices.data(),
                                  scene_.matrices.size(), scene_.aabbs.data(), scene_.aabbs.size(), scene_.vertices.data(),
                                  scene_.vertices.size(), scene_.triangles.data(), scene_.triangles.size()));
}

bool RayTracer::init_frame()
{
    return report("Cannot initialise frame", rt_init_frame(ctx_));
}

bool RayTracer::make_step()
{
    return report("Cannot execute step", rt_step(ctx_, 1));
}

bool RayTracer::draw_frame()
{
    return report("Cannot resolve image", rt_resolve_rgba8(ctx_, image_.data(), 0));
}

bool RayTracer::resolve_to_device(void *device_image)
{
    return report("Cannot resolve image", rt_resolve_rgba8(ctx_, device_image, 1));
}

bool RayTracer::enqueue_resolve_to_device(void *device_image)
{
    return report("Cannot resolve image", rt_resolve_rgba8_enqueue(ctx_, device_image));
}

bool RayTracer::sync()
{
    return report("Cannot finish", rt_sync(ctx_));
}

void *RayTracer::stream()
{
    return rt_stream(ctx_);
}

uint32_t RayTracer::rays_in_flight()
{
    rt_global_data g = {};
    report("Cannot read buffer data", rt_get_global(ctx_, &g));
    return g.ray_count;
}

uint32_t RayTracer::current_ray()
{
    uint32_t value = 0;
    report("Cannot read buffer data", rt_current_ray(ctx_, &value));
    return value;
}

bool RayTracer::counters(rt_counters &out)
{
    return report("Cannot read counters", rt_get_counters(ctx_, &out));
}

bool RayTracer::set_warp_width(uint32_t w)
{
    return report("Cannot set granularity", rt_set_warp_width(ctx_, w));
}

bool RayTracer::write_ppm(const char *path) const
{
    return write_ppm_file(path, image_.data(), width_, height());
}

bool write_ppm_file(const char *path, const uint8_t *rgba, size_t w, size_t h)
{
    FILE *out = std::fopen(path, "wb");
    if (!out) return false;
    std::fprintf(out, "P6\n%zu %zu\n255\n", w, h);
    std::vector<uint8_t> row(w * 3);
    for (size_t y = h; y-- > 0;) {
        const uint8_t *src = rgba + y * w * 4;
        for (size_t x = 0; x < w; x++) {
            row[3 * x + 0] = src[4 * x + 0];
            row[3 * x + 1] = src[4 * x + 1];
            row[3 * x + 2] = src[4 * x + 2];
        }
        std::fwrite(row.data(), 1, row.size(), out);
    }
    return std::fclose(out) == 0;
}

}  // namespace rt
