// main.cpp -- headless frame loop: the reference's ray_tracer()/main() (main.cpp:494-586) without the
// SDL window.  Each "frame" is what one mouse click triggered there: 32 steps, a resolve, and the
// line "Frame ready in X s, N rays, Y MR/s." (main.cpp:534-540).  The image goes to a PPM file.
//
//   rt_main [--device N] [--width W] [--height H] [--rays R] [--frames F] [--steps S]
//           [--models a.ply;b.ply] [--warp W] [--out image.ppm]

#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <string>

#include "raytracer.h"

namespace {

double now_seconds()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}

}  // namespace

int main(int argc, char **argv)
{
    int device = 0, frames = 4, steps = 32;  // repeat_count = 32, main.cpp:515
    size_t width = 1024, height = 1024, rays = 0;
    uint32_t warp = 64;
    std::string models = "assets/bun_zipper.ply;assets/bun_zipper.ply", out = "frame.ppm";
    for (int i = 1; i < argc; i++) {
        const std::string arg = argv[i];
        const char *value = i + 1 < argc ? argv[i + 1] : nullptr;
        auto need = [&]() {
            if (!value) {
                std::cout << "Missing value for " << arg << std::endl;
                std::exit(2);
            }
            i++;
            return value;
        };
        if (arg == "--device") device = std::atoi(need());
        else if (arg == "--width") width = std::strtoul(need(), nullptr, 10);
        else if (arg == "--height") height = std::strtoul(need(), nullptr, 10);
        else if (arg == "--rays") rays = std::strtoul(need(), nullptr, 10);
        else if (arg == "--frames") frames = std::atoi(need());
        else if (arg == "--steps") steps = std::atoi(need());
        else if (arg == "--models") models = need();
        else if (arg == "--warp") warp = uint32_t(std::atoi(need()));
        else if (arg == "--out") out = need();
        else {
            std::cout << "Usage: rt_main [--device N] [--width W] [--height H] [--rays R] [--frames F] [--steps S]"
                         " [--models a.ply;b.ply] [--warp W] [--out image.ppm]" << std::endl;
            return arg == "--help" ? 0 : 2;
        }
    }
    if (!rays) rays = width * height;  // main.cpp:516

    rt::RayTracer tracer(width, height, rays);
    tracer.model_files.clear();
    for (size_t at = 0; at <= models.size();) {
        size_t end = models.find(';', at);
        if (end == std::string::npos) end = models.size();
        if (end > at) tracer.model_files.push_back(models.substr(at, end - at));
        at = end + 1;
    }
    tracer.warp_width = warp;
    if (!tracer.init(device)) return 1;
    std::cout << "Ready." << std::endl;
    if (!tracer.init_frame() || !tracer.draw_frame()) return 1;

    uint32_t cur_ray = tracer.current_ray();
    rt_counters before = {};
    tracer.counters(before);
    std::cout << std::setprecision(3) << std::fixed;
    for (int frame = 0; frame < frames; frame++) {
        const double start = now_seconds();
        const uint32_t old_ray = cur_ray;
        for (int i = 0; i < steps; i++)
            if (!tracer.make_step()) return 1;
        if (!tracer.draw_frame()) return 1;
        const double delta = now_seconds() - start;
        cur_ray = tracer.current_ray();
        rt_counters after = {};
        tracer.counters(after);
        const double all = double(after.primary_spawned - before.primary_spawned) + double(after.shadow_spawned - before.shadow_spawned);
        before = after;
        std::cout << "Frame ready in " << delta << " s, " << (cur_ray - old_ray) << " rays, " << 1e-6 * (cur_ray - old_ray) / delta
                  << " MR/s (" << 1e-6 * all / delta << " MR/s with shadow rays)." << std::endl;
    }
    if (!tracer.write_ppm(out.c_str())) {
        std::cout << "Cannot write " << out << std::endl;
        return 1;
    }
    std::cout << "Image written to " << out << std::endl;
    return 0;
}
