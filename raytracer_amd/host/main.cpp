// main.cpp -- headless frame loop: the reference's ray_tracer()/main() (main.cpp:494-586) without the
// SDL window.  Each "frame" is what one mouse click triggered there: 32 steps, a resolve, and the
// line "Frame ready in X s, N rays, Y MR/s." (main.cpp:534-540).  The image goes to a PPM file.
//
//   rt_main [--device N | --gpus N | --devices a,b,...] [--rccl] [--width W] [--height H] [--rays R]
//           [--frames F] [--steps S] [--spp K] [--models a.ply;b.ply] [--warp W] [--schedule free|reference] [--arith strict|mad]
//           [--band-first ROW --band-rows ROWS] [--out image.ppm]
//
// --gpus N renders on devices 0..N-1, one row band each, and gathers the bands with RCCL once per frame
// (--devices names the ordinals; a repeated ordinal rehearses several bands on one GPU, gathered by device
// copies).  --band-first/--band-rows render one band alone, as one rank of a multi-GPU run would.  --spp K
// renders exactly K samples per pixel and runs until no ray is in flight, instead of --frames x --steps.

#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <iomanip>
#include <iostream>
#include <string>

#include "tiled.h"

namespace {

double now_seconds()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}

const char *kUsage =
    "Usage: rt_main [--device N | --gpus N | --devices a,b,...] [--rccl] [--width W] [--height H] [--rays R] [--frames F]"
    " [--steps S] [--spp K] [--models a.ply;b.ply] [--warp W] [--schedule free|reference] [--arith strict|mad]"
    " [--band-first ROW --band-rows ROWS] [--out image.ppm]";

// What the frame loop needs from one renderer or from a tiled set of them.
struct Loop {
    std::function<bool()> init_frame, make_step, draw_frame;
    std::function<bool(uint32_t)> set_warp_width;
    std::function<bool(rt_counters &)> counters;
    std::function<uint64_t()> current_ray, in_flight;
    std::function<bool(const char *)> write_ppm;
};

}  // namespace

int main(int argc, char **argv)
{
    int device = 0, frames = 4, steps = 32;  // repeat_count = 32, main.cpp:515
    size_t width = 1024, height = 1024, rays = 0;
    uint32_t warp = 64, spp = 0, band_first = 0, band_rows = 0, flags = RT_FLAG_FREE_SCHEDULE | RT_FLAG_ARITH_MAD;  // (-cl-mad-enable, main.cpp:253)
    bool force_rccl = false;
    std::vector<int> devices;
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
        else if (arg == "--gpus") {
            const int n = std::atoi(need());
            for (int d = 0; d < n; d++) devices.push_back(d);
        } else if (arg == "--devices") {
            const std::string list = need();
            for (size_t at = 0; at < list.size();) {
                size_t end = list.find(',', at);
                if (end == std::string::npos) end = list.size();
                devices.push_back(std::atoi(list.substr(at, end - at).c_str()));
                at = end + 1;
            }
        } else if (arg == "--rccl") force_rccl = true;
        else if (arg == "--width") width = std::strtoul(need(), nullptr, 10);
        else if (arg == "--height") height = std::strtoul(need(), nullptr, 10);
        else if (arg == "--rays") rays = std::strtoul(need(), nullptr, 10);
        else if (arg == "--frames") frames = std::atoi(need());
        else if (arg == "--steps") steps = std::atoi(need());
        else if (arg == "--spp") spp = uint32_t(std::atoi(need()));
        else if (arg == "--models") models = need();
        else if (arg == "--warp") warp = uint32_t(std::atoi(need()));
        else if (arg == "--band-first") band_first = uint32_t(std::atoi(need()));
        else if (arg == "--band-rows") band_rows = uint32_t(std::atoi(need()));
        else if (arg == "--arith") {
            const std::string s = need();
            if (s != "strict" && s != "mad") {
                std::cout << kUsage << std::endl;
                return 2;
            }
            flags = (flags & ~RT_FLAG_ARITH_MAD) | (s == "mad" ? RT_FLAG_ARITH_MAD : 0u);
        } else if (arg == "--schedule") {
            const std::string s = need();
            if (s != "free" && s != "reference") {
                std::cout << kUsage << std::endl;
                return 2;
            }
            flags = (flags & ~RT_FLAG_FREE_SCHEDULE) | (s == "free" ? RT_FLAG_FREE_SCHEDULE : 0u);
        } else if (arg == "--out") out = need();
        else {
            std::cout << kUsage << std::endl;
            return arg == "--help" ? 0 : 2;
        }
    }
    if (!rays) rays = width * height;  // main.cpp:516
    if (!width || !height || uint64_t(width) * height > 0x7FFFFFFFull) {
        std::cout << "The image must have between 1 and 2^31 - 1 pixels" << std::endl;
        return 2;
    }
    // the sample limit is a 32-bit count (sample ids are): refuse what does not fit instead of wrapping
    if (uint64_t(spp) * width * (band_rows ? band_rows : height) > 0x7FFFFFFFull) {
        std::cout << "--spp " << spp << " at " << width << "x" << (band_rows ? band_rows : height) << " exceeds 2^31 - 1 samples" << std::endl;
        return 2;
    }
    std::vector<std::string> files;
    for (size_t at = 0; at <= models.size();) {
        size_t end = models.find(';', at);
        if (end == std::string::npos) end = models.size();
        if (end > at) files.push_back(models.substr(at, end - at));
        at = end + 1;
    }
    if (!devices.empty() && band_rows) {
        std::cout << "--band-first/--band-rows render one band on one device; they do not combine with --gpus/--devices" << std::endl;
        return 2;
    }

    rt::RayTracer single(width, height, rays);
    rt::TiledRenderer tiled(width, height);
    Loop loop;
    if (devices.empty() && !force_rccl) {
        single.model_files = files;
        single.warp_width = warp;
        single.flags = flags;
        single.band_first = band_first;
        single.band_rows = band_rows;
        single.sample_limit = spp ? uint32_t(spp * width * (band_rows ? band_rows : height)) : 0;
        if (!single.init(device)) return 1;
        loop = Loop{[&] { return single.init_frame(); }, [&] { return single.make_step(); }, [&] { return single.draw_frame(); },
                    [&](uint32_t w) { return single.set_warp_width(w); }, [&](rt_counters &c) { return single.counters(c); },
                    [&] { return uint64_t(single.current_ray()); }, [&] { return uint64_t(single.rays_in_flight()); },
                    [&](const char *path) { return single.write_ppm(path); }};
    } else {
        if (devices.empty()) devices.push_back(device);
        tiled.model_files = files;
        tiled.warp_width = warp;
        tiled.flags = flags;
        tiled.sample_limit_per_pixel = spp;
        if (!tiled.init(devices)) return 1;
        std::cout << tiled.bands() << " band(s), gathered " << (tiled.uses_rccl() ? "with RCCL" : "by device copies") << std::endl;
        loop = Loop{[&] { return tiled.init_frame(); }, [&] { return tiled.make_step(); }, [&] { return tiled.draw_frame(); },
                    [&](uint32_t w) { return tiled.set_warp_width(w); }, [&](rt_counters &c) { return tiled.counters(c); },
                    [&] { return tiled.current_ray(); }, [&] { return tiled.rays_in_flight(); },
                    [&](const char *path) { return tiled.write_ppm(path); }};
    }
    std::cout << "Ready." << std::endl;
    if (!loop.init_frame() || !loop.draw_frame()) return 1;

    uint64_t cur_ray = loop.current_ray();
    rt_counters before = {};
    loop.counters(before);
    std::cout << std::setprecision(3) << std::fixed;
    auto report = [&](double delta, uint64_t old_ray) {
        cur_ray = loop.current_ray();
        rt_counters after = {};
        loop.counters(after);
        const double all = double(after.primary_spawned - before.primary_spawned) + double(after.shadow_spawned - before.shadow_spawned);
        before = after;
        std::cout << "Frame ready in " << delta << " s, " << (cur_ray - old_ray) << " rays, " << 1e-6 * double(cur_ray - old_ray) / delta
                  << " MR/s (" << 1e-6 * all / delta << " MR/s with shadow rays)." << std::endl;
    };
    if (spp) {
        // Fixed work: every sample below the limit, then granularity 1 so that parked rays finish too.  A few rays
        // can circle for ever (the reference's queue-overflow restart re-creating the same queue): bounded.
        const double start = now_seconds();
        const uint64_t old_ray = cur_ray;
        for (int i = 0; i < steps; i++)
            if (!loop.make_step()) return 1;
        if (!loop.set_warp_width(1)) return 1;
        // (the live count only covers whole warps: parked rays show up in it after the first step at granularity 1)
        uint64_t left = 1, quiet = 0, last = ~0ull;
        for (int guard = 0; left && guard < 4096 && quiet < 20; guard++) {
            for (int i = 0; i < 8; i++)
                if (!loop.make_step()) return 1;
            left = loop.in_flight();
            quiet = left == last && left <= 16 ? quiet + 1 : 0;
            last = left;
        }
        if (!loop.draw_frame()) return 1;
        report(now_seconds() - start, old_ray);
        if (left) std::cout << left << " ray(s) still in flight" << std::endl;
    } else {
        for (int frame = 0; frame < frames; frame++) {
            const double start = now_seconds();
            const uint64_t old_ray = cur_ray;
            for (int i = 0; i < steps; i++)
                if (!loop.make_step()) return 1;
            if (!loop.draw_frame()) return 1;
            report(now_seconds() - start, old_ray);
        }
    }
    if (!loop.write_ppm(out.c_str())) {
        std::cout << "Cannot write " << out << std::endl;
        return 1;
    }
    std::cout << "Image written to " << out << std::endl;
    return 0;
}
