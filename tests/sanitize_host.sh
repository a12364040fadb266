#!/bin/bash
# Not a pytest test: host scene builder + oracle under ASan/UBSan/LSan, the threaded build under TSan, on the CPU (no GPU
# needed).  It lives under tests/ because it links the oracle, which only test code may do.  usage: bash tests/sanitize_host.sh
set -e
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/rt_sanitize
mkdir -p $out/cache
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fno-fast-math"
g++ $SAN -std=c++17 -pthread -fPIC -shared -Iinclude -Iraytracer_amd/host raytracer_amd/host/model.cpp raytracer_amd/host/scene.cpp raytracer_amd/host/host_capi.cpp -o $out/librt_host.so
gcc $SAN -std=c11 -fPIC -shared -fopenmp -Iinclude oracle/rt_oracle.c -o $out/liboracle.so -lm
g++ $SAN -std=c++17 -fopenmp -Iinclude -Iraytracer_amd/host -Ioracle tests/sanitize_host.cpp -o $out/sanitize_host -L$out -lrt_host -loracle -Wl,-rpath,$out
RT_MESH_CACHE=$out/cache $out/sanitize_host assets/bun_zipper.ply $out
RT_MESH_CACHE=$out/cache $out/sanitize_host assets/bun_zipper.ply $out   # second run: model files from the binary cache
# the scene build runs its sorts and instance boxes on several threads: the same build under ThreadSanitizer
cat > $out/tsan_build.cpp <<'CPP'
#include <cstdio>
#include <string>
#include "host_capi.h"
int main(int, char **argv)
{
    for (unsigned narrow = 0; narrow < 2; narrow++) {
        const std::string files = std::string(argv[1]) + ";" + argv[2];
        rth_scene_spec spec{files.c_str(), 256, 128, narrow ? 8u : 128u, 256, 144, 36864, 0, 0, 0};
        char err[512] = "";
        rth_scene *scene = rth_scene_build(&spec, err, sizeof err);
        if (!scene) return std::printf("failed: %s\n", err), 1;
        size_t groups;
        rth_scene_groups(scene, &groups);
        std::printf("thread sanitizer: %zu groups built\n", groups);
        rth_scene_free(scene);
    }
}
CPP
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -fPIC -shared -ffp-contract=off -Iinclude -Iraytracer_amd/host raytracer_amd/host/model.cpp raytracer_amd/host/scene.cpp raytracer_amd/host/host_capi.cpp -o $out/librt_host_tsan.so
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -Iinclude -Iraytracer_amd/host $out/tsan_build.cpp -o $out/tsan_build $out/librt_host_tsan.so -Wl,-rpath,$out
RT_MESH_CACHE= RT_PLY_PARALLEL_MIN=0 RT_HOST_THREADS=8 $out/tsan_build assets/bun_zipper.ply $out/sanitize_standin.ply   # (cache off: the files are parsed line by line on several threads)
