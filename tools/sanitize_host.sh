#!/bin/bash
# Developer tool: host scene builder + oracle under ASan/UBSan/LSan on the CPU (no GPU needed).  usage: bash tools/sanitize_host.sh
set -e
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/rt_sanitize
mkdir -p $out/cache
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fno-fast-math"
g++ $SAN -std=c++17 -fPIC -shared -Iinclude -Iraytracer_amd/host raytracer_amd/host/model.cpp raytracer_amd/host/scene.cpp raytracer_amd/host/host_capi.cpp -o $out/librt_host.so
gcc $SAN -std=c11 -fPIC -shared -fopenmp -Iinclude oracle/rt_oracle.c -o $out/liboracle.so -lm
g++ $SAN -std=c++17 -fopenmp -Iinclude -Iraytracer_amd/host -Ioracle tools/sanitize_host.cpp -o $out/sanitize_host -L$out -lrt_host -loracle -Wl,-rpath,$out
RT_MESH_CACHE=$out/cache $out/sanitize_host assets/bun_zipper.ply $out
RT_MESH_CACHE=$out/cache $out/sanitize_host assets/bun_zipper.ply $out   # second run: model files from the binary cache
