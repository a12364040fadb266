// Developer microbenchmark: rate of scattered accesses to 256-byte records in a 533 MB pool (the ray slots'
// access pattern): one 16-byte load per lane from the record's first line, optionally more pieces of the
// same line, optionally a 16-byte store back.  hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

template <int PIECES, bool STORE>
__global__ void k(const uint32_t *perm, float4 *pool, float4 *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 *rec = pool + size_t(perm[i]) * 16;  // 256-byte records
    float4 acc = rec[0];
#pragma unroll
    for (int p = 1; p < PIECES; p++) {
        const float4 v = rec[p];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (STORE) rec[3] = acc;
    else if (acc.x == 12345.678f) out[i] = acc;
}

int main()
{
    const uint32_t n = 2080768;
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    uint32_t *d_perm; float4 *pool, *out;
    hipMalloc(&d_perm, n * 4); hipMalloc(&pool, size_t(n) * 256); hipMalloc(&out, size_t(n) * 16);
    hipMemcpy(d_perm, perm.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(pool, 0, size_t(n) * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch, double bytes_per) {
        for (int rep = 0; rep < 3; rep++) launch();
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; rep++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-44s %7.1f us  %6.2f G records/s  %6.2f TB/s of lines\n", name, ms * 1e3, n / (ms * 1e6), n * bytes_per / (ms * 1e9));
    };
    const dim3 grid((n + 255) / 256), block(256);
    run("1 x 16 B of line 0 (read)", [&] { k<1, false><<<grid, block>>>(d_perm, pool, out, n); }, 128);
    run("8 x 16 B = whole line 0 (read)", [&] { k<8, false><<<grid, block>>>(d_perm, pool, out, n); }, 128);
    run("16 x 16 B = both lines (read)", [&] { k<16, false><<<grid, block>>>(d_perm, pool, out, n); }, 256);
    run("whole line 0 read + 16 B store to it", [&] { k<8, true><<<grid, block>>>(d_perm, pool, out, n); }, 256);
    return 0;
}
