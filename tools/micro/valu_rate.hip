// Developer microbenchmark: VALU issue rate of independent f32 instructions per SIMD as a function of
// the number of resident waves, plain vs packed.  hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void k(float *out, uint64_t *cyc, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, s = 0.5f;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (MODE == 0) {
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_sub_f32 %4, %4, %9\n v_sub_f32 %5, %5, %9\n v_min_f32 %6, %6, %9\n v_max_f32 %7, %7, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(s));
        } else {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m};
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *out; uint64_t *cyc;
    hipMalloc(&out, 4 * 256 * 2048 * 8); hipMalloc(&cyc, 8 * 4096);
    const int iters = 2000;
    for (int mode = 0; mode < 2; mode++)
        for (int waves_per_simd : {1, 2, 4, 8}) {
            const int threads = 256 * (waves_per_simd >= 4 ? 4 : waves_per_simd);  // one block per CU
            const int blocks = 256 * (waves_per_simd == 8 ? 2 : 1);
            // threads/64 waves per block spread over 4 SIMDs: 256 threads = 1 wave per SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, threads>>>(out, cyc, iters); else k<1><<<blocks, threads>>>(out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double insts = 128.0 * iters;  // per wave
            printf("mode %s waves/SIMD %d: %.3f ms, memtime ticks/inst (one wave) %.2f, wave-insts per SIMD per us %.1f\n",
                   mode ? "packed" : "plain", waves_per_simd, ms, c / insts, insts * waves_per_simd / (ms * 1e3));
        }
    return 0;
}
