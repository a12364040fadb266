// Developer tool: do two kernels launched back to back on ONE stream overlap when the second is launched with
// hipExtAnyOrderLaunch?  Each kernel is one workgroup spinning for a fixed number of clock ticks.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out) *out = wall_clock64();
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long *d;
    hipMalloc(&d, 16);
    const unsigned long long ticks = 100000000ull / 10;  // wall_clock64 runs at 100 MHz: 100 ms
    for (int flags = 0; flags <= 1; flags++) {
        spin<<<1, 64, 0, s>>>(1000, d);
        hipStreamSynchronize(s);
        const auto t0 = std::chrono::steady_clock::now();
        hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, ticks, d);
        hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, ticks, d + 1);
        const hipError_t e = hipStreamSynchronize(s);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("flags %d: %.1f ms for two 100 ms kernels (%s)\n", flags, ms, hipGetErrorString(e));
    }
    return 0;
}
