// Developer microbenchmark: is a short Newton sequence on v_rcp_f32 equal to the correctly rounded 1.0f / x
// for every float?  Runs all 2^32 bit patterns and reports, per variant, the mismatches with and without
// the guard on the first residual.  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 rcp_exact.hip -o rcp_exact
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ bool same(float a, float b)
{
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

// variant 0: two Newton steps; variant 1: one Newton step, two residual corrections (the division expansion without scaling)
template <int VARIANT>
__device__ __forceinline__ float fast(float w, float &e0)
{
    float y = __builtin_amdgcn_rcpf(w);
    e0 = __builtin_fmaf(-w, y, 1.0f);
    y = __builtin_fmaf(y, e0, y);
    float e = __builtin_fmaf(-w, y, 1.0f);
    float q = __builtin_fmaf(y, e, y);
    if (VARIANT == 1) {
        e = __builtin_fmaf(-w, q, 1.0f);
        q = __builtin_fmaf(e, y, q);
    }
    return q;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat(float x) { return f32x2{x, x}; }
// the division's own sequence with its multiply-adds packed (rt_device.h: reciprocal2)
__device__ __forceinline__ f32x2 reciprocal2(f32x2 w)
{
    bool unused0, unused1, scaled0, scaled1;
    const f32x2 d = {__builtin_amdgcn_div_scalef(1.0f, w.x, false, &unused0), __builtin_amdgcn_div_scalef(1.0f, w.y, false, &unused1)};
    const f32x2 n = {__builtin_amdgcn_div_scalef(1.0f, w.x, true, &scaled0), __builtin_amdgcn_div_scalef(1.0f, w.y, true, &scaled1)};
    const f32x2 y0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 e = __builtin_elementwise_fma(-d, y0, splat(1.0f));
    const f32x2 y1 = __builtin_elementwise_fma(e, y0, y0);
    const f32x2 q0 = n * y1;
    const f32x2 r0 = __builtin_elementwise_fma(-d, q0, n);
    const f32x2 q1 = __builtin_elementwise_fma(r0, y1, q0);
    const f32x2 r1 = __builtin_elementwise_fma(-d, q1, n);
    return f32x2{__builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r1.x, y1.x, q1.x, scaled0), w.x, 1.0f),
                 __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r1.y, y1.y, q1.y, scaled1), w.y, 1.0f)};
}

struct Tally {
    unsigned long long bad_packed;
    uint32_t first_bad_packed;
    unsigned long long bad_raw[2], bad_guarded[2], guarded_out[2];
    uint32_t first_bad[2], min_exp_bad[2], max_exp_bad[2];
    unsigned long long bad_by_exp[2][256];  // mismatches without guard, by biased exponent of the input
};

__global__ void k(Tally *t, uint32_t base)
{
    const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float w = __uint_as_float(bits);
    const float ref = 1.0f / w;
    float e0;
    const float q0 = fast<0>(w, e0), q1 = fast<1>(w, e0);
    const bool ok = __builtin_fabsf(e0) < 0x1p-20f;
    // packed sequence: this value in one half, a neighbour's in the other
    const f32x2 both = reciprocal2(f32x2{w, __uint_as_float(bits ^ 0x00400001u)});
    const float other = 1.0f / __uint_as_float(bits ^ 0x00400001u);
    if (!same(both.x, ref) || !same(both.y, other)) {
        atomicAdd(&t->bad_packed, 1ull);
        atomicMin(&t->first_bad_packed, bits);
    }
    const float qs[2] = {q0, q1};
    for (int v = 0; v < 2; v++) {
        if (!same(qs[v], ref)) {
            atomicAdd(&t->bad_raw[v], 1ull);
            atomicAdd(&t->bad_by_exp[v][(bits >> 23) & 0xFF], 1ull);
            if (ok) {
                atomicAdd(&t->bad_guarded[v], 1ull);
                atomicMin(&t->first_bad[v], bits);
            }
            const uint32_t ex = (bits >> 23) & 0xFF;
            if (ok) {
                atomicMin(&t->min_exp_bad[v], ex);
                atomicMax(&t->max_exp_bad[v], ex);
            }
        }
        if (!ok) atomicAdd(&t->guarded_out[v], 1ull);
    }
}

int main()
{
    Tally *d, h;
    hipMalloc(&d, sizeof(Tally));
    memset(&h, 0, sizeof(h));
    for (int v = 0; v < 2; v++) { h.first_bad[v] = 0xFFFFFFFFu; h.min_exp_bad[v] = 0xFFFFFFFFu; }
    h.first_bad_packed = 0xFFFFFFFFu;
    hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
    for (uint32_t chunk = 0; chunk < 64; chunk++) {  // 64 launches of 2^26 values
        k<<<(1u << 26) / 256, 256>>>(d, chunk << 26);
    }
    hipDeviceSynchronize();
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; v++)
        printf("variant %d: mismatches without guard %llu, with guard %llu (first 0x%08x, exponents %u..%u), values sent to the full division %llu\n",
               v, h.bad_raw[v], h.bad_guarded[v], h.first_bad[v], h.min_exp_bad[v], h.max_exp_bad[v], h.guarded_out[v]);
    for (int v = 0; v < 2; v++) {
        printf("variant %d, biased exponents of the inputs with a mismatch (no guard):", v);
        for (int e = 0; e < 256; e++)
            if (h.bad_by_exp[v][e]) printf(" %d:%llu", e, h.bad_by_exp[v][e]);
        printf("\n");
    }
    printf("packed division sequence: mismatches %llu (first 0x%08x)\n", h.bad_packed, h.first_bad_packed);
    return 0;
}
