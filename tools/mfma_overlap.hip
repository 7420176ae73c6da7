// Does the matrix pipe run beside the vector pipe on gfx950? Waves of one SIMD do either a chain of
// independent v_mfma_f32_16x16x4_f32 or a chain of v_pk_fma_f32; mode 2 mixes the two roles 1:1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_ITER 8192
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    const int wave = threadIdx.x >> 6;
    // blocks alternate roles too so that every SIMD gets both kinds (a block's 4 waves sit on 4 SIMDs)
    const bool do_mfma = MODE == 0 || (MODE == 2 && ((blockIdx.x + wave) & 1) == 0);
    float r = 0;
    if (do_mfma) {
        v4f a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = threadIdx.x * 1e-3f, y = seed;
        for (int i = 0; i < N_ITER; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        f2 p0 = {seed, 1}, p1 = {2, 3}, p2 = {4, 5}, p3 = {6, 7}, m = {1.0000001f, 0.9999999f};
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {   // 32 pk_fma per iteration ~ 128+ cycles, like 4 MFMAs
                p0 = __builtin_elementwise_fma(p0, m, m); p1 = __builtin_elementwise_fma(p1, m, m);
                p2 = __builtin_elementwise_fma(p2, m, m); p3 = __builtin_elementwise_fma(p3, m, m);
            }
        }
        r = p0.x + p1.y + p2.x + p3.y;
    }
    if (r == 12345.678f) out[0] = r;
}
template <int MODE>
float run(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * 4;   // 4 waves per SIMD
    float ms = 0;
    for (int it = 0; it < 3; ++it) { hipEventRecord(e0); k<MODE><<<grid, 256>>>(d, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 4);
    float t0 = run<0>(d), t1 = run<1>(d), t2 = run<2>(d);
    printf("all waves MFMA: %.3f ms (%.1f cycles per MFMA per SIMD @2.4GHz)\n", t0, t0 * 1e-3 * 2.4e9 / (4.0 * N_ITER * 4));
    printf("all waves pk_fma: %.3f ms (%.2f cycles per pk_fma per SIMD)\n", t1, t1 * 1e-3 * 2.4e9 / (4.0 * N_ITER * 32));
    printf("half MFMA, half pk_fma (same work per wave): %.3f ms; serial would be %.3f, perfect overlap %.3f\n", t2, 0.5 * (t0 + t1), 0.5 * (t0 > t1 ? t0 : t1));
    return 0;
}
