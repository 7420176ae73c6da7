// Does the matrix pipe run beside the vector pipe on gfx950? Waves of one SIMD do either a chain of
// independent v_mfma_f32_16x16x4_f32 or a chain of v_pk_fma_f32; mode 2 mixes the two roles 1:1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_ITER 8192
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    // roles by the wave's SLOT on its SIMD (HW_REG_HW_ID.WAVE_ID), so that every SIMD runs both kinds side by
    // side whatever the block placement: with 4 waves per SIMD, slots 0..3 -> two of each
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    const bool do_mfma = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 4 || MODE == 6) && (slot & 1u) == 0);
    if (MODE == 2 && (threadIdx.x & 63) == 0) atomicAdd(&out[1 + (slot & 7u)], 1.0f);
    float r = 0;
    if (!do_mfma && MODE >= 5) {   // plain (non-packed) VALU: v_mul_f32 / v_add_f32 chains, 64 per iteration
        float q0 = seed, q1 = 1, q2 = 2, q3 = 3, q4 = 4, q5 = 5, q6 = 6, q7 = 7;
        const float m = 1.0000001f;
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(m));
            }
        }
        r = q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
    } else if (do_mfma && MODE >= 3) {   // bf16 matrix pipe: v_mfma_f32_16x16x32_bf16
        v4f a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        bf8 x, y;
        for (int q = 0; q < 8; ++q) { x[q] = (__bf16)(threadIdx.x * 1e-3f + q); y[q] = (__bf16)(seed + q); }
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {   // 8 MFMAs of ~16 cycles = the same pipe time as 4 f32 MFMAs
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (do_mfma) {
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
    float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    float t0 = run<0>(d), t1 = run<1>(d), t2 = run<2>(d), t3 = run<3>(d), t4 = run<4>(d);
    printf("all waves MFMA: %.3f ms (%.1f cycles per MFMA per SIMD @2.4GHz)\n", t0, t0 * 1e-3 * 2.4e9 / (4.0 * N_ITER * 4));
    printf("all waves pk_fma: %.3f ms (%.2f cycles per pk_fma per SIMD)\n", t1, t1 * 1e-3 * 2.4e9 / (4.0 * N_ITER * 32));
    printf("half MFMA, half pk_fma (same work per wave): %.3f ms; serial would be %.3f, perfect overlap %.3f\n", t2, 0.5 * (t0 + t1), 0.5 * (t0 > t1 ? t0 : t1));
    float t5 = run<5>(d), t6 = run<6>(d);
    printf("all waves v_mul_f32: %.3f ms (%.2f cycles per v_mul per SIMD)\n", t5, t5 * 1e-3 * 2.4e9 / (4.0 * N_ITER * 64));
    printf("half bf16 MFMA, half v_mul_f32: %.3f ms; serial would be %.3f, perfect overlap %.3f\n", t6, 0.5 * (t3 + t5), 0.5 * (t3 > t5 ? t3 : t5));
    float slots[9]; hipMemcpy(slots, d, 36, hipMemcpyDeviceToHost);
    printf("waves per slot id (3 launches):"); for (int i = 1; i < 9; ++i) printf(" %.0f", slots[i]); printf("\n");
    printf("all waves bf16 MFMA: %.3f ms (%.1f cycles per 16x16x32 MFMA per SIMD)\n", t3, t3 * 1e-3 * 2.4e9 / (4.0 * N_ITER * 8));
    printf("half bf16 MFMA, half pk_fma: %.3f ms; serial would be %.3f, perfect overlap %.3f\n", t4, 0.5 * (t3 + t1), 0.5 * (t3 > t1 ? t3 : t1));
    return 0;
}
