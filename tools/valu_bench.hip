// Microbenchmark: issue rate of plain vs packed FP32 VALU on gfx950 (one-off measurement that fixes
// the instruction budget of the closest-hit loop; see DESIGN.md "VALU issue model").
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 16384
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m};
    double q0 = a0 + 1.0, q1 = a1 + 1.0, q2 = a2 + 1.0, q3 = a3 + 1.0, qm = 1.0000000001;
    for (int i = 0; i < N_ITER; ++i) {
      _Pragma("unroll") for (int rep = 0; rep < 4; ++rep) {
        if (MODE == 0) {   // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 1) {   // 4 independent v_pk_mul_f32 (8 multiplies)
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        } else if (MODE == 2) {   // 8 v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 3) {   // 4 v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        } else if (MODE == 4) {   // 8 v_add_f32 with an SGPR operand
            asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                         "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));
        } else if (MODE == 5) {   // 4 v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        } else if (MODE == 6) {   // 8 v_alignbit_b32 (VOP3, 3 operands)
            asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n"
                         "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE >= 17) {   // integer ops of the counter RNG (round 2)
            uint32_t *u0 = reinterpret_cast<uint32_t*>(&a0), *u1 = reinterpret_cast<uint32_t*>(&a1), *u2 = reinterpret_cast<uint32_t*>(&a2), *u3 = reinterpret_cast<uint32_t*>(&a3);
            const uint32_t c = 747796405u;
            if (MODE == 17) asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                                         : "+v"(*u0), "+v"(*u1), "+v"(*u2), "+v"(*u3) : "v"(c));
            if (MODE == 18) asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4\n"
                                         : "+v"(*u0), "+v"(*u1), "+v"(*u2), "+v"(*u3) : "v"(c));
            if (MODE == 19) asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n"
                                         : "+v"(*u0), "+v"(*u1), "+v"(*u2), "+v"(*u3) : "v"(c));
            if (MODE == 20) asm volatile("v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3\n"
                                         : "+v"(*u0), "+v"(*u1), "+v"(*u2), "+v"(*u3) : "v"(c));
            if (MODE == 21) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
                                         : "+v"(q0), "+v"(q1) : "v"(c), "v"(*u0) : "vcc");
            if (MODE == 23) asm volatile("v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %4, %4\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %4, %4\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            if (MODE == 24) asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            if (MODE == 25) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");
            if (MODE == 26) asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n"
                                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(m) : "vcc");
            if (MODE == 27) asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n"
                                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(m));
            if (MODE == 28) asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            if (MODE == 29) asm volatile("v_max3_f32 %0, %0, %4, %4\n v_max3_f32 %1, %1, %4, %4\n v_min3_f32 %2, %2, %4, %4\n v_min3_f32 %3, %3, %4, %4\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            if (MODE == 30) asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(pm));
            if (MODE == 31) asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                                         : "+v"(a0) : "v"(m));
            if (MODE == 22) asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n"
                                         : : "v"(*u0), "v"(*u1), "v"(*u2), "v"(*u3) : "s20", "s21", "s22", "s23");
        } else if (MODE >= 8) {   // FP64 and transcendental issue rates without conversions in the loop (round 2)
            static_assert(MODE <= 31, "");
            if (MODE == 8) asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                                        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qm));
            if (MODE == 9) asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                                        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qm));
            if (MODE == 10) asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qm));
            if (MODE == 11) asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            if (MODE == 12) asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            if (MODE == 13) asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            if (MODE == 14) asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q0), "v"(q1), "v"(q2), "v"(q3));
            if (MODE == 15) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            if (MODE == 16) asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (MODE == 7) {   // 8 v_mul_f64? use 4 v_fma_f64
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3, dm = m;
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));
            a0 = d0; a1 = d1; a2 = d2; a3 = d3;
        }
      }
    }
    a7 += float(q0 + q1 + q2 + q3);
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.678f) out[0] = r;
}
template <int MODE>
void run(const char* name, int ops_per_iter, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves/SIMD
    k<MODE><<<grid, 256>>>(d, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<grid, 256>>>(d, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double lane_ops = double(grid) * 256 * N_ITER * 4.0 * ops_per_iter;
    // cycles per wave-instruction on one SIMD: 8 waves per SIMD share it; clock from the device
    int dev = 0, khz = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
    const double instr_per_wave = double(N_ITER) * 4.0 * (MODE == 1 || MODE == 3 || MODE == 5 || MODE == 30 ? ops_per_iter / 2 : ops_per_iter);
    const double cyc = ms * 1e-3 * khz * 1e3 / (instr_per_wave * 8.0);
    printf("%-28s %8.3f ms  %7.2f T lane-ops/s  %6.2f cycles per wave-instruction (8 waves/SIMD, %d MHz)\n", name, ms, lane_ops / (ms * 1e-3) / 1e12, cyc, khz / 1000);
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("v_mul_f32 x8", 8, d);
    run<1>("v_pk_mul_f32 x4 (8 mul)", 8, d);
    run<2>("v_fma_f32 x8", 8, d);
    run<3>("v_pk_add_f32 x4 (8 add)", 8, d);
    run<4>("v_add_f32 sgpr x8", 8, d);
    run<5>("v_pk_fma_f32 x4 (8 fma)", 8, d);
    run<6>("v_alignbit_b32 x8", 8, d);
    run<7>("v_fma_f64 x4 (+cvt)", 4, d);
    run<8>("v_fma_f64 x4", 4, d);
    run<9>("v_mul_f64 x4", 4, d);
    run<10>("v_add_f64 x4", 4, d);
    run<11>("v_rcp_f64 x4", 4, d);
    run<12>("v_rsq_f64 x4", 4, d);
    run<13>("v_cvt_f64_f32 x4", 4, d);
    run<14>("v_cvt_f32_f64 x4", 4, d);
    run<15>("v_rcp_f32 x4", 4, d);
    run<16>("v_sqrt_f32 x4", 4, d);
    run<17>("v_mul_lo_u32 x4", 4, d);
    run<18>("v_mad_u32_u24 x4", 4, d);
    run<19>("v_xor_b32 x4", 4, d);
    run<20>("v_lshrrev_b32 x4", 4, d);
    run<21>("v_mad_u64_u32 x4", 4, d);
    run<22>("v_readlane_b32 x4", 4, d);
    run<23>("v_fmac_f32 (VOP2) x4", 4, d);
    run<24>("v_max/min_f32 (VOP2) x4", 4, d);
    run<25>("v_cndmask_b32 vcc x4", 4, d);
    run<26>("v_cmp_lt_f32 vcc x4", 4, d);
    run<27>("v_mov_b32 x4", 4, d);
    run<28>("v_add/sub_f32 vgpr x4", 4, d);
    run<29>("v_max3/min3_f32 x4", 4, d);
    run<30>("v_pk_fma_f32 sgpr-pair x4 (8 fma)", 8, d);
    run<31>("v_fma_f32 dependent chain x4", 4, d);
    return 0;
}
