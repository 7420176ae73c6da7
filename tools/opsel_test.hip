// Diagnostic: does v_pk_fma_f32 honour op_sel / op_sel_hi on a VGPR pair (pick one half of a 64-bit
// register for both results), and what does it cost? Build: hipcc --offload-arch=gfx950 -O3 -o opsel_test opsel_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 fma_lo(f2 c, f2 x, f2 acc) {   // (c.x * x.x + acc.x, c.x * x.y + acc.y)
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(c), "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ f2 fma_hi(f2 c, f2 x, f2 acc) {   // (c.y * x.x + acc.x, c.y * x.y + acc.y)
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(c), "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ f2 fma_lo_s(f2 c, f2 x, f2 acc) {   // x in SGPRs
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(c), "s"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ f2 fma_hi_s(f2 c, f2 x, f2 acc) {
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(c), "s"(x), "v"(acc));
    return r;
}

__global__ void check(const float* in, float* out) {
    f2 c = {in[threadIdx.x * 2], in[threadIdx.x * 2 + 1]};
    f2 x = {in[200], in[201]};
    f2 acc = {in[202], in[203]};
    f2 a = fma_lo(c, x, acc), b = fma_hi(c, x, acc);
    f2 xs = {__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x.x))),
             __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x.y)))};
    f2 a2 = fma_lo_s(c, xs, acc), b2 = fma_hi_s(c, xs, acc);
    float* o = out + threadIdx.x * 8;
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = a2.x; o[5] = a2.y; o[6] = b2.x; o[7] = b2.y;
}

template <int MODE>
__global__ void __launch_bounds__(256) timing(const float* in, float* out, int iters) {
    f2 c0 = {in[threadIdx.x], in[threadIdx.x + 1]}, c1 = {in[threadIdx.x + 2], in[threadIdx.x + 3]};
    f2 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f2{in[k], in[k + 8]};
    f2 xs = {__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, in[300]))),
             __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, in[301])))};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MODE == 0) acc[k] = __builtin_elementwise_fma(c0, xs, acc[k]);       // plain packed, SGPR operand
                else if (MODE == 1) acc[k] = (k & 1) ? fma_hi_s(c0, xs, acc[k]) : fma_lo_s(c0, xs, acc[k]);
                else if (MODE == 2) acc[k] = __builtin_elementwise_fma(c0, c1, acc[k]);  // plain packed, VGPR operands
                else acc[k] = (k & 1) ? fma_hi(c0, c1, acc[k]) : fma_lo(c0, c1, acc[k]);
            }
        }
    }
    f2 s = acc[0];
    for (int k = 1; k < 8; ++k) s += acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

int main() {
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 0.001f * float((i * 7919) % 1000) - 0.3f;
    float *din, *dout;
    hipMalloc(&din, 512 * 4); hipMalloc(&dout, 1024 * 256 * 4 * 8);
    hipMemcpy(din, h.data(), 512 * 4, hipMemcpyHostToDevice);
    check<<<1, 64>>>(din, dout);
    std::vector<float> o(64 * 8);
    hipMemcpy(o.data(), dout, 64 * 8 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        float cx = h[2 * t], cy = h[2 * t + 1], x0 = h[200], x1 = h[201], a0 = h[202], a1 = h[203];
        float e[8] = {fmaf(cx, x0, a0), fmaf(cx, x1, a1), fmaf(cy, x0, a0), fmaf(cy, x1, a1),
                      fmaf(cx, x0, a0), fmaf(cx, x1, a1), fmaf(cy, x0, a0), fmaf(cy, x1, a1)};
        for (int k = 0; k < 8; ++k) if (o[t * 8 + k] != e[k]) { if (bad < 8) printf("lane %d out %d: %g expected %g\n", t, k, o[t * 8 + k], e[k]); ++bad; }
    }
    printf("op_sel check: %d mismatches\n", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 1024 * 2;
    const char* names[4] = {"plain, SGPR-pair operand", "op_sel, SGPR-pair operand", "plain, VGPR operands", "op_sel, VGPR operands"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) timing<0><<<blocks, 256>>>(din, dout, iters);
            else if (mode == 1) timing<1><<<blocks, 256>>>(din, dout, iters);
            else if (mode == 2) timing<2><<<blocks, 256>>>(din, dout, iters);
            else timing<3><<<blocks, 256>>>(din, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) {
                double instr_per_simd = double(blocks) * 4 * iters * 32 / 1024.0;   // wave-instructions per SIMD
                printf("%-28s %.3f ms  %.2f cycles per v_pk_fma_f32 per SIMD (at 2.4 GHz)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
            }
        }
    }
    return bad != 0;
}
