// Brute-force check of div3_zero_plus (rt_device.hpp) against the compiler's IEEE division on the device: random operand
// triples over the whole guarded exponent range and beyond it, plus zeros, denormals, infinities and NaNs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
//         -I ray_amd/csrc -o tools/div_check tools/div_check.hip && tools/div_check
#include "rt_device.hpp"
#include <cstdio>
using namespace rt;

__device__ uint32_t hash(uint32_t v) { return pcg_hash(v); }
__device__ float make(uint32_t h, uint32_t sel) {
    // sel: 0..15 special values now and then, otherwise sign | exponent in [127 - 48, 127 + 48] | random mantissa
    const uint32_t k = sel & 63u;
    if (k == 0) return 0.0f;
    if (k == 1) return -0.0f;
    if (k == 2) return __uint_as_float(h & 0x007fffffu);                  // denormal
    if (k == 3) return __uint_as_float(0x7f800000u | (h & 0x80000000u));  // inf
    if (k == 4) return __uint_as_float(0x7fc00000u);                      // NaN
    if (k == 5) { const float f = __uint_as_float(h); return f != f ? __uint_as_float(h | 0x00400000u) : f; }   // anything; NaNs quiet, as
                                                                          // every result of arithmetic is (the caller's operands are products / conversions)
    const uint32_t ex = 127u - 48u + (hash(h ^ 0x9e3779b9u) % 97u);
    return __uint_as_float((h & 0x807fffffu) | (ex << 23));
}
__device__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

__global__ void check(unsigned long long n, unsigned long long* bad, unsigned long long* fast, float* ex) {
    unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long nb = 0, nf = 0;
    for (; i < n; i += stride) {
        const uint32_t h0 = hash(uint32_t(i)), h1 = hash(h0 + uint32_t(i >> 32)), h2 = hash(h1), h3 = hash(h2), h4 = hash(h3);
        V3 v = v3(make(h0, h4), make(h1, h4 >> 6), make(h2, h4 >> 12));
        if ((h4 >> 30) == 0u) v = v3(0.0f, (h4 & 1u) ? -0.0f : 0.0f, 0.0f);              // all three zero
        float d = make(h3, h4 >> 18);
        if ((h4 >> 24 & 3u) != 0u) d = __builtin_fabsf(d);                               // mostly positive, like a pdf
        const V3 got = div3_zero_plus(v, d);
        const V3 want = v3(0.0f + v.x / d, 0.0f + v.y / d, 0.0f + v.z / d);
        const V3 got2 = div3(v, d);      // the form that keeps signed zeros
        const bool bad2 = !same(got2.x, v.x / d) || !same(got2.y, v.y / d) || !same(got2.z, v.z / d);
        if (bad2 || !same(got.x, want.x) || !same(got.y, want.y) || !same(got.z, want.z)) {
            ++nb;
            const unsigned long long slot = atomicAdd(bad + 2, 1ull);
            if (slot < 8) { float* e = ex + 10 * slot; e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = d; e[4] = got.x; e[5] = got.y; e[6] = got.z; e[7] = want.x; e[8] = want.y; e[9] = want.z; }
        }
        if (div3_shared_ok(v, d)) ++nf;
    }
    if (nb) atomicAdd(bad, nb);
    if (nf) atomicAdd(fast, nf);
}

int main() {
    unsigned long long *d_out, out[3] = {0, 0, 0};
    float *d_ex, ex[80] = {};
    if (hipMalloc(&d_out, 24) != hipSuccess || hipMalloc(&d_ex, sizeof ex) != hipSuccess) { std::printf("no device\n"); return 1; }
    (void)hipMemset(d_out, 0, 24);
    (void)hipMemset(d_ex, 0, sizeof ex);
    const unsigned long long n = 1ull << 32;
    check<<<4096, 256>>>(n, d_out, d_out + 1, d_ex);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 1; }
    (void)hipMemcpy(out, d_out, 24, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ex, d_ex, sizeof ex, hipMemcpyDeviceToHost);
    for (unsigned long long k = 0; k < out[2] && k < 8; ++k) {
        const float* e = ex + 10 * k;
        std::printf("  n = (%a, %a, %a) d = %a: got (%a, %a, %a) want (%a, %a, %a)\n", e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9]);
    }
    std::printf("div3_zero_plus and div3 against IEEE division: %llu operand triples, %llu through the shared-reciprocal chain, %llu differing results\n", n, out[1], out[0]);
    return out[0] != 0;
}
