// Prototype of the exact ray-sphere discriminant scan (shapes.h:80-88 arithmetic, no contraction):
// measures cycles per ray-sphere test for several instruction selections. Build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define NS 512   // spheres (multiple of 8)
#define REPS 128  // scans per thread

struct Sph { const float* __restrict__ cx; const float* __restrict__ cy; const float* __restrict__ cz; const float* __restrict__ r2; };

// V0: scalar ops, sphere data through scalar loads
__device__ __forceinline__ uint32_t scan_v0(const Sph& s, float ox, float oy, float oz, float dx, float dy, float dz, float a) {
    uint32_t acc = 0;
    for (int j0 = 0; j0 < NS; j0 += 32) {
        uint32_t m = 0;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            int j = j0 + jj;
            float x = ox - s.cx[j], y = oy - s.cy[j], z = oz - s.cz[j];
            float b = x * dx + y * dy + z * dz;
            float c = x * x + y * y + z * z - s.r2[j];
            float D = b * b - a * c;
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(D), 31);
        }
        acc ^= m * 2654435761u + (acc >> 3);
    }
    return acc;
}
// V1: packed float2 ops on sphere pairs
__device__ __forceinline__ uint32_t scan_v1(const Sph& s, float ox, float oy, float oz, float dx, float dy, float dz, float a) {
    uint32_t acc = 0;
    const f2 *cx = (const f2*)s.cx, *cy = (const f2*)s.cy, *cz = (const f2*)s.cz, *r2 = (const f2*)s.r2;
    f2 OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz}, DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz}, A = {a, a};
    for (int j0 = 0; j0 < NS / 2; j0 += 16) {
        uint32_t m = 0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            int j = j0 + jj;
            f2 x = OX - cx[j], y = OY - cy[j], z = OZ - cz[j];
            f2 b = x * DX + y * DY + z * DZ;
            f2 c = x * x + y * y + z * z - r2[j];
            f2 D = b * b - A * c;
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(D.x), 31);
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(D.y), 31);
        }
        acc ^= m * 2654435761u + (acc >> 3);
    }
    return acc;
}


typedef const float __attribute__((address_space(4))) * cfp;
typedef const f2 __attribute__((address_space(4))) * cf2p;
// V2: conservative FMA line-distance filter on sphere pairs (8 packed ops per pair)
__device__ __forceinline__ uint32_t scan_v2(const Sph& s, float ox, float oy, float oz, float dx, float dy, float dz, float a) {
    uint32_t acc = 0;
    cf2p cx = (cf2p)(uintptr_t)s.cx, cy = (cf2p)(uintptr_t)s.cy, cz = (cf2p)(uintptr_t)s.cz, ck = (cf2p)(uintptr_t)s.r2;
    float od = ox * dx + oy * dy + oz * dz, oo = ox * ox + oy * oy + oz * oz;
    float M2 = 1e-3f * a - a * oo, w = 2.0f * a;
    f2 DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz}, OD = {od, od}, MM = {M2, M2}, NA = {-a, -a};
    f2 WX = {w * ox, w * ox}, WY = {w * oy, w * oy}, WZ = {w * oz, w * oz};
    for (int j0 = 0; j0 < NS / 2; j0 += 32) {
        uint32_t m = 0;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            int j = j0 + jj;
            f2 X = cx[j], Y = cy[j], Z = cz[j], K = ck[j];
            f2 b = __builtin_elementwise_fma(-Z, DZ, __builtin_elementwise_fma(-Y, DY, __builtin_elementwise_fma(-X, DX, OD)));
            f2 Q = __builtin_elementwise_fma(NA, K, MM);
            Q = __builtin_elementwise_fma(WX, X, Q);
            Q = __builtin_elementwise_fma(WY, Y, Q);
            Q = __builtin_elementwise_fma(WZ, Z, Q);
            f2 T = __builtin_elementwise_fma(b, b, Q);
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x) & __float_as_uint(T.y), 31);
        }
        acc ^= m * 2654435761u + (acc >> 3);
    }
    return acc;
}


// V3: V2 with the next batch's scalar loads issued before the current batch is computed (4 pairs per batch)
#define FILTER4(X, Y, Z, K)                                                                                             \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                     \
        f2 b = __builtin_elementwise_fma(-Z[q], DZ, __builtin_elementwise_fma(-Y[q], DY, __builtin_elementwise_fma(-X[q], DX, OD))); \
        f2 Q = __builtin_elementwise_fma(NA, K[q], MM);                                                                 \
        Q = __builtin_elementwise_fma(WX, X[q], Q);                                                                     \
        Q = __builtin_elementwise_fma(WY, Y[q], Q);                                                                     \
        Q = __builtin_elementwise_fma(WZ, Z[q], Q);                                                                     \
        f2 T = __builtin_elementwise_fma(b, b, Q);                                                                      \
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x) & __float_as_uint(T.y), 31);                              \
    }
__device__ __forceinline__ uint32_t scan_v3(const Sph& s, float ox, float oy, float oz, float dx, float dy, float dz, float a) {
    uint32_t acc = 0;
    cf2p cx = (cf2p)(uintptr_t)s.cx, cy = (cf2p)(uintptr_t)s.cy, cz = (cf2p)(uintptr_t)s.cz, ck = (cf2p)(uintptr_t)s.r2;
    float od = ox * dx + oy * dy + oz * dz, oo = ox * ox + oy * oy + oz * oz;
    float M2 = 1e-3f * a - a * oo, w = 2.0f * a;
    f2 DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz}, OD = {od, od}, MM = {M2, M2}, NA = {-a, -a};
    f2 WX = {w * ox, w * ox}, WY = {w * oy, w * oy}, WZ = {w * oz, w * oz};
    f2 X[4], Y[4], Z[4], K[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { X[q] = cx[q]; Y[q] = cy[q]; Z[q] = cz[q]; K[q] = ck[q]; }
    uint32_t m = 0;
    for (int j0 = 4; j0 < NS / 2; j0 += 4) {
        f2 Xn[4], Yn[4], Zn[4], Kn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { Xn[q] = cx[j0 + q]; Yn[q] = cy[j0 + q]; Zn[q] = cz[j0 + q]; Kn[q] = ck[j0 + q]; }
        FILTER4(X, Y, Z, K)
        if ((j0 & 28) == 0) { acc ^= m * 2654435761u + (acc >> 3); m = 0; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { X[q] = Xn[q]; Y[q] = Yn[q]; Z[q] = Zn[q]; K[q] = Kn[q]; }
    }
    FILTER4(X, Y, Z, K)
    acc ^= m * 2654435761u + (acc >> 3);
    return acc;
}

template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ pcx, const float* __restrict__ pcy, const float* __restrict__ pcz, const float* __restrict__ pr2, const float* __restrict__ rays, uint32_t* __restrict__ out, long long* __restrict__ cyc) {
    extern __shared__ float lds_dummy[];
    int t = blockIdx.x * 256 + threadIdx.x;
    if (rays[0] == 1234.5f) lds_dummy[threadIdx.x] = 1.0f;
    Sph s{pcx, pcy, pcz, pr2};
    float ox = rays[t * 6], oy = rays[t * 6 + 1], oz = rays[t * 6 + 2], dx = rays[t * 6 + 3], dy = rays[t * 6 + 4], dz = rays[t * 6 + 5];
    uint32_t acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime(); long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < REPS; ++r) {
        float a = dx * dx + dy * dy + dz * dz;
        uint32_t m = V == 0 ? scan_v0(s, ox, oy, oz, dx, dy, dz, a) : V == 1 ? scan_v1(s, ox, oy, oz, dx, dy, dz, a) : V == 2 ? scan_v2(s, ox, oy, oz, dx, dy, dz, a) : scan_v3(s, ox, oy, oz, dx, dy, dz, a);
        acc ^= m;
        ox += 0.25f * float(m & 3u) - 0.3f;   // perturb the ray so that scans differ
        dz -= 0.001f * float((m >> 5) & 3u);
    }
    long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = acc;
    if (threadIdx.x == 0 && blockIdx.x < 2) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    std::vector<float> cx(NS), cy(NS), cz(NS), r2(NS);
    srand(1);
    auto rnd = []() { return float(rand()) / RAND_MAX; };
    for (int i = 0; i < NS; ++i) { cx[i] = 22 * rnd() - 11; cy[i] = 0.2f; cz[i] = 22 * rnd() - 11; r2[i] = 0.2f * 0.2f; }
    int bpc = getenv("BPC") ? atoi(getenv("BPC")) : 4; int grid = 256 * bpc, nthreads = grid * 256; size_t lds = 160 * 1024 / bpc; printf("blocks/CU %d lds %zu\n", bpc, lds);
    std::vector<float> rays(nthreads * 6);
    for (int i = 0; i < nthreads; ++i) { rays[6*i] = 13; rays[6*i+1] = 2; rays[6*i+2] = 3; rays[6*i+3] = -13 + 4*rnd(); rays[6*i+4] = -2 + rnd(); rays[6*i+5] = -3 + 4*rnd(); }
    float *dcx, *dcy, *dcz, *dr2, *drays; uint32_t* dout[4]; long long* dcyc;
    hipMalloc(&dcx, NS*4); hipMalloc(&dcy, NS*4); hipMalloc(&dcz, NS*4); hipMalloc(&dr2, NS*4); hipMalloc(&drays, rays.size()*4);
    hipMalloc(&dout[0], nthreads*4); hipMalloc(&dout[1], nthreads*4); hipMalloc(&dout[2], nthreads*4); hipMalloc(&dout[3], nthreads*4); hipMalloc(&dcyc, 32);
    hipMemcpy(dcx, cx.data(), NS*4, hipMemcpyHostToDevice); hipMemcpy(dcy, cy.data(), NS*4, hipMemcpyHostToDevice);
    hipMemcpy(dcz, cz.data(), NS*4, hipMemcpyHostToDevice); hipMemcpy(dr2, r2.data(), NS*4, hipMemcpyHostToDevice);
    hipMemcpy(drays, rays.data(), rays.size()*4, hipMemcpyHostToDevice);
    Sph s{dcx, dcy, dcz, dr2};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint32_t> h[4];
    for (int v = 0; v < 4; ++v) {
        float ms = 0;
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            if (v == 0) k<0><<<grid, 256, lds>>>(dcx, dcy, dcz, dr2, drays, dout[v], dcyc); else if (v == 1) k<1><<<grid, 256, lds>>>(dcx, dcy, dcz, dr2, drays, dout[v], dcyc); else if (v == 2) k<2><<<grid, 256, lds>>>(dcx, dcy, dcz, dr2, drays, dout[v], dcyc); else k<3><<<grid, 256, lds>>>(dcx, dcy, dcz, dr2, drays, dout[v], dcyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        long long cy4[4]; hipMemcpy(cy4, dcyc, 32, hipMemcpyDeviceToHost); long long cyc = cy4[0]; printf("  block0 memtime %lld realtime(100MHz) %lld -> %.3f ms, clock %.3f GHz; block1 %lld\n", cy4[0], cy4[1], cy4[1] / 1e5, cy4[0] / (cy4[1] * 10.0), cy4[2]);
        h[v].resize(nthreads); hipMemcpy(h[v].data(), dout[v], nthreads*4, hipMemcpyDeviceToHost);
        double tests = double(nthreads) * REPS * NS;
        double tflops = tests * 17 / (ms * 1e-3) / 1e12;
        // SIMD cycles per wave-level test at 2.4 GHz: 1024 SIMDs
        double wave_tests = tests / 64;
        printf("V%d: %.3f ms, %.1f Gtests/s, %.1f TFLOP/s algorithmic (%.1f%% of 157.3), block0 wave cycles %lld (clock64), est %.1f SIMD-cycles/wave-test @2.4GHz\n",
               v, ms, tests / (ms * 1e-3) / 1e9, tflops, 100 * tflops / 157.3, cyc, ms * 1e-3 * 2.4e9 * 1024 / wave_tests);
    }
    size_t diff = 0; for (int i = 0; i < nthreads; ++i) diff += h[0][i] != h[1][i];
    printf("mismatching threads V0 vs V1: %zu (acc schemes differ in bit order; expected nonzero unless identical order)\n", diff);
    return 0;
}
