// Closest-hit scan over the sphere arrays for gfx950 (hitFromList, hitPosition.cpp:6-22, over
// sphere::hit, shapes.h:77-97), brute force: every ray is tested against every sphere.
//
// Two levels, results bit-identical to the reference's arithmetic:
//  1. FILTER: a conservative line-sphere test in packed FMA arithmetic (v_pk_fma_f32, two spheres per
//     instruction, 8 instructions per pair; sphere data wave-uniform through scalar loads). It flags
//     every pair that holds a sphere whose REFERENCE discriminant d = b*b - a*c (separately rounded
//     FP32, shapes.h:80-88) could be >= 0: the margin M bounds the rounding error of both
//     evaluations (derivation in DESIGN.md "sphere filter"). One flag bit per pair per lane.
//  2. EXACT: each lane walks its own flagged pairs and evaluates the reference's FP32 expressions
//     (contraction off, same operation order) for both spheres of the pair with packed mul/add; a
//     sphere with d >= 0 goes through the reference's FP64 sqrt / divide / interval tests, skipped
//     only when FP32 bounds on its roots prove that the reference would reject both.
// Spheres that are huge or far from the rest (the ground sphere) would blow up the filter margin of
// every ray; rt_upload_scene puts them in the "exact group", which gets level 2 directly.
#pragma once
#include "rt_device.hpp"

namespace rt {

typedef float f2 __attribute__((ext_vector_type(2)));
// constant address space: a load through it with a wave-uniform address is a scalar load (s_load)
typedef const f2 __attribute__((address_space(4)))* cf2p;
typedef const float __attribute__((address_space(4)))* cfp;
#define RT_CONST_F2(p) ((rt::cf2p)(uintptr_t)(p))
#define RT_CONST_F(p) ((rt::cfp)(uintptr_t)(p))

constexpr int kMaskWords = 16;                 // flag words per lane and chunk (32 pairs each), kept in LDS
constexpr int kChunkPairs = kMaskWords * 32;   // 512 pairs = 1024 spheres per chunk
constexpr int kScanBlock = 256;                // threads per block of every kernel that scans
// Filter margin factor: M = kMarginK * a * (S^2 + So^2 + r2max) with S = |o'| + cmax, So = |o - g| + cmax.
// 96 * 2^-24, against a derived bound of < 64 * 2^-24.
constexpr float kMarginK = 96.0f * 5.9604644775390625e-08f;
// a ray takes the fast path only inside these magnitudes (no overflow, no NaN in either level)
constexpr float kMaxCoord = 1.0e6f, kMinA = 1.0e-12f, kMaxA = 3.0e12f;

struct SphereGroups {
    int n_exact;    // spheres [0, n_exact) of the sphere arrays: level 2 directly
    int n_fpairs;   // filter group: n_fpairs pairs (a multiple of 8), spheres [n_exact, n_exact + 2 n_fpairs)
    const float* __restrict__ fx;    // [2 n_fpairs] centre - g                     (pads: 0)
    const float* __restrict__ fy;
    const float* __restrict__ fz;
    const float* __restrict__ fk;    // [2 n_fpairs] |centre - g|^2 - r^2           (pads: 3e38)
    const float4* __restrict__ fq;   // [2 n_fpairs] per pair (cx0,cx1,cy0,cy1), (cz0,cz1,r2_0,r2_1); pads NaN
    float gx, gy, gz;                // reference point of the filter group
    float cmax, r2max;               // max |centre - g| (rounded up), max r^2 over the filter group
    int fast_ok;                     // 0: some sphere is non-finite / out of range -> list-order scan only
};

// Diagnostic build only (-DRT_STAMPS): cycles per section of the render loop (s_memtime, wave-uniform)
// and wave-level pass counts of the divergent loops. Never compiled into the product library.
#ifdef RT_STAMPS
struct Stamps {
    unsigned long long t[8];     // 0 assign+new sample, 1 exact group, 2 filter, 3 candidate walk, 4 rect+tri, 5 shade, 6 unwind+store, 7 other
    unsigned long long last;
    unsigned int cnt[4];         // per lane, counted on the first active lane only: 0 walk passes, 1 FP64 root passes, 2 exact-group FP64 passes, 3 scans
};
#define RT_STAMP(i) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); stp.t[i] += now_ - stp.last; stp.last = now_; } while (0)
#define RT_COUNT(i) do { if ((__ffsll((long long)__ballot(1)) - 1) == int(threadIdx.x & 63)) ++stp.cnt[i]; } while (0)
#define RT_STAMP_PARAM , Stamps& stp
#define RT_STAMP_ARG , stp
#else
#define RT_STAMP(i) do {} while (0)
#define RT_COUNT(i) do {} while (0)
#define RT_STAMP_PARAM
#define RT_STAMP_ARG
#endif

struct HitState {
    double bt;    // closest_so_far (hitPosition.cpp:9)
    int bi;       // its flattened primitive index, -1 = none
    float btf;    // bt rounded up to float
};

RT_DEV bool reject_t(double t, double t_min, double bt, int idx, int bi) {
    return t < t_min || bt < t || (t == bt && idx < bi);
}

// A sphere whose reference discriminant D is >= 0 (a > 0, everything finite): the reference's roots
// (shapes.h:90-97) in FP64 unless FP32 bounds on them show that both are rejected.
__device__ __forceinline__ void sphere_candidate(float a, float b, float D, int idx, double t_min, float tminf, HitState& h RT_STAMP_PARAM, int cnt_slot = 1) {
    float nb = -b;
    float ia = __builtin_amdgcn_rcpf(a);
    float sf = __builtin_amdgcn_sqrtf(D);
    float e = (__builtin_fabsf(nb) + sf) * ia * 2.0e-6f;   // >= 32 ulp of the larger term; true error < 8 ulp
    float lo1 = (nb - sf) * ia - e;                        // lower bound of the near root
    float hi2 = (nb + sf) * ia + e;                        // upper bound of the far root
    if (lo1 > h.btf || hi2 < tminf) return;                // near root > closest_so_far, or far root < t_min
#ifdef RT_STAMPS
    if (cnt_slot == 1) RT_COUNT(1); else RT_COUNT(2);
#endif
    double sqrtd = __builtin_sqrt(double(D));
    double root = (double(nb) - sqrtd) / double(a);
    if (reject_t(root, t_min, h.bt, idx, h.bi)) {
        root = (double(nb) + sqrtd) / double(a);
        if (reject_t(root, t_min, h.bt, idx, h.bi)) return;
    }
    h.bt = root; h.bi = idx; h.btf = __double2float_ru(root);
}

RT_DEV bool ray_in_fast_range(V3 o, V3 d, float a) {
    return __builtin_fabsf(o.x) <= kMaxCoord && __builtin_fabsf(o.y) <= kMaxCoord && __builtin_fabsf(o.z) <= kMaxCoord &&
           __builtin_fabsf(d.x) <= kMaxCoord && __builtin_fabsf(d.y) <= kMaxCoord && __builtin_fabsf(d.z) <= kMaxCoord &&
           a >= kMinA && a <= kMaxA;
}

// Spheres of one ray (o, d, a = d.lengthSquared() as Ray.h:8 caches it). `words` is this thread's
// column of the block's flag-word array in LDS (stride kScanBlock). sph / sph_prim: (cx,cy,cz,r) and
// flattened index per sphere, exact group first. Requires ray_in_fast_range and sg.fast_ok.
__device__ __forceinline__ void scan_spheres(const SphereGroups& sg, const float4* __restrict__ sph, const int* __restrict__ sph_prim,
                                             V3 o, V3 d, float a, double t_min, float tminf, HitState& h, uint32_t* words RT_STAMP_PARAM) {
    // ---- exact group: reference arithmetic on every sphere ------------------------------------------
    for (int j = 0; j < sg.n_exact; ++j) {
        cfp q = RT_CONST_F(sph + j);
        float cx = q[0], cy = q[1], cz = q[2], r = q[3];
        V3 oc = o - v3(cx, cy, cz);
        float b = dot(oc, d);
        float c = length_squared(oc) - r * r;
        float D = b * b - a * c;
        if (D >= 0.0f) sphere_candidate(a, b, D, sph_prim[j], t_min, tminf, h RT_STAMP_ARG, 2);
    }
    RT_STAMP(1);
    if (sg.n_fpairs == 0) return;
    // ---- filter constants of this ray -----------------------------------------------------------------
    V3 orel = o - v3(sg.gx, sg.gy, sg.gz);
    float ia = __builtin_amdgcn_rcpf(a);
    float s = -(orel.x * d.x + orel.y * d.y + orel.z * d.z) * ia;
    V3 op = v3(__builtin_fmaf(s, d.x, orel.x), __builtin_fmaf(s, d.y, orel.y), __builtin_fmaf(s, d.z, orel.z));   // a point of the line near g
    float od = __builtin_fmaf(op.z, d.z, __builtin_fmaf(op.y, d.y, op.x * d.x));
    float oo = __builtin_fmaf(op.z, op.z, __builtin_fmaf(op.y, op.y, op.x * op.x));
    float S = __builtin_amdgcn_sqrtf(oo) * 1.001f + sg.cmax;
    float So = __builtin_amdgcn_sqrtf(length_squared(orel)) * 1.001f + sg.cmax;
    float M = kMarginK * a * (S * S + So * So + sg.r2max);
    float m2 = M - a * oo, wa = 2.0f * a;
    const f2 DX = {d.x, d.x}, DY = {d.y, d.y}, DZ = {d.z, d.z}, OD = {od, od}, MM = {m2, m2}, NA = {-a, -a};
    const f2 WX = {wa * op.x, wa * op.x}, WY = {wa * op.y, wa * op.y}, WZ = {wa * op.z, wa * op.z};
    const f2 OX = {o.x, o.x}, OY = {o.y, o.y}, OZ = {o.z, o.z}, AA = {a, a};

    for (int p0 = 0; p0 < sg.n_fpairs; p0 += kChunkPairs) {
        const int n_batches = ((sg.n_fpairs - p0 < kChunkPairs ? sg.n_fpairs - p0 : kChunkPairs)) >> 3;   // 8 pairs each
        const int n_words = (n_batches + 3) >> 2;
        // ---- level 1: one flag bit per pair, 32 pairs per word, words to LDS ---------------------------
        cf2p fx = RT_CONST_F2(sg.fx) + p0, fy = RT_CONST_F2(sg.fy) + p0, fz = RT_CONST_F2(sg.fz) + p0, fk = RT_CONST_F2(sg.fk) + p0;
        uint32_t summary = 0;
        int batch = 0;
        for (int w = 0; w < n_words; ++w) {
            uint32_t m = 0;
            const int nb = n_batches - batch < 4 ? n_batches - batch : 4;
            for (int bb = 0; bb < nb; ++bb, ++batch) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = batch * 8 + jj;
                    f2 X = fx[j], Y = fy[j], Z = fz[j], K = fk[j];
                    f2 bq = __builtin_elementwise_fma(-Z, DZ, __builtin_elementwise_fma(-Y, DY, __builtin_elementwise_fma(-X, DX, OD)));
                    f2 Q = __builtin_elementwise_fma(NA, K, MM);
                    Q = __builtin_elementwise_fma(WX, X, Q);
                    Q = __builtin_elementwise_fma(WY, Y, Q);
                    Q = __builtin_elementwise_fma(WZ, Z, Q);
                    f2 T = __builtin_elementwise_fma(bq, bq, Q);   // >= 0 (sign bit clear): flagged
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x) & __float_as_uint(T.y), 31);
                }
            }
            if (nb < 4) m = (m << (32 - 8 * nb)) | (0xffffffffu >> (8 * nb));
            const uint32_t flagged = ~m;   // pair p of this word <-> bit 31 - p
            words[w * kScanBlock] = flagged;
            summary |= (flagged != 0u ? 1u : 0u) << w;
        }
        RT_STAMP(2);
        // ---- level 2: each lane walks its own flagged pairs ---------------------------------------------
        uint32_t cur = 0;
        int wbase = 0;
        while (cur != 0u || summary != 0u) {
            RT_COUNT(0);
            if (cur == 0u) {
                const int w = __builtin_ctz(summary);
                summary &= summary - 1u;
                cur = words[w * kScanBlock];
                wbase = w * 32;
            }
            const int p = __builtin_clz(cur);
            cur &= ~(0x80000000u >> p);
            const int pair = p0 + wbase + p;
            const float4 A = sg.fq[2 * pair], B = sg.fq[2 * pair + 1];
            const f2 cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, r2 = {B.z, B.w};
            const f2 x = OX - cx, y = OY - cy, z = OZ - cz;
            const f2 b = x * DX + y * DY + z * DZ;        // dotProduct(originCenter, direction), shapes.h:83
            const f2 c = x * x + y * y + z * z - r2;      // lengthSquared() - radiusSquared, shapes.h:84
            const f2 D = b * b - AA * c;                  // shapes.h:86
            const int j = sg.n_exact + 2 * pair;
            uint32_t pend = (D.x >= 0.0f ? 1u : 0u) | (D.y >= 0.0f ? 2u : 0u);   // mostly 0 or one bit
            while (pend != 0u) {
                const bool second = (pend & 1u) == 0u;
                pend &= pend - 1u;
                sphere_candidate(a, second ? b.y : b.x, second ? D.y : D.x, sph_prim[j + (second ? 1 : 0)], t_min, tminf, h RT_STAMP_ARG);
            }
        }
        RT_STAMP(3);
    }
}

}  // namespace rt
