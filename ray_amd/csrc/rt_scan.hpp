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
//     (contraction off, same operation order) for both spheres of the pair with packed mul/add. A
//     sphere with d >= 0 is a candidate. FP32 bounds on its roots order it against the best
//     candidate so far; the reference's FP64 sqrt / divide / interval tests (shapes.h:90-97) run only
//     for the candidate that survives, or when two candidates' bounds overlap (ties included).
// Spheres that are huge or far from the rest (the ground sphere) would blow up the filter margin of
// every ray; rt_upload_scene puts them in the "exact group", whose pairs are always flagged.
//  0. BATCH BOUNDS: the filter group is stored in spatially compact batches of 8 pairs (16 items), each
//     with a bounding ball. The same packed test, on the ball widened by the ray's filter margin, tells
//     per lane whether the line can come near any item of the batch, and, for t_min >= 0, whether
//     the whole ball lies behind the ray's origin. A batch that no lane of the wave needs is not
//     filtered at all (wave-uniform branch); a lane that does not need it gets no flags from it.
//     Still brute force in the contract's sense: no item that the reference could hit is ever dropped.
#pragma once
#include "rt_device.hpp"

namespace rt {

typedef float f2 __attribute__((ext_vector_type(2)));
// constant address space: a load through it with a wave-uniform address is a scalar load (s_load)
typedef const f2 __attribute__((address_space(4)))* cf2p;
#define RT_CONST_F2(p) ((rt::cf2p)(uintptr_t)(p))

constexpr int kMaskWords = 16;                 // flag words per lane and chunk (32 pairs each), kept in LDS
constexpr int kScanBlock = 256;                // threads per block of every kernel that scans
// Filter margin factor: M = kMarginK * a * (S^2 + So^2 + r2max) with S = |o'| + cmax, So = |o - g| + cmax.
// 96 * 2^-24, against a derived bound of < 64 * 2^-24.
constexpr float kMarginK = 96.0f * 5.9604644775390625e-08f;
// a ray takes the fast path only inside these magnitudes (no overflow, no NaN in either level)
constexpr float kMaxCoord = 1.0e6f, kMinA = 1.0e-12f, kMaxA = 3.0e12f;

// One filter group (the spheres, or the triangles through their bounding spheres, ray_amd_tri.h).
// Pair index space: words [0, n_ewords) hold the exact group (n_epairs real pairs, always flagged, the
// rest of the last word unused), words from n_ewords on hold the filter group (n_fpairs pairs, a
// multiple of 32: whole words of 4 batches). Item slot 2*pair + h of the kind's arrays belongs to pair `pair`.
struct FilterGroup {
    int n_epairs, n_ewords;
    int n_fpairs;
    const float* __restrict__ fx;    // [2 n_fpairs] centre - g, filter group only          (pads: 0)
    const float* __restrict__ fy;
    const float* __restrict__ fz;
    const float* __restrict__ fk;    // [2 n_fpairs] |centre - g|^2 - r^2                   (pads: 3e38)
    const float4* __restrict__ fq;   // spheres only: [2 * all pairs] per pair (cx0,cx1,cy0,cy1), (cz0,cz1,r2_0,r2_1); pads NaN
    const float* __restrict__ bnd;   // [20 per filter word] batch balls of the word's 4 batches: x0..x3, y0..y3, z0..z3 (centre - g),
                                     // k0..k3 (|centre - g|^2 - rho^2), 2 rho 0..3; empty batch: k = 3e38
    float gx, gy, gz;                // reference point of the filter group
    float cmax, r2max;               // max |centre - g| over items and balls (rounded up), max r^2 over the filter group
    float rhomax;                    // max ball radius rho (rounded up)
    float margin_k;                  // M = margin_k * a * (S^2 + So^2 + r2max)
};
constexpr float kMarginKTri = 160.0f * 5.9604644775390625e-08f;   // + the FP32 hit point's rounding (ray_amd_tri.h)

// Diagnostic build only (-DRT_STAMPS): cycles per section of the render loop (s_memtime, wave-uniform)
// and wave-level pass counts of the divergent loops. Never compiled into the product library.
#ifdef RT_STAMPS
struct Stamps {
    unsigned long long t[8];     // 0 assign+new sample, 1 -, 2 filter, 3 candidate walk, 4 rect+tri, 5 shade, 6 unwind+store, 7 other
    unsigned long long last;
    unsigned int cnt[4];         // 0 walk passes and 3 scans (first active lane only); 1 flagged sphere pairs and 2 sphere candidates (every lane)
};
#define RT_STAMP(i) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); stp.t[i] += now_ - stp.last; stp.last = now_; } while (0)
#define RT_COUNT(i) do { if ((__ffsll((long long)__ballot(1)) - 1) == int(threadIdx.x & 63)) ++stp.cnt[i]; } while (0)
#define RT_COUNT_LANES(i) do { ++stp.cnt[i]; } while (0)
#define RT_STAMP_PARAM , Stamps& stp
#define RT_STAMP_ARG , stp
#else
#define RT_COUNT_LANES(i) do {} while (0)
#define RT_STAMP(i) do {} while (0)
#define RT_COUNT(i) do {} while (0)
#define RT_STAMP_PARAM
#define RT_STAMP_ARG
#endif

RT_DEV bool reject_t(double t, double t_min, double bt, int idx, int bi) {
    return t < t_min || bt < t || (t == bt && idx < bi);
}

// Best hit so far of one ray. Either exact (bt, idx as the reference's closest_so_far) or deferred
// (dD >= 0): candidate `idx` with b = db, d = dD, whose accepted root is known to lie in [blo, bhi],
// inside [t_min, t_max] and below everything seen before, but has not been computed in FP64 yet.
struct HitState {
    double bt;              // exact best t (valid when dD < 0)
    int idx;                // flattened primitive index of the best, -1 = none
    float blo, bhi;         // bounds of the best's t (exact: bt rounded down / up)
    float db, dD;           // deferred candidate's b and d (shapes.h:83-86); dD = -1: the best is exact
};
__device__ __forceinline__ void hit_state_init(HitState& h, double t_max) {
    h.bt = t_max; h.idx = -1; h.blo = __double2float_rd(t_max); h.bhi = __double2float_ru(t_max);
    h.db = 0.0f; h.dD = -1.0f;
}
// sphere::hit's root selection (shapes.h:90-97) in FP64 against the exact best
__device__ __forceinline__ void exact_accept(float a, float b, float D, int idx, double t_min, HitState& h) {
    double sqrtd = sqrt64(double(D));
    double nb = double(-b);
    double root = (nb - sqrtd) / double(a);
    if (reject_t(root, t_min, h.bt, idx, h.idx)) {
        root = (nb + sqrtd) / double(a);
        if (reject_t(root, t_min, h.bt, idx, h.idx)) return;
    }
    h.bt = root; h.idx = idx;
}
// FP64 work of one ray: first the deferred candidate (it replaces whatever exact best was there: it is
// known to be closer and inside [t_min, t_max]), then, if with_c, candidate c against the exact best.
__device__ __forceinline__ void resolve_exact(float a, double t_min, double t_max, HitState& h, bool with_c, float cb, float cD, int cidx) {
    int n = (h.dD >= 0.0f ? 1 : 0) + (with_c ? 1 : 0);
    while (n > 0) {
        const bool do_deferred = h.dD >= 0.0f;
        const float eb = do_deferred ? h.db : cb, eD = do_deferred ? h.dD : cD;
        const int eidx = do_deferred ? h.idx : cidx;
        if (do_deferred) { h.bt = t_max; h.idx = -1; h.dD = -1.0f; }
        exact_accept(a, eb, eD, eidx, t_min, h);
        --n;
    }
    h.blo = __double2float_rd(h.bt); h.bhi = __double2float_ru(h.bt);
}

// A sphere whose reference discriminant D is >= 0 (a > 0, everything finite and in range). Straight-line
// FP32 code with selects; FP64 only behind the `need_exact` branch.
__device__ __forceinline__ void sphere_candidate(float a, float ia, float b, float D, int idx, double t_min, double t_max, float tmin_lo,
                                                 float tmin_hi, HitState& h RT_STAMP_PARAM) {
    // FP32 enclosures of the reference's two roots (-b -+ sqrt(d)) / a. sqrt(d) = sf + sl to ~1e-13
    // relative (sf: v_sqrt_f32, 1 ulp; sl: first-order correction from the exact residual d - sf*sf), so a
    // root that is a small difference of large terms (a ray leaving the surface of a huge sphere) still
    // gets bounds relative to its own size: e = |root| * 1e-6 (true error < 8 ulp) + (|b| + sf)/a * 1e-12.
    const float nb = -b;
    const float sf = __builtin_amdgcn_sqrtf(D);
    const float sl = sf > 0.0f ? __builtin_fmaf(-sf, sf, D) * (0.5f * __builtin_amdgcn_rcpf(sf)) : 0.0f;
    const float r1 = ((nb - sf) - sl) * ia, r2 = ((nb + sf) + sl) * ia;   // near and far root
    const float eabs = (__builtin_fabsf(nb) + sf) * ia * 1.0e-12f;
    const float e1 = __builtin_fabsf(r1) * 1.0e-6f + eabs, e2 = __builtin_fabsf(r2) * 1.0e-6f + eabs;
    const bool near_ok = r1 - e1 > tmin_hi;                      // near root certainly >= t_min: it is the one tested
    const bool near_below = r1 + e1 < tmin_lo;                   // near root certainly < t_min: the far root is tried
    const bool far_ok = near_below && (r2 - e2 > tmin_hi);
    const bool miss = near_below && (r2 + e2 < tmin_lo);         // both roots < t_min
    const float lo = near_ok ? r1 - e1 : r2 - e2, hi = near_ok ? r1 + e1 : r2 + e2;
    const bool known = near_ok || far_ok;                        // which root the reference accepts is certain
    const bool skip = miss || (known && lo > h.bhi);             // no hit, or behind the best so far
    const bool take = known && !skip && hi < h.blo;              // in front of the best: new best, FP64 deferred
    h.idx = take ? idx : h.idx; h.db = take ? b : h.db; h.dD = take ? D : h.dD;
    h.blo = take ? lo : h.blo; h.bhi = take ? hi : h.bhi;
    RT_COUNT_LANES(2);
    if (!skip && !take) {                                        // overlap, tie, or a root within rounding of t_min
        resolve_exact(a, t_min, t_max, h, true, b, D, idx);
    }
}

RT_DEV bool ray_in_fast_range(V3 o, V3 d, float a) {
    return __builtin_fabsf(o.x) <= kMaxCoord && __builtin_fabsf(o.y) <= kMaxCoord && __builtin_fabsf(o.z) <= kMaxCoord &&
           __builtin_fabsf(d.x) <= kMaxCoord && __builtin_fabsf(d.y) <= kMaxCoord && __builtin_fabsf(d.z) <= kMaxCoord &&
           a >= kMinA && a <= kMaxA;
}

// Filter constants of one ray for one group (DESIGN.md "sphere filter", "batch bounds").
struct FilterRay {
    f2 DX, DY, DZ, OD, MM, NA, WX, WY, WZ;
    f2 MB, AM, SE;   // level 0: margin term, a * m, -(s a + margin) (SE = -3e38: no behind test)
};
// behind_ok: t_min >= 0, so a batch whose ball lies wholly behind the origin cannot hold a hit
__device__ __forceinline__ FilterRay filter_ray(const FilterGroup& fg, V3 o, V3 d, float a, float ia, bool behind_ok) {
    if (fg.n_fpairs == 0) {   // wave-uniform: only always-flagged pairs, the constants are not read
        FilterRay z; z.OD = z.MM = z.NA = z.WX = z.WY = z.WZ = z.MB = z.AM = z.SE = f2{0.0f, 0.0f};
        z.DX = f2{d.x, d.x}; z.DY = f2{d.y, d.y}; z.DZ = f2{d.z, d.z};
        return z;
    }
    V3 orel = o - v3(fg.gx, fg.gy, fg.gz);
    float s = -(orel.x * d.x + orel.y * d.y + orel.z * d.z) * ia;
    V3 op = v3(__builtin_fmaf(s, d.x, orel.x), __builtin_fmaf(s, d.y, orel.y), __builtin_fmaf(s, d.z, orel.z));   // a point of the line near g
    float od = __builtin_fmaf(op.z, d.z, __builtin_fmaf(op.y, d.y, op.x * d.x));
    float oo = __builtin_fmaf(op.z, op.z, __builtin_fmaf(op.y, op.y, op.x * op.x));
    float S = __builtin_amdgcn_sqrtf(oo) * 1.001f + fg.cmax;
    float So = __builtin_amdgcn_sqrtf(length_squared(orel)) * 1.001f + fg.cmax;
    float SS = S * S + So * So;
    float M = fg.margin_k * a * (SS + fg.r2max);
    float m2 = M - a * oo, wa = 2.0f * a;
    // level 0: a flagged item's centre is within r + m of the line, m = sqrt(M / a); the ball test runs with radius rho + m
    float m = __builtin_amdgcn_sqrtf(M * ia) * 1.001f + 1.0e-30f;
    float Rm = fg.rhomax + m;
    float Mb = fg.margin_k * a * (SS + Rm * Rm);
    float am = a * m;
    float mb = (Mb - a * oo) + am * m;
    float se = behind_ok ? -(s * a + __builtin_amdgcn_sqrtf(Mb) * 1.001f) : -3.0e38f;
    FilterRay fr;
    fr.DX = f2{d.x, d.x}; fr.DY = f2{d.y, d.y}; fr.DZ = f2{d.z, d.z}; fr.OD = f2{od, od}; fr.MM = f2{m2, m2}; fr.NA = f2{-a, -a};
    fr.WX = f2{wa * op.x, wa * op.x}; fr.WY = f2{wa * op.y, wa * op.y}; fr.WZ = f2{wa * op.z, wa * op.z};
    fr.MB = f2{mb, mb}; fr.AM = f2{am, am}; fr.SE = f2{se, se};
    return fr;
}

// Level 0 + 1 for words [w0, w0 + nw) of a group: one flag bit per pair (pair p of a word <-> bit 31 - p),
// words to this thread's LDS column; returns the mask of non-empty words.
__device__ __forceinline__ uint32_t filter_words(const FilterGroup& fg, const FilterRay& fr, int w0, int nw, uint32_t* words) {
    cf2p fx = RT_CONST_F2(fg.fx), fy = RT_CONST_F2(fg.fy), fz = RT_CONST_F2(fg.fz), fk = RT_CONST_F2(fg.fk);
    uint32_t summary = 0;
    for (int wi = 0; wi < nw; ++wi) {
        const int gw = w0 + wi;
        uint32_t flagged;
        if (gw < fg.n_ewords) {                                // exact group: every real pair
            const int real = fg.n_epairs - gw * 32;
            flagged = real >= 32 ? 0xffffffffu : ~(0xffffffffu >> real);
        } else {
            const int fw = gw - fg.n_ewords;
            // level 0: the word's 4 batch balls, two per packed operation. Sign bit of sk[b] set: this lane skips batch b.
            cf2p bn = RT_CONST_F2(fg.bnd) + fw * 10;
            uint32_t sk[4];
            uint32_t act = 0;                                  // wave-uniform: batches that some lane needs
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 X = bn[h], Y = bn[2 + h], Z = bn[4 + h], K = bn[6 + h], R2 = bn[8 + h];
                const f2 bq = __builtin_elementwise_fma(-Z, fr.DZ, __builtin_elementwise_fma(-Y, fr.DY, __builtin_elementwise_fma(-X, fr.DX, fr.OD)));
                f2 Q = __builtin_elementwise_fma(fr.NA, K, fr.MB);
                Q = __builtin_elementwise_fma(fr.AM, R2, Q);
                Q = __builtin_elementwise_fma(fr.WX, X, Q);
                Q = __builtin_elementwise_fma(fr.WY, Y, Q);
                Q = __builtin_elementwise_fma(fr.WZ, Z, Q);
                const f2 T = __builtin_elementwise_fma(bq, bq, Q);     // < 0: the line misses the widened ball
                const f2 e = bq + fr.SE;                                // > 0: the ball's centre is behind the origin by the margin
                const f2 V = __builtin_elementwise_fma(e, e, -T);       // >= 0 with e >= 0: the ball ends before the origin
                sk[2 * h] = __float_as_uint(T.x) | ~(__float_as_uint(e.x) | __float_as_uint(V.x));
                sk[2 * h + 1] = __float_as_uint(T.y) | ~(__float_as_uint(e.y) | __float_as_uint(V.y));
                act |= (__ballot(int(sk[2 * h]) >= 0) != 0ull ? 1u : 0u) << (2 * h);
                act |= (__ballot(int(sk[2 * h + 1]) >= 0) != 0ull ? 1u : 0u) << (2 * h + 1);
            }
            uint32_t m = 0;                                    // bit set: pair NOT flagged
            if (act == 0u) {
                m = 0xffffffffu;
            } else {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    if ((act >> bb & 1u) == 0u) { m = (m << 8) | 0xffu; continue; }
                    const int batch = fw * 4 + bb;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = batch * 8 + jj;
                        f2 X = fx[j], Y = fy[j], Z = fz[j], K = fk[j];
                        f2 bq = __builtin_elementwise_fma(-Z, fr.DZ, __builtin_elementwise_fma(-Y, fr.DY, __builtin_elementwise_fma(-X, fr.DX, fr.OD)));
                        f2 Q = __builtin_elementwise_fma(fr.NA, K, fr.MM);
                        Q = __builtin_elementwise_fma(fr.WX, X, Q);
                        Q = __builtin_elementwise_fma(fr.WY, Y, Q);
                        Q = __builtin_elementwise_fma(fr.WZ, Z, Q);
                        f2 T = __builtin_elementwise_fma(bq, bq, Q);   // >= 0 (sign bit clear): flagged
                        m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x) & __float_as_uint(T.y), 31);
                    }
                    m |= uint32_t(int(sk[bb]) >> 31) & 0xffu;          // this lane does not need the batch
                }
            }
            flagged = ~m;
        }
        words[wi * kScanBlock] = flagged;
        summary |= (flagged != 0u ? 1u : 0u) << wi;
    }
    return summary;
}

// Level 2 driver: each lane walks its own flagged pairs of the chunk and calls visit(pair).
template <class Visit>
__device__ __forceinline__ void walk_flagged(uint32_t summary, int w0, const uint32_t* words, Visit&& visit RT_STAMP_PARAM) {
    uint32_t cur = 0;
    int wbase = 0;
    while (cur != 0u || summary != 0u) {
        RT_COUNT(0);
        if (cur == 0u) {
            const int wi = __builtin_ctz(summary);
            summary &= summary - 1u;
            cur = words[wi * kScanBlock];
            wbase = (w0 + wi) * 32;
        }
        const int p = __builtin_clz(cur);
        cur &= ~(0x80000000u >> p);
        visit(wbase + p);
    }
}

__device__ __forceinline__ int group_words(const FilterGroup& fg) { return fg.n_ewords + (fg.n_fpairs >> 5); }

// Spheres of one ray (o, d, a = d.lengthSquared() as Ray.h:8 caches it). `words` is this thread's
// column of the block's flag-word array in LDS (stride kScanBlock). sph_prim: flattened primitive
// index per sphere slot (-1 for pads). Requires ray_in_fast_range and a finite scene. On return h is exact.
__device__ __forceinline__ void scan_spheres(const FilterGroup& sg, const int* __restrict__ sph_prim, V3 o, V3 d, float a, double t_min,
                                             double t_max, HitState& h, uint32_t* words RT_STAMP_PARAM) {
    const int n_words = group_words(sg);
    if (n_words == 0) return;
    const float tmin_lo = __double2float_rd(t_min), tmin_hi = __double2float_ru(t_min);
    const float ia = __builtin_amdgcn_rcpf(a);
    const FilterRay fr = filter_ray(sg, o, d, a, ia, t_min >= 0.0);
    const f2 OX = {o.x, o.x}, OY = {o.y, o.y}, OZ = {o.z, o.z}, AA = {a, a};
    for (int w0 = 0; w0 < n_words; w0 += kMaskWords) {
        const int nw = n_words - w0 < kMaskWords ? n_words - w0 : kMaskWords;
        const uint32_t summary = filter_words(sg, fr, w0, nw, words);
        RT_STAMP(2);
        walk_flagged(summary, w0, words, [&](int pair) {
            RT_COUNT_LANES(1);
            const float4 A = sg.fq[2 * pair], B = sg.fq[2 * pair + 1];
            const f2 cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, r2 = {B.z, B.w};
            const f2 x = OX - cx, y = OY - cy, z = OZ - cz;
            const f2 b = x * fr.DX + y * fr.DY + z * fr.DZ;   // dotProduct(originCenter, direction), shapes.h:83
            const f2 c = x * x + y * y + z * z - r2;          // lengthSquared() - radiusSquared, shapes.h:84
            const f2 D = b * b - AA * c;                      // shapes.h:86
            uint32_t pend = (D.x >= 0.0f ? 1u : 0u) | (D.y >= 0.0f ? 2u : 0u);   // mostly 0 or one bit
            while (pend != 0u) {
                const bool second = (pend & 1u) == 0u;
                pend &= pend - 1u;
                sphere_candidate(a, ia, second ? b.y : b.x, second ? D.y : D.x, sph_prim[2 * pair + (second ? 1 : 0)], t_min, t_max, tmin_lo,
                                 tmin_hi, h RT_STAMP_ARG);
            }
        } RT_STAMP_ARG);
        RT_STAMP(3);
    }
    if (h.dD >= 0.0f) { resolve_exact(a, t_min, t_max, h, false, 0.0f, 0.0f, -1); }
    RT_STAMP(3);
}

}  // namespace rt
