// Closest-hit scan over the sphere arrays for gfx950 (hitFromList, hitPosition.cpp:6-22, over
// sphere::hit, shapes.h:77-97). The contract is the brute-force loop's: every sphere is a candidate for
// every ray and the result is bit-identical to the reference's arithmetic. Three levels decide, with
// growing cost, which spheres a ray still has to be tested against exactly:
//  0. BOXES: the filter group is stored in spatially compact batches of 8 pairs (16 items), each with an
//     axis-aligned box around its items; words of 4 batches and chunks of words have boxes too. A slab
//     test of the ray against a box, widened by the ray's filter margin, tells per lane whether the ray
//     can come near any item inside [max(t_min, 0), best t so far]. A batch (word, chunk) that no lane
//     of the wave needs is not filtered at all (wave-uniform branch); a lane that does not need a batch
//     gets no flags from it. No item that the reference could hit is ever dropped.
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
// every ray; rt_upload_scene puts them in the "exact group", whose pairs are always visited, and first,
// so that the ground's hit already bounds t when the boxes are tested.
#pragma once
#include "rt_device.hpp"

namespace rt {

typedef float f2 __attribute__((ext_vector_type(2)));
// constant address space: a load through it with a wave-uniform address is a scalar load (s_load)
typedef const f2 __attribute__((address_space(4)))* cf2p;
#define RT_CONST_F2(p) ((rt::cf2p)(uintptr_t)(p))

#ifndef RT_SHARE_UNROLL
#define RT_SHARE_UNROLL 8          // pairs per trip of the shared filter's loop over a batch (more: more loads in flight, more VGPRs)
#endif
constexpr int kMaskWords = 16;                 // flag words per lane and chunk (32 pairs each), kept in LDS
constexpr int kScanBlock = 256;                // threads per block of every kernel that scans
// Groups with one flag per ITEM (triangles): chunks of kItemChunkWords filter words, item 0's flags in the first kItemChunkWords
// words of the lane's column, item 1's in the next.
#ifdef RT_EXPERIMENT_TRI_STAGED
// Experiment (tools/build_exp_tri_staged.sh): chunks of 6, and rows 12..15 of the wave's 64 columns (4 x 256 B) are
// its staging rows for the filter data of one word's 4 batches (filter_words_staged)
constexpr int kItemChunkWords = 6, kStageWord = 12;
#else
#ifndef RT_ITEM_CHUNK_WORDS
#define RT_ITEM_CHUNK_WORDS 8
#endif
constexpr int kItemChunkWords = RT_ITEM_CHUNK_WORDS;
#endif
// scan_triangles: the wave's queue of flagged items is worked off after a chunk once it holds this many entries (one pass's worth)
#ifndef RT_ITEM_FLUSH
#define RT_ITEM_FLUSH 64
#endif
constexpr int kItemFlush = RT_ITEM_FLUSH;
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
    const float* __restrict__ fi;    // [8 n_fpairs] filter group only, per pair x0 x1 y0 y1 z0 z1 k0 k1 (256 B per batch, one base pointer):
                                     // x, y, z = centre - g (pads: 0), k = |centre - g|^2 - r^2 (pads: 3e38)
    const float4* __restrict__ fq;   // spheres only: [2 * all pairs] per pair (cx0,cx1,cy0,cy1), (cz0,cz1,r2_0,r2_1); pads NaN
    const float* __restrict__ bnd;   // [24 per filter word] boxes of the word's 4 batches, relative to g, item radii included, as centre C
                                     // and half extent E: C x0..x3, C y0..y3, C z0..z3, E x0..x3, E y0..y3, E z0..z3
    const float* __restrict__ wbnd;  // [12 per pair of filter words] box around each word's batches: C x (word 0, word 1), C y, C z, E x, E y, E z
    const float* __restrict__ cbnd;  // [12 per pair of chunks] the same around each chunk of chunk_words filter words
    int chunk_words;                 // filter words per chunk: kMaskWords (one flag per pair) or kItemChunkWords (one per item)
    int n_fbatches;                  // batches that hold items (the rest of the last word is padding)
    float gx, gy, gz;                // reference point of the filter group
    float cmax, r2max;               // max |centre - g| (rounded up), max r^2 over the filter group
    float margin_k;                  // M = margin_k * a * (S^2 + So^2 + r2max)
#ifdef RT_EXPERIMENT_SPHERE_BVH
    // Experiment (tools/build_exp_sphere_bvh.sh): every lane walks a BVH over the sphere PAIRS of the filter
    // group instead of the wave-level box / filter levels. Nodes as in scan_triangles_bvh; a leaf ~p is pair p.
    const float4* __restrict__ sbvh;
    float sbvh_extent;
#endif
};
constexpr float kMarginKTri = 160.0f * 5.9604644775390625e-08f;   // + the FP32 hit point's rounding (ray_amd_tri.h)

// Diagnostic build only (-DRT_STAMPS): cycles per section of the render loop (s_memtime, wave-uniform)
// and wave-level pass counts of the divergent loops. Never compiled into the product library.
#ifdef RT_STAMPS
struct Stamps {
    unsigned long long t[8];     // 0 assign+new sample, 1 (count, not cycles) batches filtered, 2 filter, 3 candidate walk, 4 rect+tri, 5 shade, 6 unwind+store, 7 other
    unsigned long long last;
    unsigned int cnt[4];         // 0 walk passes and 3 scans (first active lane only); 1 flagged sphere pairs and 2 sphere candidates (every lane)
};
#define RT_STAMP(i) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); stp.t[i] += now_ - stp.last; stp.last = now_; } while (0)
#define RT_COUNT(i) do { if ((__ffsll((long long)__ballot(1)) - 1) == int(threadIdx.x & 63)) ++stp.cnt[i]; } while (0)
#define RT_COUNT_LANES(i) do { ++stp.cnt[i]; } while (0)
#define RT_STAMP_PARAM , Stamps& stp
#define RT_EVENT(i) do { stp.t[i] += 1; } while (0)   /* wave-uniform event counts in the unused time slots */
#define RT_BATCHES(act) do { stp.t[1] += __builtin_popcount(act); } while (0)
#define RT_STAMP_ARG , stp
#else
#define RT_COUNT_LANES(i) do {} while (0)
#define RT_EVENT(i) do {} while (0)
#define RT_BATCHES(act) do {} while (0)
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
    // kMinA <= a <= kMaxA and |b|, sqrt(D) finite floats (ray_in_fast_range): the shared-reciprocal division is `/` here
    double sqrtd = sqrt64(double(D));
    double nb = double(-b);
    const Recip64 ra = recip64(double(a));
    double root = div64(nb - sqrtd, ra);
    if (reject_t(root, t_min, h.bt, idx, h.idx)) {
        root = div64(nb + sqrtd, ra);
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

// v_pk_fma_f32 with one half of a register pair as the multiplier of both results (op_sel), so that two
// different per-ray constants share a pair instead of each being duplicated: r = c[SC] * x + a (cv) or
// r = c[SC] * x + a[SA] (cc). x: wave-uniform data in an SGPR pair.
template <int SC>
__device__ __forceinline__ f2 pk_fma_cv(f2 c, f2 x, f2 a) {
    f2 r;
    if (SC == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(c), "s"(x), "v"(a));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(c), "s"(x), "v"(a));
    return r;
}
template <int SC, int SA>
__device__ __forceinline__ f2 pk_fma_cc(f2 c, f2 x, f2 a) {
    f2 r;
    if (SC == 0 && SA == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(c), "s"(x), "v"(a));
    else if (SC == 0 && SA == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(c), "s"(x), "v"(a));
    else if (SC == 1 && SA == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(c), "s"(x), "v"(a));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(c), "s"(x), "v"(a));
    return r;
}

// min / max on results of the above (asm: the compiler would first canonicalise values it did not compute itself)
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// the same for v_pk_mul_f32 / v_pk_add_f32 with per-lane data x: r = c[SC] * x, r = c[SC] - x
template <int SC>
__device__ __forceinline__ f2 pk_mul_c(f2 c, f2 x) {
    f2 r;
    if (SC == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(c), "v"(x));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(c), "v"(x));
    return r;
}
template <int SC>
__device__ __forceinline__ f2 pk_sub_c(f2 c, f2 x) {
    f2 r;
    if (SC == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(c), "v"(x));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(c), "v"(x));
    return r;
}

// Filter constants of one ray for one group (DESIGN.md "sphere filter", "batch boxes").
struct FilterRay {
    // level 1: T = bq^2 + Q, bq = od - c.d, Q = (M - a |op|^2) - a k + 2 a op.c
    f2 D01, D2O, NAM, W01, W2;   // (-d.x, -d.y), (-d.z, od), (-a, M - a |op|^2), (2a op.x, 2a op.y), (2a op.z, -)
    // level 0 (boxes as centre C and half extent E): per axis u = inv C + k, w = |inv| E + mw; the slab is [u - w, u + w]
    f2 I01, I2, A01, A2;         // (inv.x, inv.y), (inv.z, -), (|inv.x|, |inv.y|), (|inv.z|, -)
    f2 KX, KY, KZ;               // per axis (k, mw) = (-o inv, m |inv|): o the origin relative to the boxes' reference point, m the widening
    float t_lo;
};
// Slab-test constants (box_pair) of the ray o + t d, o given relative to the boxes' reference point: the boxes are
// widened by m; direction components below 1e-7 |d| are clamped
__device__ __forceinline__ void set_box_ray(FilterRay& fr, V3 orel, V3 d, float a, float m, double t_min) {
    const float dmin = 1.0e-7f * __builtin_amdgcn_sqrtf(a);
    const float ix = __builtin_amdgcn_rcpf(__builtin_copysignf(__builtin_fmaxf(__builtin_fabsf(d.x), dmin), d.x));
    const float iy = __builtin_amdgcn_rcpf(__builtin_copysignf(__builtin_fmaxf(__builtin_fabsf(d.y), dmin), d.y));
    const float iz = __builtin_amdgcn_rcpf(__builtin_copysignf(__builtin_fmaxf(__builtin_fabsf(d.z), dmin), d.z));
    const float ax = __builtin_fabsf(ix), ay = __builtin_fabsf(iy), az = __builtin_fabsf(iz);
    fr.I01 = f2{ix, iy}; fr.I2 = f2{iz, 0.0f};
    fr.A01 = f2{ax, ay}; fr.A2 = f2{az, 0.0f};
    fr.KX = f2{-orel.x * ix, m * ax};
    fr.KY = f2{-orel.y * iy, m * ay};
    fr.KZ = f2{-orel.z * iz, m * az};
    fr.t_lo = t_min >= 0.0 ? 0.0f : -__builtin_inff();
}
__device__ __forceinline__ FilterRay filter_ray(const FilterGroup& fg, V3 o, V3 d, float a, float ia, double t_min) {
    FilterRay fr;
    if (fg.n_fpairs == 0) {   // wave-uniform: only always-flagged pairs, the constants are not read
        fr.D01 = fr.D2O = fr.NAM = fr.W01 = fr.W2 = fr.I01 = fr.I2 = fr.A01 = fr.A2 = fr.KX = fr.KY = fr.KZ = f2{0.0f, 0.0f};
        fr.t_lo = 0.0f;
        return fr;
    }
    V3 orel = o - v3(fg.gx, fg.gy, fg.gz);
    float s = -(orel.x * d.x + orel.y * d.y + orel.z * d.z) * ia;
    V3 op = v3(__builtin_fmaf(s, d.x, orel.x), __builtin_fmaf(s, d.y, orel.y), __builtin_fmaf(s, d.z, orel.z));   // a point of the line near g
    float od = __builtin_fmaf(op.z, d.z, __builtin_fmaf(op.y, d.y, op.x * d.x));
    float oo = __builtin_fmaf(op.z, op.z, __builtin_fmaf(op.y, op.y, op.x * op.x));
    float S = __builtin_amdgcn_sqrtf(oo) * 1.001f + fg.cmax;
    float So = __builtin_amdgcn_sqrtf(length_squared(orel)) * 1.001f + fg.cmax;
    float M = fg.margin_k * a * (S * S + So * So + fg.r2max);
    float m2 = M - a * oo, wa = 2.0f * a;
    fr.D01 = f2{-d.x, -d.y}; fr.D2O = f2{-d.z, od}; fr.NAM = f2{-a, m2};
    fr.W01 = f2{wa * op.x, wa * op.y}; fr.W2 = f2{wa * op.z, 0.0f};
    // level 0 (DESIGN.md "batch boxes"): a flagged item's centre is within r + m of the line, m = sqrt(M / a); the
    // boxes hold the items' radii and are widened by 1.01 m, which also covers the rounding of the slab
    // arithmetic and the clamp of direction components below 1e-7 |d|
    const float m = __builtin_amdgcn_sqrtf(M * ia) * 1.01f + 1.0e-30f;
    set_box_ray(fr, orel, d, a, m, t_min);
    return fr;
}

// Slab test of the ray's [t_lo, t_hi] against two widened boxes at once: n0 / n1 = this lane needs box 0 / 1. The boxes come as
// centre and half extent per axis (C - E, C + E enclose the box; rounded outwards by rt_upload_scene): per axis and box pair one
// packed FMA each for u = inv C - inv o and w = |inv| E + |inv| m, one packed subtraction and addition for the slab's two ends -
// no per-box min / max to sort them (v_min_f32 / v_max_f32 cost as much as a packed instruction on gfx950, tools/valu_bench).
// The rounding of these four operations (a few ulp of |u| + |w|) is covered by the widening m like that of the lo / hi form was.
__device__ __forceinline__ void box_pair(const FilterRay& fr, float t_hi, f2 CX, f2 CY, f2 CZ, f2 EX, f2 EY, f2 EZ, bool& n0, bool& n1) {
    const f2 ux = pk_fma_cc<0, 0>(fr.I01, CX, fr.KX), wx = pk_fma_cc<0, 1>(fr.A01, EX, fr.KX);
    const f2 uy = pk_fma_cc<1, 0>(fr.I01, CY, fr.KY), wy = pk_fma_cc<1, 1>(fr.A01, EY, fr.KY);
    const f2 uz = pk_fma_cc<0, 0>(fr.I2, CZ, fr.KZ), wz = pk_fma_cc<0, 1>(fr.A2, EZ, fr.KZ);
    const f2 nx = ux - wx, fx = ux + wx, ny = uy - wy, fy = uy + wy, nz = uz - wz, fz = uz + wz;
    n0 = vmax(vmax3(nx.x, ny.x, nz.x), fr.t_lo) <= vmin(vmin3(fx.x, fy.x, fz.x), t_hi);
    n1 = vmax(vmax3(nx.y, ny.y, nz.y), fr.t_lo) <= vmin(vmin3(fx.y, fy.y, fz.y), t_hi);
}

// The same slab test with the two boxes in VGPRs (every lane looks at its own BVH node) and the entry parameters returned.
template <int SC, int SA>
__device__ __forceinline__ f2 pk_fma_cc_v(f2 c, f2 x, f2 a) {
    f2 r;
    if (SC == 0 && SA == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(c), "v"(x), "v"(a));
    else if (SC == 0 && SA == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(c), "v"(x), "v"(a));
    else if (SC == 1 && SA == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(c), "v"(x), "v"(a));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(c), "v"(x), "v"(a));
    return r;
}
template <int SC>
__device__ __forceinline__ f2 pk_fma_cv_v(f2 c, f2 x, f2 a) {
    f2 r;
    if (SC == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(c), "v"(x), "v"(a));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(c), "v"(x), "v"(a));
    return r;
}
__device__ __forceinline__ void box_pair_v(const FilterRay& fr, float t_hi, f2 CX, f2 CY, f2 CZ, f2 EX, f2 EY, f2 EZ, bool& n0, bool& n1, float& e0,
                                           float& e1) {
    const f2 ux = pk_fma_cc_v<0, 0>(fr.I01, CX, fr.KX), wx = pk_fma_cc_v<0, 1>(fr.A01, EX, fr.KX);
    const f2 uy = pk_fma_cc_v<1, 0>(fr.I01, CY, fr.KY), wy = pk_fma_cc_v<1, 1>(fr.A01, EY, fr.KY);
    const f2 uz = pk_fma_cc_v<0, 0>(fr.I2, CZ, fr.KZ), wz = pk_fma_cc_v<0, 1>(fr.A2, EZ, fr.KZ);
    const f2 nx = ux - wx, fx = ux + wx, ny = uy - wy, fy = uy + wy, nz = uz - wz, fz = uz + wz;
    e0 = vmax(vmax3(nx.x, ny.x, nz.x), fr.t_lo);
    n0 = e0 <= vmin(vmin3(fx.x, fy.x, fz.x), t_hi);
    e1 = vmax(vmax3(nx.y, ny.y, nz.y), fr.t_lo);
    n1 = e1 <= vmin(vmin3(fx.y, fy.y, fz.y), t_hi);
}

// Level 0, chunks: does any lane of the wave need chunk ci (wave-uniform answer)? Groups of one chunk skip the test.
__device__ __forceinline__ bool chunk_needed(const FilterGroup& fg, const FilterRay& fr, float t_hi, int ci) {
    if ((fg.n_fpairs >> 5) <= fg.chunk_words) return true;
    cf2p cb = RT_CONST_F2(fg.cbnd) + (ci >> 1) * 6;
    bool n0, n1;
    box_pair(fr, t_hi, cb[0], cb[1], cb[2], cb[3], cb[4], cb[5], n0, n1);
    return __ballot((ci & 1) ? n1 : n0) != 0ull;
}

// Level 0 + 1 for filter words [fw0, fw0 + nw) of a group: one flag bit per pair (pair p of a word <-> bit
// 31 - p), words to this thread's LDS column; returns the mask of non-empty words. t_hi: upper bound of the
// best t so far (boxes wholly beyond it are not needed).
// ITEMS: one flag per ITEM instead (for kinds whose exact test is expensive per item: triangles): item 0's
// flags of word wi at words[wi], item 1's at words[wi + kItemChunkWords]; nw <= kItemChunkWords.
// Level 0 only, for filter words [fw0, fw0 + nw): on_word(fw, nd, real) is called for every word that some lane of the wave needs,
// with this lane's answer per batch (nd[bb]: the lane needs batch bb of the word) and the number of real batches of the word.
template <class OnWord>
__device__ __forceinline__ void filter_words_level0(const FilterGroup& fg, const FilterRay& fr, float t_hi, int fw0, int nw, OnWord&& on_word) {
    uint32_t wact = 0;                                     // wave-uniform: words that some lane needs
    for (int q = 0; 2 * q < nw; ++q) {
        cf2p wb = RT_CONST_F2(fg.wbnd) + (fw0 / 2 + q) * 6;
        bool n0, n1;
        box_pair(fr, t_hi, wb[0], wb[1], wb[2], wb[3], wb[4], wb[5], n0, n1);
        wact |= (__ballot(n0) != 0ull ? 1u : 0u) << (2 * q);
        wact |= (__ballot(n1) != 0ull && 2 * q + 1 < nw ? 1u : 0u) << (2 * q + 1);
    }
    for (int wi = 0; wi < nw; ++wi) {
        const int fw = fw0 + wi;
        cf2p bn = RT_CONST_F2(fg.bnd) + fw * 12;
        if ((wact >> wi & 1u) == 0u) continue;
        bool nd[4];
        box_pair(fr, t_hi, bn[0], bn[2], bn[4], bn[6], bn[8], bn[10], nd[0], nd[1]);
        box_pair(fr, t_hi, bn[1], bn[3], bn[5], bn[7], bn[9], bn[11], nd[2], nd[3]);
        on_word(fw, nd, fg.n_fbatches - fw * 4);
    }
}

// emit(wi, flags, flags1): called once per word wi that some lane of the wave needs, with this lane's flags (ITEMS: of item 0 and of item 1)
template <bool ITEMS, class Emit>
__device__ __forceinline__ void filter_words_emit(const FilterGroup& fg, const FilterRay& fr, float t_hi, int fw0, int nw, Emit&& emit RT_STAMP_PARAM) {
    cf2p fi = RT_CONST_F2(fg.fi);
    // level 0, words: the box around each word's 4 batches, two words per packed operation (fw0 is even)
    uint32_t wact = 0;                                     // wave-uniform: words that some lane needs
    for (int q = 0; 2 * q < nw; ++q) {
        cf2p wb = RT_CONST_F2(fg.wbnd) + (fw0 / 2 + q) * 6;
        bool n0, n1;
        box_pair(fr, t_hi, wb[0], wb[1], wb[2], wb[3], wb[4], wb[5], n0, n1);
        wact |= (__ballot(n0) != 0ull ? 1u : 0u) << (2 * q);
        wact |= (__ballot(n1) != 0ull && 2 * q + 1 < nw ? 1u : 0u) << (2 * q + 1);
    }
    for (int wi = 0; wi < nw; ++wi) {
        const int fw = fw0 + wi;
        // level 0, batches: the word's 4 batch boxes
        cf2p bn = RT_CONST_F2(fg.bnd) + fw * 12;
        if ((wact >> wi & 1u) == 0u) continue;           // no lane needs anything of this word (its summary bit stays 0)
        bool nd[4];                                        // this lane needs batch b
        box_pair(fr, t_hi, bn[0], bn[2], bn[4], bn[6], bn[8], bn[10], nd[0], nd[1]);
        box_pair(fr, t_hi, bn[1], bn[3], bn[5], bn[7], bn[9], bn[11], nd[2], nd[3]);
        const int real = fg.n_fbatches - fw * 4;               // padding batches hold no items
        uint32_t act = 0;                                      // wave-uniform: batches that some lane needs
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) act |= (__ballot(nd[bb]) != 0ull && bb < real ? 1u : 0u) << bb;
        RT_BATCHES(act);
        uint32_t m = 0xffffffffu, m1 = 0xffffffffu;            // bit set: pair (ITEMS: item 0, item 1) NOT flagged
        if (act != 0u) {
            m = 0; m1 = 0;
            // the word's 1 KB of filter data: one SGPR pair for its 4 batches (left to itself the compiler re-derives the address
            // per batch from a spilled 8-register tuple: 8 v_readlane per batch)
            cf2p fiw = fi + fw * 128;
#ifndef RT_NO_FIW_ASM
            asm volatile("" : "+s"(fiw));
#endif
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                if ((act >> bb & 1u) == 0u) { m = (m << 8) | 0xffu; if (ITEMS) m1 = (m1 << 8) | 0xffu; continue; }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = bb * 8 + jj;
                    const f2 X = fiw[4 * j], Y = fiw[4 * j + 1], Z = fiw[4 * j + 2], K = fiw[4 * j + 3];
                    f2 bq = pk_fma_cc<0, 1>(fr.D01, X, fr.D2O);
                    bq = pk_fma_cv<1>(fr.D01, Y, bq);
                    bq = pk_fma_cv<0>(fr.D2O, Z, bq);
                    f2 Q = pk_fma_cc<0, 1>(fr.NAM, K, fr.NAM);
                    Q = pk_fma_cv<0>(fr.W01, X, Q);
                    Q = pk_fma_cv<1>(fr.W01, Y, Q);
                    Q = pk_fma_cv<0>(fr.W2, Z, Q);
                    const f2 T = __builtin_elementwise_fma(bq, bq, Q);   // >= 0 (sign bit clear): flagged
                    if (ITEMS) {
                        m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x), 31);
                        m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(T.y), 31);
                    } else {
                        m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x) & __float_as_uint(T.y), 31);
                    }
                }
                m |= nd[bb] ? 0u : 0xffu;     // this lane does not need the batch
                if (ITEMS) m1 |= nd[bb] ? 0u : 0xffu;
            }
        }
        emit(wi, ~m, ~m1);
    }
}
template <bool ITEMS = false>
__device__ __forceinline__ uint32_t filter_words(const FilterGroup& fg, const FilterRay& fr, float t_hi, int fw0, int nw, uint32_t* words RT_STAMP_PARAM) {
    uint32_t summary = 0;
    filter_words_emit<ITEMS>(fg, fr, t_hi, fw0, nw, [&](int wi, uint32_t f0, uint32_t f1) {
        const uint32_t flagged = ITEMS ? (f0 | f1) : f0;
        words[wi * kScanBlock] = f0;
        if (ITEMS) words[(wi + kItemChunkWords) * kScanBlock] = f1;
        summary |= (flagged != 0u ? 1u : 0u) << wi;
    } RT_STAMP_ARG);
    return summary;
}

#ifdef RT_EXPERIMENT_TRI_STAGED
// filter_words<true> with the filter data of a word staged through LDS (BASELINE north_star, config 4: "primitive batches staged
// through LDS") instead of read through the scalar cache, which a triangle group of thousands of items does not fit (50 % hits
// on the 9.8k-triangle frame, profiles/r2_scalar_cache.txt): the wave's active lanes copy the 256 B of every batch that some lane
// needs into the wave's staging rows with one coalesced load each, and every lane then reads each pair's 32 B back with two
// broadcast reads. LDS operations of one wave complete in order, so the wave needs no barrier, only the compiler must keep the order.
// rank / nact: this lane's rank among the active lanes of the wave and their number (all 64 lanes in the steady state of
// render_kernel: one trip of the copy loop).
// MEASURED SLOWER than the scalar-cache path (9.8k triangles, 1080p, 64 spp: 86.9 ms copy-then-filter, 90.3 ms with the next word's
// loads in flight as below, against 82.5 ms for the scalar path with the same chunks of 6 and 78.3 ms with the product's chunks of 8;
// profiles/r2_exp_tri_staged.txt): an experiment build only.
__device__ __forceinline__ uint32_t filter_words_staged(const FilterGroup& fg, const FilterRay& fr, float t_hi, int fw0, int nw, uint32_t* words, int rank,
                                                        int nact RT_STAMP_PARAM) {
    uint32_t* stage = words - (threadIdx.x & 63) + kStageWord * kScanBlock;   // wave-uniform: batch bb of the word at stage + bb * kScanBlock
    const uint32_t* __restrict__ fi = reinterpret_cast<const uint32_t*>(fg.fi);
    uint32_t summary = 0;
    uint32_t wact = 0;
    for (int q = 0; 2 * q < nw; ++q) {
        cf2p wb = RT_CONST_F2(fg.wbnd) + (fw0 / 2 + q) * 6;
        bool n0, n1;
        box_pair(fr, t_hi, wb[0], wb[1], wb[2], wb[3], wb[4], wb[5], n0, n1);
        wact |= (__ballot(n0) != 0ull ? 1u : 0u) << (2 * q);
        wact |= (__ballot(n1) != 0ull && 2 * q + 1 < nw ? 1u : 0u) << (2 * q + 1);
    }
    // level 0, batches of word wi: which this lane needs (nd), which any lane needs (returned)
    auto batch_boxes = [&](int wi, bool (&nd)[4]) {
        const int fw = fw0 + wi;
        cf2p bn = RT_CONST_F2(fg.bnd) + fw * 12;
        box_pair(fr, t_hi, bn[0], bn[2], bn[4], bn[6], bn[8], bn[10], nd[0], nd[1]);
        box_pair(fr, t_hi, bn[1], bn[3], bn[5], bn[7], bn[9], bn[11], nd[2], nd[3]);
        const int real = fg.n_fbatches - fw * 4;
        uint32_t act = 0;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) act |= (__ballot(nd[bb]) != 0ull && bb < real ? 1u : 0u) << bb;
        RT_BATCHES(act);
        return act;
    };
    // software pipeline over the words that some lane needs: while word wi is filtered out of LDS, the loads of the next
    // needed word's batches are in flight (v); a wave that is not full copies without the pipeline (each lane several dwords)
    const bool full = nact == 64;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    auto issue = [&](int wi, uint32_t act) {
        const uint32_t* __restrict__ src = fi + size_t(fw0 + wi) * 256 + rank;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) if (act >> bb & 1u) v[bb] = src[bb * 64];
    };
    int wi = wact != 0u ? __builtin_ctz(wact) : nw;
    bool nd[4] = {false, false, false, false};
    uint32_t act = 0;
    if (wi < nw) { act = batch_boxes(wi, nd); if (full) issue(wi, act); }
    while (wi < nw) {
        const uint32_t rest = wact & ~((2u << wi) - 1u);
        const int wj = rest != 0u ? __builtin_ctz(rest) : nw;
        bool nd2[4] = {false, false, false, false};
        uint32_t act2 = 0;
        if (wj < nw) act2 = batch_boxes(wj, nd2);
        uint32_t m = 0xffffffffu, m1 = 0xffffffffu;            // bit set: item 0 / item 1 of the pair NOT flagged
        asm volatile("" ::: "memory");                         // the previous word's reads stay before these writes
        if (full) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) if (act >> bb & 1u) stage[bb * kScanBlock + rank] = v[bb];
        } else {
            const uint32_t* __restrict__ src = fi + size_t(fw0 + wi) * 256;
            for (int dw = rank; dw < 64; dw += nact) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) if (act >> bb & 1u) stage[bb * kScanBlock + dw] = src[bb * 64 + dw];
            }
        }
        asm volatile("" ::: "memory");                         // and the reads below after them
        if (full && wj < nw) issue(wj, act2);
        if (act != 0u) {
            m = 0; m1 = 0;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                if ((act >> bb & 1u) == 0u) { m = (m << 8) | 0xffu; m1 = (m1 << 8) | 0xffu; continue; }
                const float4* __restrict__ row = reinterpret_cast<const float4*>(stage + bb * kScanBlock);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float4 A = row[2 * jj], B = row[2 * jj + 1];
                    const f2 X = {A.x, A.y}, Y = {A.z, A.w}, Z = {B.x, B.y}, K = {B.z, B.w};
                    f2 bq = pk_fma_cc_v<0, 1>(fr.D01, X, fr.D2O);
                    bq = pk_fma_cv_v<1>(fr.D01, Y, bq);
                    bq = pk_fma_cv_v<0>(fr.D2O, Z, bq);
                    f2 Q = pk_fma_cc_v<0, 1>(fr.NAM, K, fr.NAM);
                    Q = pk_fma_cv_v<0>(fr.W01, X, Q);
                    Q = pk_fma_cv_v<1>(fr.W01, Y, Q);
                    Q = pk_fma_cv_v<0>(fr.W2, Z, Q);
                    const f2 T = __builtin_elementwise_fma(bq, bq, Q);   // >= 0 (sign bit clear): flagged
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x), 31);
                    m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(T.y), 31);
                }
                m |= nd[bb] ? 0u : 0xffu;                          // this lane does not need the batch
                m1 |= nd[bb] ? 0u : 0xffu;
            }
        }
        words[wi * kScanBlock] = ~m;
        words[(wi + kItemChunkWords) * kScanBlock] = ~m1;
        summary |= (~(m & m1) != 0u ? 1u : 0u) << wi;
        wi = wj; act = act2;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) nd[bb] = nd2[bb];
    }
    return summary;
}
#endif

// Level 2 driver: each lane walks its own flagged pairs of the chunk and calls visit(pair).
// w0: word index of the chunk in pair space.
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
// The same over item flags (filter_words<true>): calls visit(2 * pair + item) for every flagged item.
template <class Visit>
__device__ __forceinline__ void walk_flagged_items(uint32_t summary, int w0, const uint32_t* words, Visit&& visit RT_STAMP_PARAM) {
    uint32_t cur0 = 0, cur1 = 0;
    int wbase = 0;
    while ((cur0 | cur1) != 0u || summary != 0u) {
        RT_COUNT(0);
        if ((cur0 | cur1) == 0u) {
            const int wi = __builtin_ctz(summary);
            summary &= summary - 1u;
            cur0 = words[wi * kScanBlock];
            cur1 = words[(wi + kItemChunkWords) * kScanBlock];
            wbase = (w0 + wi) * 32;
        }
        const bool second = cur0 == 0u;                    // item 0 flags of the word first, then item 1
        const uint32_t c = second ? cur1 : cur0;
        const int p = __builtin_clz(c);
        const uint32_t rest = c & ~(0x80000000u >> p);
        if (second) cur1 = rest; else cur0 = rest;
        visit(2 * (wbase + p) + (second ? 1 : 0));
    }
}

// Spheres of one ray (o, d, a = d.lengthSquared() as Ray.h:8 caches it). `words` is this thread's
// column of the block's flag-word array in LDS (stride kScanBlock). sph_prim: flattened primitive
// index per sphere slot (-1 for pads). Requires ray_in_fast_range and a finite scene. On return h is exact.
// redo: set when the shared walk (RT_SPHERES_SHARED_WALK) cannot order two candidates; the caller then takes the list-order scan.
__device__ __forceinline__ void scan_spheres(const FilterGroup& sg, const int* __restrict__ sph_prim, V3 o, V3 d, float a, double t_min,
                                             double t_max, HitState& h, uint32_t* words, bool& redo RT_STAMP_PARAM) {
    (void)redo;
    if (sg.n_epairs == 0 && sg.n_fpairs == 0) return;
    const float tmin_lo = __double2float_rd(t_min), tmin_hi = __double2float_ru(t_min);
    const float ia = __builtin_amdgcn_rcpf(a);
    const f2 O01 = {o.x, o.y}, O2A = {o.z, a}, D01 = {d.x, d.y}, D2 = {d.z, d.z};
    auto visit = [&](int pair) {
        RT_COUNT_LANES(1);
        const float4 A = sg.fq[2 * pair], B = sg.fq[2 * pair + 1];
        const f2 cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, r2 = {B.z, B.w};
        const f2 x = pk_sub_c<0>(O01, cx), y = pk_sub_c<1>(O01, cy), z = pk_sub_c<0>(O2A, cz);
        const f2 b = pk_mul_c<0>(D01, x) + pk_mul_c<1>(D01, y) + pk_mul_c<0>(D2, z);   // dotProduct(originCenter, direction), shapes.h:83
        const f2 c = x * x + y * y + z * z - r2;          // lengthSquared() - radiusSquared, shapes.h:84
        const f2 D = b * b - pk_mul_c<1>(O2A, c);         // shapes.h:86
        uint32_t pend = (D.x >= 0.0f ? 1u : 0u) | (D.y >= 0.0f ? 2u : 0u);   // mostly 0 or one bit
        while (pend != 0u) {
            const bool second = (pend & 1u) == 0u;
            pend &= pend - 1u;
            sphere_candidate(a, ia, second ? b.y : b.x, second ? D.y : D.x, sph_prim[2 * pair + (second ? 1 : 0)], t_min, t_max, tmin_lo,
                             tmin_hi, h RT_STAMP_ARG);
        }
    };
    for (int p = 0; p < sg.n_epairs; ++p) visit(p);       // exact group first: its hit bounds t for the boxes
    RT_STAMP(3);
    if (sg.n_fpairs == 0) {
        if (h.dD >= 0.0f) { resolve_exact(a, t_min, t_max, h, false, 0.0f, 0.0f, -1); }
        return;
    }
#ifdef RT_EXPERIMENT_SPHERE_BVH
    if (sg.sbvh != nullptr) {
        // Conservative boxes: a pair's box encloses both spheres; the reference accepts a root only if the line passes
        // within sqrt(r^2 + 23 u (|oc|^2 + r^2)) of the centre (|D_ref - D*| <= 23 u a (|oc|^2 + r^2)), i.e. within
        // r + 1.17e-3 |oc| with |oc| <= 1.74 (|o - g|_inf + extent); widened by m = 2.1e-3 (|o - g|_inf + extent), which also covers the
        // slab arithmetic. The exact group (the ground sphere) has been visited above and is not in the tree: its size would set the margin.
        const V3 orel = o - v3(sg.gx, sg.gy, sg.gz);
        const float omax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(orel.x), __builtin_fabsf(orel.y)), __builtin_fabsf(orel.z));
        FilterRay br;
        set_box_ray(br, orel, d, a, 2.1e-3f * (omax + sg.sbvh_extent) + 1.0e-30f, t_min);
        int node = 0, sp = 0;
        for (;;) {
            const float4* __restrict__ q = sg.sbvh + 4 * node;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            bool n0, n1;
            float e0, e1;
            box_pair_v(br, h.bhi, f2{q0.x, q0.y}, f2{q0.z, q0.w}, f2{q1.x, q1.y}, f2{q1.z, q1.w}, f2{q2.x, q2.y}, f2{q2.z, q2.w}, n0, n1, e0, e1);
            const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            if (n0 && c0 < 0) { visit(~c0); n0 = false; }
            if (n1 && c1 < 0) { visit(~c1); n1 = false; }
            if (n0 && n1) {
                const bool first0 = e0 <= e1;
                words[sp * kScanBlock] = uint32_t(first0 ? c1 : c0);
                ++sp;
                node = first0 ? c0 : c1;
            } else if (n0) node = c0;
            else if (n1) node = c1;
            else {
                if (sp == 0) break;
                --sp;
                node = int(words[sp * kScanBlock]);
            }
        }
        if (h.dD >= 0.0f) { resolve_exact(a, t_min, t_max, h, false, 0.0f, 0.0f, -1); }
        return;
    }
#endif
    const FilterRay fr = filter_ray(sg, o, d, a, ia, t_min);
    const int n_fwords = sg.n_fpairs >> 5;
#ifdef RT_SPHERES_WAVE_UNIFORM_FILTER
    for (int fw0 = 0; fw0 < n_fwords; fw0 += kMaskWords) {
        const int nw = n_fwords - fw0 < kMaskWords ? n_fwords - fw0 : kMaskWords;
        const uint32_t summary = filter_words(sg, fr, h.bhi, fw0, nw, words RT_STAMP_ARG);
        RT_STAMP(2);
        walk_flagged(summary, sg.n_ewords + fw0, words, visit RT_STAMP_ARG);
        RT_STAMP(3);
    }
#elif defined(RT_SPHERES_SHARED_WALK)
    // Levels 1 AND 2 shared out among the lanes. Level 1 as below, except that the lane that filtered a batch for
    // its owner pushes the flagged PAIRS to a second queue under the owner's name instead of writing flag words; when that queue
    // is worked off every lane takes one entry, whoever owns it: the owner's ray through ds_bpermute (o, d, a: 7 dwords), the pair's b, c, D
    // with the reference's operations (shapes.h:83-86), and for each sphere with D >= 0 the root the reference would accept against
    // t_min alone (near root first, shapes.h:90-97), in FP64 -- every candidate, not only the surviving one as the per-lane walk
    // does. The smallest root of an owner's candidates is kept in the owner's cell A (ds_min_u64 on the double's bits, mapped to an
    // ordered integer), its primitive in cell I, written by the one lane whose root IS the cell's value after all of the round's
    // minima are in (LDS operations of a wave complete in order). min is order-independent, and the reference's sequential rule --
    // a sphere replaces the best so far when its accepted root is smaller, or equal with a larger index -- picks exactly the
    // smallest accepted root (a near root beyond the best so far fails, and then so does its far root). Two candidates of one ray
    // with the SAME root bits (duplicate spheres) cannot be ordered by cell A: the second ds_min returns its own value, and that
    // ray takes the list-order scan (`redo`), as does a root of +-0. The owner accepts the cell against its exact-group hit.
    constexpr int kChunk = 12, kPairRows = 11, kPairCap = kPairRows * 64, kIdxRow = 11, kCellRow = 12, kQueueRow = 14, kQueueCap = 128;
    const int lane = int(threadIdx.x & 63);
    uint32_t* wq = words - lane;                               // the wave's 64 columns: row k at wq + k * kScanBlock
    unsigned long long* const cells = reinterpret_cast<unsigned long long*>(wq + kCellRow * kScanBlock);
    auto cell_of = [&](int l) { return cells + (l >> 5) * (kScanBlock / 2) + (l & 31); };   // rows 12, 13: 32 cells each
    constexpr unsigned long long kNoHit = ~0ull;
    const unsigned long long here = __ballot(true);            // (the loops below are wave-uniform: the set does not change)
    const int rank_here = __popcll(here & ((1ull << lane) - 1ull)), n_here = __popcll(here);
    int q_len = 0, bq_len = 0;                                 // wave-uniform
    auto walk_queue = [&]() {
        if (q_len == 0) return;
        *cell_of(lane) = kNoHit;
        asm volatile("" ::: "memory");                         // (LDS operations of one wave complete in order: only the compiler must keep it)
        for (int base = 0; base < q_len; base += n_here) {
            RT_COUNT(0);
            const int e = base + rank_here;
            const bool work = e < q_len;
            const uint32_t ent = work ? wq[(e >> 6) * kScanBlock + (e & 63)] : uint32_t(lane);
            const int owner = int(ent & 63u), pair = int(ent >> 6);
            const f2 oO01 = {__shfl(o.x, owner), __shfl(o.y, owner)}, oO2A = {__shfl(o.z, owner), __shfl(a, owner)};
            const f2 oD01 = {__shfl(d.x, owner), __shfl(d.y, owner)}, oD2 = {__shfl(d.z, owner), 0.0f};
            f2 b = {0.0f, 0.0f}, D = {-1.0f, -1.0f};
            if (work) {
                RT_COUNT_LANES(1);
                const float4 A = sg.fq[2 * pair], B = sg.fq[2 * pair + 1];
                const f2 cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, r2 = {B.z, B.w};
                const f2 x = pk_sub_c<0>(oO01, cx), y = pk_sub_c<1>(oO01, cy), z = pk_sub_c<0>(oO2A, cz);
                b = pk_mul_c<0>(oD01, x) + pk_mul_c<1>(oD01, y) + pk_mul_c<0>(oD2, z);   // shapes.h:83
                const f2 c = x * x + y * y + z * z - r2;                                // shapes.h:84
                D = b * b - pk_mul_c<1>(oO2A, c);                                       // shapes.h:86
            }
            uint32_t pend = (D.x >= 0.0f ? 1u : 0u) | (D.y >= 0.0f ? 2u : 0u);          // mostly 0 or one bit
            const Recip64 ra = recip64(double(oO2A.y));
            while (__ballot(pend != 0u) != 0ull) {                                      // one candidate per lane and trip
                unsigned long long key = kNoHit;
                int idx = -1;
                if (pend != 0u) {
                    RT_COUNT_LANES(2);
                    const bool second = (pend & 1u) == 0u;
                    pend &= pend - 1u;
                    const float bb = second ? b.y : b.x, DD = second ? D.y : D.x;
                    idx = sph_prim[2 * pair + (second ? 1 : 0)];
                    const double sqrtd = sqrt64(double(DD)), nb = double(-bb);
                    double root = div64(nb - sqrtd, ra);
                    if (root < t_min) root = div64(nb + sqrtd, ra);
                    if (root >= t_min) {
                        const unsigned long long u = (unsigned long long)__double_as_longlong(root);
                        key = root == 0.0 ? 0ull : ((u >> 63) ? ~u : (u | 0x8000000000000000ull));   // ordered like the doubles; +-0 cannot be told apart
                        if (atomicMin(cell_of(owner), key) == key) atomicMin(cell_of(owner), 0ull);   // two candidates with the same root bits
                    }
                }
                asm volatile("" ::: "memory");
                if (key != kNoHit && *cell_of(owner) == key) wq[kIdxRow * kScanBlock + owner] = uint32_t(idx);
                asm volatile("" ::: "memory");
            }
        }
        const unsigned long long k = *cell_of(lane);
        if (k == 0ull) redo = true;
        else if (k != kNoHit) {
            const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
            const double t = __longlong_as_double((long long)u);
            const int idx = int(wq[kIdxRow * kScanBlock + lane]);
            if (h.dD >= 0.0f) resolve_exact(a, t_min, t_max, h, false, 0.0f, 0.0f, -1);
            if (!reject_t(t, t_min, h.bt, idx, h.idx)) {
                h.bt = t; h.idx = idx;
                h.blo = __double2float_rd(t); h.bhi = __double2float_ru(t);
            }
        }
        q_len = 0;
    };
    auto filter_batches = [&](int fw0) {
        for (int base = 0; base < bq_len; base += n_here) {
            const int e = base + rank_here;
            const bool work = e < bq_len;
            const uint32_t ent = work ? wq[(kQueueRow + (e >> 6)) * kScanBlock + (e & 63)] : uint32_t(lane);
            const int owner = int(ent & 63u), lb = int(ent >> 6);          // lb: batch of the chunk, 4 per word
            const f2 D01 = {__shfl(fr.D01.x, owner), __shfl(fr.D01.y, owner)}, D2O = {__shfl(fr.D2O.x, owner), __shfl(fr.D2O.y, owner)};
            const f2 NAM = {__shfl(fr.NAM.x, owner), __shfl(fr.NAM.y, owner)}, W01 = {__shfl(fr.W01.x, owner), __shfl(fr.W01.y, owner)};
            const f2 W2 = {__shfl(fr.W2.x, owner), 0.0f};
            uint32_t m = 0u;                                   // bit set: pair jj (bit 7 - jj) NOT flagged
            if (work) {
                const float4* __restrict__ bd = reinterpret_cast<const float4*>(sg.fi) + size_t(fw0 * 4 + lb) * 16;
#pragma unroll RT_SHARE_UNROLL
                for (int jj = 0; jj < 8; ++jj) {
                    const float4 A = bd[2 * jj], B = bd[2 * jj + 1];
                    const f2 X = {A.x, A.y}, Y = {A.z, A.w}, Z = {B.x, B.y}, K = {B.z, B.w};
                    f2 bq = pk_fma_cc_v<0, 1>(D01, X, D2O);
                    bq = pk_fma_cv_v<1>(D01, Y, bq);
                    bq = pk_fma_cv_v<0>(D2O, Z, bq);
                    f2 Q = pk_fma_cc_v<0, 1>(NAM, K, NAM);
                    Q = pk_fma_cv_v<0>(W01, X, Q);
                    Q = pk_fma_cv_v<1>(W01, Y, Q);
                    Q = pk_fma_cv_v<0>(W2, Z, Q);
                    const f2 T = __builtin_elementwise_fma(bq, bq, Q);   // >= 0 (sign bit clear): flagged
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x) & __float_as_uint(T.y), 31);
                }
            }
            uint32_t cur = work ? (~m & 0xffu) : 0u;           // bit 7 - jj: pair jj of the batch
            const int pbase = (sg.n_ewords + fw0) * 32 + lb * 8;
            for (;;) {
                const unsigned long long pushing = __ballot(cur != 0u);
                if (pushing == 0ull) break;
                const int cnt = __popcll(pushing);
                if (q_len + cnt > kPairCap) walk_queue();
                if (cur != 0u) {
                    const int pp = __builtin_clz(cur) - 24;
                    cur &= ~(0x80u >> pp);
                    const int slot = q_len + __popcll(pushing & ((1ull << lane) - 1ull));
                    wq[(slot >> 6) * kScanBlock + (slot & 63)] = (uint32_t(pbase + pp) << 6) | uint32_t(owner);
                }
                q_len += cnt;
            }
        }
        bq_len = 0;
    };
    for (int fw0 = 0; fw0 < n_fwords; fw0 += kChunk) {
        const int nw = n_fwords - fw0 < kChunk ? n_fwords - fw0 : kChunk;
        filter_words_level0(sg, fr, h.bhi, fw0, nw, [&](int fw, const bool (&nd)[4], int real) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const bool need = nd[bb] && bb < real;
                const unsigned long long pushing = __ballot(need);
                if (pushing == 0ull) continue;
                const int cnt = __popcll(pushing);
                if (bq_len + cnt > kQueueCap) filter_batches(fw0);
                if (need) {
                    const int slot = bq_len + __popcll(pushing & ((1ull << lane) - 1ull));
                    wq[(kQueueRow + (slot >> 6)) * kScanBlock + (slot & 63)] = (uint32_t((fw - fw0) * 4 + bb) << 6) | uint32_t(lane);
                }
                bq_len += cnt;
            }
        });
        filter_batches(fw0);
        RT_STAMP(2);
        walk_queue();                                          // (a hit found now prunes the boxes of the chunks after it)
        RT_STAMP(3);
    }
#else
    // Level 1 shared out among the lanes (as scan_triangles does): the wave-uniform filter runs every batch that
    // SOME lane needs for all 64 lanes (6.9 batches per scan on the 485-sphere frame where a lane needs 2). Level 0 says per lane
    // which batches it needs; (batch, owner) entries go to a queue in rows 14, 15 of the wave's part of the flag-word array, and
    // every lane filters ONE entry's 8 pairs for its owner (the batch's 256 bytes through vector loads, the owner's nine filter
    // constants through ds_bpermute, the same packed FMAs) and writes the 8 flags as one BYTE of the owner's flag word (a batch
    // is a byte of its word) plus the word's bit of the owner's summary (row 12). Chunks of 12 words; the owner's walk is as before.
    constexpr int kChunk = 12, kSumRow = 12, kQueueRow = 14, kQueueCap = 128;
    const int lane = int(threadIdx.x & 63);
    uint32_t* wq = words - lane;                               // the wave's 64 columns: row k at wq + k * kScanBlock
    const unsigned long long here = __ballot(true);            // (the loops below are wave-uniform: the set does not change)
    const int rank_here = __popcll(here & ((1ull << lane) - 1ull)), n_here = __popcll(here);
    for (int fw0 = 0; fw0 < n_fwords; fw0 += kChunk) {
        const int nw = n_fwords - fw0 < kChunk ? n_fwords - fw0 : kChunk;
        for (int wi = 0; wi < nw; ++wi) words[wi * kScanBlock] = 0u;
        words[kSumRow * kScanBlock] = 0u;
        int bq_len = 0;                                        // wave-uniform
        auto filter_batches = [&]() {
            asm volatile("" ::: "memory");                     // (LDS operations of one wave complete in order: only the compiler must keep it)
            for (int base = 0; base < bq_len; base += n_here) {
                const int e = base + rank_here;
                const bool work = e < bq_len;
                const uint32_t ent = work ? wq[(kQueueRow + (e >> 6)) * kScanBlock + (e & 63)] : uint32_t(lane);
                const int owner = int(ent & 63u), lb = int(ent >> 6);          // lb: batch of the chunk, 4 per word
                const f2 D01 = {__shfl(fr.D01.x, owner), __shfl(fr.D01.y, owner)}, D2O = {__shfl(fr.D2O.x, owner), __shfl(fr.D2O.y, owner)};
                const f2 NAM = {__shfl(fr.NAM.x, owner), __shfl(fr.NAM.y, owner)}, W01 = {__shfl(fr.W01.x, owner), __shfl(fr.W01.y, owner)};
                const f2 W2 = {__shfl(fr.W2.x, owner), 0.0f};
                if (work) {
                    const float4* __restrict__ bd = reinterpret_cast<const float4*>(sg.fi) + size_t(fw0 * 4 + lb) * 16;
                    uint32_t m = 0u;                           // bit set: pair jj (bit 7 - jj) NOT flagged
#pragma unroll RT_SHARE_UNROLL
                    for (int jj = 0; jj < 8; ++jj) {
                        const float4 A = bd[2 * jj], B = bd[2 * jj + 1];
                        const f2 X = {A.x, A.y}, Y = {A.z, A.w}, Z = {B.x, B.y}, K = {B.z, B.w};
                        f2 bq = pk_fma_cc_v<0, 1>(D01, X, D2O);
                        bq = pk_fma_cv_v<1>(D01, Y, bq);
                        bq = pk_fma_cv_v<0>(D2O, Z, bq);
                        f2 Q = pk_fma_cc_v<0, 1>(NAM, K, NAM);
                        Q = pk_fma_cv_v<0>(W01, X, Q);
                        Q = pk_fma_cv_v<1>(W01, Y, Q);
                        Q = pk_fma_cv_v<0>(W2, Z, Q);
                        const f2 T = __builtin_elementwise_fma(bq, bq, Q);   // >= 0 (sign bit clear): flagged
                        m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x) & __float_as_uint(T.y), 31);
                    }
                    const uint32_t fb = ~m & 0xffu;
                    if (fb != 0u) {
                        const int wi = lb >> 2, bb = lb & 3;
                        reinterpret_cast<uint8_t*>(wq + wi * kScanBlock + owner)[3 - bb] = uint8_t(fb);   // pairs 8 bb .. 8 bb + 7 <-> bits 31 - 8 bb ..
                        atomicOr(wq + kSumRow * kScanBlock + owner, 1u << wi);
                    }
                }
            }
            bq_len = 0;
        };
        filter_words_level0(sg, fr, h.bhi, fw0, nw, [&](int fw, const bool (&nd)[4], int real) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const bool need = nd[bb] && bb < real;
                const unsigned long long pushing = __ballot(need);
                if (pushing == 0ull) continue;
                const int cnt = __popcll(pushing);
                if (bq_len + cnt > kQueueCap) filter_batches();
                if (need) {
                    const int slot = bq_len + __popcll(pushing & ((1ull << lane) - 1ull));
                    wq[(kQueueRow + (slot >> 6)) * kScanBlock + (slot & 63)] = (uint32_t((fw - fw0) * 4 + bb) << 6) | uint32_t(lane);
                }
                bq_len += cnt;
            }
        });
        filter_batches();
        asm volatile("" ::: "memory");
        const uint32_t summary = words[kSumRow * kScanBlock];
        RT_STAMP(2);
        walk_flagged(summary, sg.n_ewords + fw0, words, visit RT_STAMP_ARG);
        RT_STAMP(3);
    }
#endif
    if (h.dD >= 0.0f) { resolve_exact(a, t_min, t_max, h, false, 0.0f, 0.0f, -1); }
    RT_STAMP(3);
}

}  // namespace rt
