// libray_amd.so -- MI355X-native implementation of the joblobob/ray per-pixel path behind the C ABI
// of include/ray_amd.h: camera::get_ray (camera.h:21-30) -> hitFromList (hitPosition.cpp:6-22) over
// structure-of-arrays primitives -> calc::ray_color (calc.cpp:90-131) as an iterative bounce loop.
// gfx950 only; compiled with -ffp-contract=off (see rt_device.hpp for the arithmetic contract).
//
// No CPU fallback: every compute entry point needs rt_init() to have bound a HIP device.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "../../include/ray_amd.h"
#include "../../include/ray_amd_tri.h"
#include "rt_device.hpp"
#include "rt_scan.hpp"

using namespace rt;

// ================================================================================================
// Device-side scene (HBM layout, DESIGN.md "data layout")
// ================================================================================================
struct DevScene {
    int n_sph, n_rect, n_tri, n_prims;
    int n_rect_xy, n_rect_xz;              // the rect arrays are sorted by axis: XY first, then XZ, then YZ
    const float4* __restrict__ sph;        // [n_sph]  (cx, cy, cz, r)
    const int* __restrict__ sph_prim;      // [n_sph]  flattened primitive index, ascending
    const double* __restrict__ rect;       // [n_rect][5] (a0, a1, b0, b1, k)
    const float* __restrict__ rect_f;      // [n_rect][5] (k nearest, a0 down, a1 up, b0 down, b1 up): FP32 enclosures
    const int* __restrict__ rect_axis;     // [n_rect] 0 = XY (z = k), 1 = XZ (y = k), 2 = YZ (x = k)
    const int* __restrict__ rect_prim;     // [n_rect]
    const float* __restrict__ tri;         // [n_tri][16] v0, e1, e2, unit normal, bounding sphere centre, radius^2 (ray_amd_tri.h)
    const int* __restrict__ tri_prim;      // [n_tri]
    const int* __restrict__ prim_kind;     // [n_prims] 0 sphere, 1 rect, 2 triangle (device kinds)
    const int* __restrict__ prim_local;    // [n_prims] index inside its kind's arrays
    const int* __restrict__ prim_mat;      // [n_prims]
    const int* __restrict__ mat_kind;      // [n_mats]
    const float* __restrict__ mat_rgb;     // [n_mats][3]
    const float* __restrict__ mat_fuzz;    // [n_mats]
    const double* __restrict__ mat_ir;     // [n_mats]
    const double* __restrict__ mat_inv_ir; // [n_mats] 1.0 / ir (host IEEE division, hitPosition.h:256)
    int light_kind;                        // RT_LIGHT_BASE, RT_PRIM_SPHERE, RT_PRIM_XY_RECT, RT_PRIM_XZ_RECT
    float light_c[3], light_r;             // sphere light
    double light_rect[5];                  // rect light
    float bg[3];
    FilterGroup sg, tg;                    // exact / filter partition of the sphere and triangle slots (rt_scan.hpp)
    int fast_ok;                           // 0: a primitive is non-finite / out of range -> list-order scan only
    const float* __restrict__ rect_box;    // [12 per pair of rects] centre x (rect 0, rect 1), y, z, half extent x, y, z (box_pair's form), rounded outwards
    float rect_extent;                     // max |coordinate| of the rects
    const float* __restrict__ rect_bbox;   // [12 per pair of batches] the same around every batch of 16 rects (scenes with >= kRectFlagMin rects:
                                           // the rects are then in spatially compact batches, not sorted by axis)
    const float4* __restrict__ tbvh;       // triangle BVH (null: none), 4 float4 per node: the two children's boxes relative to tg's g as
                                           // (C.x0, C.x1, C.y0, C.y1), (C.z0, C.z1, E.x0, E.x1), (E.y0, E.y1, E.z0, E.z1) (centre, half extent), then
                                           // (child 0, child 1, -, -) as int bits: >= 0 inner node, < 0 leaf ~item (an item: tri slots 2i, 2i + 1)
    float tri_extent;                      // max |box coordinate| of the BVH (relative to g)
    int tbvh_leaf_items;                   // a leaf ~s is the run of item slots [s, s + tbvh_leaf_items) (1 up to 32k items; consecutive
                                           // slots are spatially compact, see build_group)
    int n_med;                             // constant_medium primitives (shapes.h:374-455); any of them: list-order scan only
    const float* __restrict__ med_bound;   // [n_med][6] sphere boundary: cx, cy, cz, r, -, - ; box boundary: p0 xyz, p1 xyz
    const double* __restrict__ med_niv;    // [n_med] neg_inv_density
    const int* __restrict__ med_is_box;    // [n_med]
};

struct DevCamera {
    V3 origin, llc, horizontal, vertical, u, v, w;
    double lens_radius;
};

struct Ray {
    V3 o, d;
    float dl2;   // Ray.h:8 cached direction.lengthSquared()
};
RT_DEV Ray make_ray(V3 o, V3 d) { Ray r; r.o = o; r.d = d; r.dl2 = length_squared(d); return r; }
RT_DEV V3 ray_at(const Ray& r, double t) { return r.o + float(t) * r.d; }   // Ray.h:11

// camera::get_ray, camera.h:21-30
__device__ __forceinline__ Ray get_ray(const DevCamera& c, Rng& g, float s, float t) {
    V3 rd = float(c.lens_radius) * random_in_unit_disk(g);
    V3 offset = c.u * rd.x + c.v * rd.y;
    return make_ray(c.origin + offset, c.llc + s * c.horizontal + t * c.vertical - c.origin - offset);
}

// ------------------------------------------------------------------------------------------------
// Per-primitive hit tests. Each returns true and the accepted t when the reference's shape::hit
// would return a record for the interval [t_min, bt], where (bt, bi) is the best hit so far and the
// tie rule "equal t: the later list entry wins" of the sequential scan (hitPosition.cpp:12-18 with
// the non-strict rejects of shapes.h:93,175) is made explicit through the flattened index `idx`:
// reject(t) = t < t_min || bt < t || (t == bt && idx < bi). NaN compares false everywhere, exactly
// as in the reference, so a NaN t is "accepted"; the caller handles that case (see closest_hit).
// ------------------------------------------------------------------------------------------------
// reject_t: rt_scan.hpp

// sphere::hit, shapes.h:77-97
RT_DEV bool sphere_t(const Ray& r, float4 s, double t_min, double bt, int idx, int bi, double* t_out) {
    V3 oc = r.o - v3(s.x, s.y, s.z);
    float a = r.dl2;
    float b = dot(oc, r.d);
    float c = length_squared(oc) - s.w * s.w;
    float d = b * b - a * c;
    if (d < 0.0f) return false;
    double sqrtd = __builtin_sqrt(double(d));
    double root = (double(-b) - sqrtd) / double(a);
    if (reject_t(root, t_min, bt, idx, bi)) {
        root = (double(-b) + sqrtd) / double(a);
        if (reject_t(root, t_min, bt, idx, bi)) return false;
    }
    *t_out = root;
    return true;
}

// rectangle / xz_rect / yz_rect ::hit, shapes.h:172-180, 238-247, 304-312 (all-double arithmetic:
// k and the bounds are doubles in the reference)
// (ok, dk): origin / direction component along the rect's normal; (oa, da), (ob, db): in-plane components
RT_DEV bool rect_core(float ok, float dk, float oa, float da, float ob, float db, double q0, double q1, double q2, double q3, double k,
                      double t_min, double bt, int idx, int bi, double* t_out) {
    double t = (k - double(ok)) / double(dk);
    if (reject_t(t, t_min, bt, idx, bi)) return false;
    double a = double(oa) + t * double(da);
    double b = double(ob) + t * double(db);
    if (a < q0 || a > q1 || b < q2 || b > q3) return false;
    *t_out = t;
    return true;
}
template <int AXIS>
RT_DEV bool rect_axis_t(const Ray& r, const double* __restrict__ q, double t_min, double bt, int idx, int bi, double* t_out) {
    const float ok = AXIS == 0 ? r.o.z : (AXIS == 1 ? r.o.y : r.o.x), dk = AXIS == 0 ? r.d.z : (AXIS == 1 ? r.d.y : r.d.x);
    const float oa = AXIS == 2 ? r.o.y : r.o.x, da = AXIS == 2 ? r.d.y : r.d.x;
    const float ob = AXIS == 0 ? r.o.y : r.o.z, db = AXIS == 0 ? r.d.y : r.d.z;
    return rect_core(ok, dk, oa, da, ob, db, q[0], q[1], q[2], q[3], q[4], t_min, bt, idx, bi, t_out);
}
// branches, not selects: a select of ray components by a run-time axis becomes an indexed load from a
// copy of the ray in scratch memory
RT_DEV bool rect_t(const Ray& r, int axis, const double* __restrict__ q, double t_min, double bt, int idx, int bi,
                   double* t_out) {
    if (axis == 0) return rect_axis_t<0>(r, q, t_min, bt, idx, bi, t_out);
    if (axis == 1) return rect_axis_t<1>(r, q, t_min, bt, idx, bi, t_out);
    return rect_axis_t<2>(r, q, t_min, bt, idx, bi, t_out);
}
// the rects of one axis, [j0, j1) of the axis-sorted rect arrays; wave-uniform data through scalar loads
typedef const double __attribute__((address_space(4)))* cdp;
typedef const int __attribute__((address_space(4)))* cip;
typedef const float __attribute__((address_space(4)))* cfp;
// FP32 enclosures first (rect_f: k rounded to nearest, the bounds rounded outwards): a rect whose plane
// distance or in-plane point is certainly out of range is skipped; only the others pay the reference's
// FP64 division. Enclosure: tf = (kf - o) * rcp(d) is within et of the FP64 t (k's rounding u|k|/|d|, the
// subtraction, rcp's 1 ulp and the product: < 5 ulp of |t| + u|k|/|d|; et takes 16 ulp of both), the
// in-plane coordinate within ea = et*|da| + 16 ulp (|oa| + |tf*da|). inf / NaN never reject.
template <int AXIS>
__device__ __forceinline__ void scan_rects(const double* __restrict__ rect, const float* __restrict__ rect_f, const int* __restrict__ rect_prim,
                                           int j0, int j1, const Ray& r, double t_min, float tminf, double& bt, int& bi, float& btf,
                                           bool& saw_nan) {
    const float ok = AXIS == 0 ? r.o.z : (AXIS == 1 ? r.o.y : r.o.x), dk = AXIS == 0 ? r.d.z : (AXIS == 1 ? r.d.y : r.d.x);
    const float oa = AXIS == 2 ? r.o.y : r.o.x, da = AXIS == 2 ? r.d.y : r.d.x;
    const float ob = AXIS == 0 ? r.o.y : r.o.z, db = AXIS == 0 ? r.d.y : r.d.z;
    const float idk = __builtin_amdgcn_rcpf(dk);
    cdp q = (cdp)(uintptr_t)rect;
    cfp f = (cfp)(uintptr_t)rect_f;
    cip pr = (cip)(uintptr_t)rect_prim;
    for (int j = j0; j < j1; ++j) {
        const float kf = f[5 * j], a0 = f[5 * j + 1], a1 = f[5 * j + 2], b0 = f[5 * j + 3], b1 = f[5 * j + 4];
        const float tf = (kf - ok) * idk;
        const float et = ((__builtin_fabsf(kf) + __builtin_fabsf(ok)) * __builtin_fabsf(idk) + __builtin_fabsf(tf)) * 1.0e-6f;
        const float pa = tf * da, pb = tf * db;
        const float af = oa + pa, bf = ob + pb;
        const float ea = et * __builtin_fabsf(da) + (__builtin_fabsf(oa) + __builtin_fabsf(pa)) * 1.0e-6f;
        const float eb = et * __builtin_fabsf(db) + (__builtin_fabsf(ob) + __builtin_fabsf(pb)) * 1.0e-6f;
        const bool out = tf - et > btf || tf + et < tminf || af + ea < a0 || af - ea > a1 || bf + eb < b0 || bf - eb > b1;
        if (!out) {
            double t;
            const int idx = pr[j];
            if (rect_core(ok, dk, oa, da, ob, db, q[5 * j], q[5 * j + 1], q[5 * j + 2], q[5 * j + 3], q[5 * j + 4], t_min, bt, idx, bi, &t)) {
                if (t != t) saw_nan = true; else { bt = t; bi = idx; btf = __double2float_ru(t); }
            }
        }
    }
}

// Scenes with many rects (box_scene: 156): a flag pass, one bit per rect and lane (rect j <-> bit 31 - j % 32 of
// word j / 32 of the chunk), then a per-lane walk that runs the reference's FP64 test on the lane's own flagged
// rects only. The flag is the slab test of rt_scan.hpp on the rect as a box of zero thickness, two rects per
// packed operation, widened by m = 2e-5 (|o| + extent of the rects): the reference's hit point lies on the rect
// up to FP64 rounding, the FP32 slab arithmetic moves a plane by < 3e-7 (|o| + extent) and the clamp of a tiny
// direction component moves the ray by less than that inside the scene. Axis-parallel rays (a direction
// component exactly 0, where the reference's plane distance is inf or NaN and list order decides) do not come here.
__device__ __forceinline__ uint32_t flag_rects(const DevScene& sc, const FilterRay& br, float t_hi, int c0, int c1, uint32_t* words) {
    cf2p rb = RT_CONST_F2(sc.rect_box), bb = RT_CONST_F2(sc.rect_bbox);
    uint32_t summary = 0u;
    for (int w = c0 >> 5; 32 * w < c1; ++w) {              // one word = 32 rects = two batches of 16
        bool b0, b1;
        box_pair(br, t_hi, bb[6 * w], bb[6 * w + 1], bb[6 * w + 2], bb[6 * w + 3], bb[6 * w + 4], bb[6 * w + 5], b0, b1);
        const bool any0 = __ballot(b0) != 0ull, any1 = __ballot(b1) != 0ull && 32 * w + 16 < c1;
        if (!any0 && !any1) continue;                      // no lane needs anything of this word: its summary bit stays 0
        uint32_t cur = 0u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!(h ? any1 : any0)) { cur <<= 16; continue; }
            uint32_t half = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int p = 16 * w + 8 * h + k;          // pair of rects 2p, 2p + 1 (pads beyond the last pair repeat zeros: box at the origin)
                bool n0, n1;
                box_pair(br, t_hi, rb[6 * p], rb[6 * p + 1], rb[6 * p + 2], rb[6 * p + 3], rb[6 * p + 4], rb[6 * p + 5], n0, n1);
                half = (half << 2) | (n0 && 2 * p < c1 ? 2u : 0u) | (n1 && 2 * p + 1 < c1 ? 1u : 0u);
            }
            cur = (cur << 16) | ((h ? b1 : b0) ? half : 0u);
        }
        const int wi = w - (c0 >> 5);
        words[wi * kScanBlock] = cur;
        summary |= (cur != 0u ? 1u : 0u) << wi;
    }
    return summary;
}
constexpr int kRectFlagMin = 16;   // fewer rects: the direct loop of scan_rects

#define RT_WORDS_ALIGN __align__(16)   // scan_triangles keeps 8-byte cells in it (the staged experiment reads rows as float4)
// Triangle (not in the reference; DESIGN.md "triangles"): Moeller-Trumbore, FP32, NaN = miss.
RT_DEV bool tri_t(const Ray& r, const float* __restrict__ q, double t_min, double bt, int idx, int bi, double* t_out) {
    V3 v0 = v3(q[0], q[1], q[2]), e1 = v3(q[3], q[4], q[5]), e2 = v3(q[6], q[7], q[8]);
    V3 pvec = cross(r.d, e2);
    float det = dot(e1, pvec);
    if (!(det != 0.0f)) return false;
    float inv = 1.0f / det;
    V3 tvec = r.o - v0;
    float u = dot(tvec, pvec) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(r.d, qvec) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    double t = double(dot(e2, qvec) * inv);
    V3 dp = ray_at(r, t) - v3(q[12], q[13], q[14]);   // hit point inside the bounding sphere (ray_amd_tri.h)
    if (!(length_squared(dp) <= q[15])) return false;
    if (!(t >= t_min && t <= bt) || (t == bt && idx < bi)) return false;
    *t_out = t;
    return true;
}

struct Hit {
    double t;
    int prim;   // flattened index, -1 = miss
};

// box::hit of a constant_medium's boundary (shapes.h:357-360): hitFromList over the six sides in constructor order
// (shapes.h:347-354) with the reference's own comparisons; only t is needed. idx = 1 > bi = 0: equal t replaces.
RT_DEV bool box_bound_t(const Ray& r, const float* __restrict__ b, double t_min, double t_max, double* t_out) {
    const double x0 = double(b[0]), y0 = double(b[1]), z0 = double(b[2]), x1 = double(b[3]), y1 = double(b[4]), z1 = double(b[5]);
    double closest = t_max, t;
    bool any = false;
    if (rect_core(r.o.z, r.d.z, r.o.x, r.d.x, r.o.y, r.d.y, x0, x1, y0, y1, z1, t_min, closest, 1, 0, &t)) { closest = t; any = true; }
    if (rect_core(r.o.z, r.d.z, r.o.x, r.d.x, r.o.y, r.d.y, x0, x1, y0, y1, z0, t_min, closest, 1, 0, &t)) { closest = t; any = true; }
    if (rect_core(r.o.y, r.d.y, r.o.x, r.d.x, r.o.z, r.d.z, x0, x1, z0, z1, y1, t_min, closest, 1, 0, &t)) { closest = t; any = true; }
    if (rect_core(r.o.y, r.d.y, r.o.x, r.d.x, r.o.z, r.d.z, x0, x1, z0, z1, y0, t_min, closest, 1, 0, &t)) { closest = t; any = true; }
    if (rect_core(r.o.x, r.d.x, r.o.y, r.d.y, r.o.z, r.d.z, y0, y1, z0, z1, x1, t_min, closest, 1, 0, &t)) { closest = t; any = true; }
    if (rect_core(r.o.x, r.d.x, r.o.y, r.d.y, r.o.z, r.d.z, y0, y1, z0, z1, x0, t_min, closest, 1, 0, &t)) { closest = t; any = true; }
    *t_out = closest;
    return any;
}
// constant_medium::hit, shapes.h:390-444: entry and exit of the boundary over the whole line, clamped to
// [t_min, t_max]; ONE draw from the ray's stream (gen01 in the reference) whenever that interval is non-empty.
__device__ __forceinline__ bool medium_t(const float* __restrict__ b, bool is_box, double neg_inv_density, const Ray& r, Rng& g, double t_min,
                                         double t_max, double* t_out) {
    double t1, t2;
    const float4 s = make_float4(b[0], b[1], b[2], b[3]);
    if (is_box) {
        if (!box_bound_t(r, b, -kInf, kInf, &t1)) return false;
        if (!box_bound_t(r, b, t1 + 0.0001, kInf, &t2)) return false;
    } else {
        if (!sphere_t(r, s, -kInf, kInf, 0, -1, &t1)) return false;
        if (!sphere_t(r, s, t1 + 0.0001, kInf, 0, -1, &t2)) return false;
    }
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return false;
    if (t1 < 0) t1 = 0;
    const float ray_length = length(r.d);
    const double distance_inside_boundary = (t2 - t1) * double(ray_length);
    const double hit_distance = neg_inv_density * log_portable(double(g.u01()));
    if (hit_distance > distance_inside_boundary) return false;
    *t_out = t1 + hit_distance / double(ray_length);
    return true;
}

// hitFromList (hitPosition.cpp:6-22) over the per-kind arrays. For totally ordered t the sequential
// scan equals "smallest t, ties to the later entry", which is what the grouped scan computes. If any
// accepted t is NaN the list order matters (a NaN closest_so_far accepts everything after it), so
// that lane redoes the scan in list order -- rare (a ray component exactly 0 on a plane it lies in).
// MEDIA: false = the scene has no constant_medium (the launcher knows), its code is compiled out of this instance
template <bool MEDIA = true>
__device__ __forceinline__ Hit closest_hit_listorder(const DevScene& sc, const Ray& r, double t_min, double t_max, Rng& g) {
    Hit h; h.t = t_max; h.prim = -1;
    for (int i = 0; i < sc.n_prims; ++i) {
        int k = sc.prim_kind[i], j = sc.prim_local[i];
        double t; bool ok;
        if (k == 0) ok = sphere_t(r, sc.sph[j], t_min, h.t, i, h.prim, &t);
        else if (k == 1) ok = rect_t(r, sc.rect_axis[j], sc.rect + 5 * j, t_min, h.t, i, h.prim, &t);
        else if (MEDIA && k == 3) ok = medium_t(sc.med_bound + 6 * j, sc.med_is_box[j] != 0, sc.med_niv[j], r, g, t_min, h.t, &t);
        else ok = tri_t(r, sc.tri + 16 * j, t_min, h.t, i, h.prim, &t);
        if (ok) { h.t = t; h.prim = i; }
    }
    return h;
}

// Triangles of one ray: the sphere filter of rt_scan.hpp on the bounding spheres of items of one or two
// triangles (every hit
// point lies inside its triangle's sphere by definition of the primitive, ray_amd_tri.h), then
// tri_t on the flagged pairs. h is exact on entry and on return.
__device__ __forceinline__ void scan_triangles(const DevScene& sc, const Ray& r, double t_min, Hit& h, uint32_t* words, bool& redo RT_STAMP_PARAM) {
    const FilterGroup& tg = sc.tg;
    if (tg.n_epairs == 0 && tg.n_fpairs == 0) return;
    // The lanes that test an item for its owner (walk_queue) take the owner's range as FLOATS, rounded outwards, and a cell keeps
    // only the smallest key: with a t_min that is no float (rt_hit_rays / shape::hit with any double; the render path's is
    // double(0.001f)), a hit at rd(t_min) <= t < t_min would take the cell and hide the valid hit behind it. Such rays take the
    // list-order scan.
    if (double(float(t_min)) != t_min) { redo = true; return; }
    auto visit = [&](int item) {       // an item: up to two triangles under one bounding sphere
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int slot = 2 * item + hh;
            const int idx = sc.tri_prim[slot];
            double t;
            if (idx >= 0 && tri_t(r, sc.tri + 16 * slot, t_min, h.t, idx, h.prim, &t)) { h.t = t; h.prim = idx; }
        }
    };
    for (int p = 0; p < 2 * tg.n_epairs; ++p) visit(p);
    if (tg.n_fpairs == 0) return;
    const FilterRay fr = filter_ray(tg, r.o, r.d, r.dl2, __builtin_amdgcn_rcpf(r.dl2), t_min);
    const int n_fwords = tg.n_fpairs >> 5;
#ifdef RT_EXPERIMENT_TRI_STAGED
    // the lanes that are here copy the filter data to LDS for each other (the loops below are wave-uniform: the set does not change)
    const unsigned long long here = __ballot(true);
    const int rank = int(__builtin_amdgcn_mbcnt_hi(uint32_t(here >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(here), 0u))), nact = __popcll(here);
#endif
#ifdef RT_EXPERIMENT_TRI_STAGED
    for (int fw0 = 0; fw0 < n_fwords; fw0 += kItemChunkWords) {
        if (!chunk_needed(tg, fr, __double2float_ru(h.t), fw0 / kItemChunkWords)) continue;
        const int nw = n_fwords - fw0 < kItemChunkWords ? n_fwords - fw0 : kItemChunkWords;
        const uint32_t summary = filter_words_staged(tg, fr, __double2float_ru(h.t), fw0, nw, words, rank, nact RT_STAMP_ARG);
        RT_STAMP(2);
        walk_flagged_items(summary, tg.n_ewords + fw0, words, visit RT_STAMP_ARG);
        RT_STAMP(3);
    }
#else
    // The flagged items of the wave's 64 rays go to ONE queue in the wave's part of the flag-word array (entries: item | owner lane)
    // and the lanes share the triangle tests out evenly: a walk in which every lane tests its own items takes as many passes as
    // the lane with most items needs - 19 passes per scan on the 9.8k-triangle frame for 2.4 items per lane, 12 % of the lanes on
    // in 41 % of the frame's time. A worker lane fetches its entry's ray from the owner lane (ds_bpermute), tests the item's one
    // or two triangles against the owner's t range, and folds a hit into the owner's cell with an LDS atomic min on the key
    // (t, ~index): the order of tri_t's tie rule (equal t: the later primitive), so the cell ends up holding what the owner's
    // own walk would have found, whatever the order of the tests. The owner then accepts it against its best hit as tri_t does.
    // The queue is worked off when it is full, after a chunk once it holds a pass's worth (a hit found then prunes the chunks
    // after it), and at the end. A hit at t == 0 (either sign: the key could not tell them apart) sends its ray to the list-order scan.
    const int lane = int(threadIdx.x & 63);
    uint32_t* wq = words - lane;                               // the wave's 64 columns: row k at wq + k * kScanBlock
    constexpr int kBatchRows = 2;                              // rows 0, 1: the queue of (batch, owner) entries of the shared filter
    constexpr int kChunkRows = 2;                              // rows 2, 3: the queue of (chunk, owner) entries of the shared level 0
    constexpr int kQueueRows = kMaskWords - 2, kQueueCap = (kQueueRows - kBatchRows - kChunkRows) * 64;   // the last two rows hold the 64 cells
    uint32_t* const cq = wq + kBatchRows * kScanBlock;         // the chunk queue's first row
    uint32_t* const iq = wq + (kBatchRows + kChunkRows) * kScanBlock;   // the item queue's first row
    unsigned long long* const cells = reinterpret_cast<unsigned long long*>(wq + kQueueRows * kScanBlock);
    auto cell_of = [&](int l) { return cells + (l >> 5) * (kScanBlock / 2) + (l & 31); };
    constexpr unsigned long long kNoHit = ~0ull;
    int q_len = 0;                                             // wave-uniform
    const unsigned long long here = __ballot(true);            // (the loops below are wave-uniform: the set does not change)
    const int rank_here = __popcll(here & ((1ull << lane) - 1ull)), n_here = __popcll(here);
    // (all = false, here and in the two queues before this one: whole rounds only, what is left stays queued for a fuller round later)
    auto keep_rest = [&](uint32_t* q, int n_take, int& len) {
        const int rem = len - n_take;                          // < n_here <= 64 entries: moved to the front (by rank: lanes outside the scan are not here)
        if (rem > 0 && n_take > 0) {
            asm volatile("" ::: "memory");
            uint32_t ent = 0u;
            if (rank_here < rem) ent = q[((n_take + rank_here) >> 6) * kScanBlock + ((n_take + rank_here) & 63)];
            asm volatile("" ::: "memory");
            if (rank_here < rem) q[rank_here] = ent;
            asm volatile("" ::: "memory");
        }
        len = rem;
    };
    auto walk_queue = [&](bool all) {
        const int n_take = all ? q_len : q_len / n_here * n_here;
        if (n_take == 0) return;
        *cell_of(lane) = kNoHit;
        const float bt_f = __double2float_ru(h.t);
        const float tmin_f = __double2float_rd(t_min);
        asm volatile("" ::: "memory");                         // (LDS operations of one wave complete in order: only the compiler must keep it)
        const int lshift = n_take * 2 <= n_here ? 1 : 0;       // few entries: two lanes to an item, a triangle each
        for (int base = 0; base < (n_take << lshift); base += n_here) {   // the lanes that are in this scan share the entries out
            RT_COUNT(0);
            const int u = base + rank_here;
            const int e = u >> lshift;
            const bool work = e < n_take;
            const uint32_t ent = work ? iq[(e >> 6) * kScanBlock + (e & 63)] : uint32_t(lane);
            const int owner = int(ent & 63u), item = int(ent >> 6);
            Ray rr;                                            // the owner's ray and t range (every lane of the scan takes part in the exchange)
            rr.o = v3(__shfl(r.o.x, owner), __shfl(r.o.y, owner), __shfl(r.o.z, owner));
            rr.d = v3(__shfl(r.d.x, owner), __shfl(r.d.y, owner), __shfl(r.d.z, owner));
            rr.dl2 = 0.0f;
            const float bt_o = __shfl(bt_f, owner), tmin_o = __shfl(tmin_f, owner);
            if (work) {
                const int h0 = lshift ? (u & 1) : 0, h1 = lshift ? (u & 1) + 1 : 2;
                for (int hh = h0; hh < h1; ++hh) {
                    const int slot = 2 * item + hh;
                    const int idx = sc.tri_prim[slot];
                    double t;
                    if (idx >= 0 && tri_t(rr, sc.tri + 16 * slot, double(tmin_o), double(bt_o), idx, -1, &t)) {
                        const float tf = float(t);             // exact: t is a widened float
                        uint32_t tb = __float_as_uint(tf);
                        tb = (tb & 0x80000000u) ? ~tb : (tb | 0x80000000u);                    // ordered like the floats
                        unsigned long long key = ((unsigned long long)tb << 32) | (unsigned long long)(0xffffffffu - uint32_t(idx));
                        if (tf == 0.0f) key = 0ull;
                        atomicMin(cell_of(owner), key);
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
        const unsigned long long k = *cell_of(lane);
        if (k == 0ull) redo = true;
        else if (k != kNoHit) {
            uint32_t tb = uint32_t(k >> 32);
            tb = (tb & 0x80000000u) ? (tb & 0x7fffffffu) : ~tb;
            const double t = double(__uint_as_float(tb));
            const int idx = int(0xffffffffu - uint32_t(k));
            if (t >= t_min && t <= h.t && !(t == h.t && idx < h.prim)) { h.t = t; h.prim = idx; }   // tri_t's last test, against the exact range
        }
        keep_rest(iq, n_take, q_len);
    };
    // Level 1 is shared out the same way. The wave-uniform filter runs every batch that SOME lane needs for all 64 lanes (18.9
    // batches per scan on the 9.8k-triangle frame where a lane needs 3). Level 0 says per lane which batches it needs: (batch,
    // owner) entries go to a second queue (one ballot per batch, no per-lane loop), and every lane filters ONE entry's 8 pairs
    // for the entry's owner - the batch's 256 bytes through vector loads instead of SGPR operands, the owner's nine filter
    // constants through ds_bpermute, the same packed FMAs - and pushes the flagged items to the item queue under the owner's name.
    constexpr int kBatchCap = kBatchRows * 64;
    int bq_len = 0;
    auto filter_batches = [&](bool all) {
        const int n_take = all ? bq_len : bq_len / n_here * n_here;
        for (int base = 0; base < n_take; base += n_here) {
            const int e = base + rank_here;
            const bool work = e < n_take;
            const uint32_t ent = work ? wq[(e >> 6) * kScanBlock + (e & 63)] : uint32_t(lane);
            const int owner = int(ent & 63u), batch = int(ent >> 6);
            const f2 D01 = {__shfl(fr.D01.x, owner), __shfl(fr.D01.y, owner)}, D2O = {__shfl(fr.D2O.x, owner), __shfl(fr.D2O.y, owner)};
            const f2 NAM = {__shfl(fr.NAM.x, owner), __shfl(fr.NAM.y, owner)}, W01 = {__shfl(fr.W01.x, owner), __shfl(fr.W01.y, owner)};
            const f2 W2 = {__shfl(fr.W2.x, owner), 0.0f};
            uint32_t m = 0u, m1 = 0u;                           // bit set: item 0 / item 1 of pair jj (bit 7 - jj) NOT flagged
            if (work) {
                const float4* __restrict__ bd = reinterpret_cast<const float4*>(tg.fi) + size_t(batch) * 16;
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
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(T.x), 31);
                    m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(T.y), 31);
                }
            }
            uint32_t cur0 = work ? (~m & 0xffu) : 0u, cur1 = work ? (~m1 & 0xffu) : 0u;   // bit 7 - jj: pair jj of the batch
            const int pbase = tg.n_ewords * 32 + batch * 8;
            for (;;) {
                const unsigned long long push0 = __ballot(cur0 != 0u), push1 = __ballot(cur1 != 0u);
                if ((push0 | push1) == 0ull) break;
                const int cnt0 = __popcll(push0), cnt = cnt0 + __popcll(push1);
                if (q_len + cnt > kQueueCap) walk_queue(false);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (cur0 != 0u) {
                    const int p = __builtin_clz(cur0) - 24;
                    cur0 &= ~(0x80u >> p);
                    const int slot = q_len + __popcll(push0 & below);
                    iq[(slot >> 6) * kScanBlock + (slot & 63)] = (uint32_t(2 * (pbase + p)) << 6) | uint32_t(owner);
                }
                if (cur1 != 0u) {
                    const int p = __builtin_clz(cur1) - 24;
                    cur1 &= ~(0x80u >> p);
                    const int slot = q_len + cnt0 + __popcll(push1 & below);
                    iq[(slot >> 6) * kScanBlock + (slot & 63)] = (uint32_t(2 * (pbase + p) + 1) << 6) | uint32_t(owner);
                }
                q_len += cnt;
            }
        }
        keep_rest(wq, n_take, bq_len);
    };
    // Level 0 is shared out as well (round 3). The wave-uniform box tests run every chunk, word and batch that SOME
    // lane needs for all 64 lanes: on the 9.8k-triangle frame 4.2 of the 10 chunks per scan where a lane needs 1.4, and that was
    // 47 % of the frame. Now only the chunk boxes are tested wave-uniformly (one slab test per pair of chunks); every (chunk, owner)
    // need goes to a third queue, and a lane takes one entry: the owner's slab constants through ds_bpermute (10 dwords; |inv| is
    // recomputed), the chunk's 32 batch boxes through vector loads (768 bytes), 16 packed slab tests with the same arithmetic, and
    // pushes the (batch, owner) entries of level 1. A group of a single chunk keeps the wave-uniform form (every lane needs it).
    constexpr int kChunkCap = kChunkRows * 64;
#ifndef RT_L0_ROUND_COST
#define RT_L0_ROUND_COST 10
#endif
    constexpr int kLevel0RoundCost = RT_L0_ROUND_COST;         // what a round of the shared level 0 costs, in fifths of a wave-uniform chunk (swept: 1.6 .. 4, best at 2)
    int cq_len = 0;
    // all: every entry; otherwise whole rounds only, and what is left stays in the queue for the next call (its round will be fuller,
    // or, at the scan's end, shared by several lanes per entry)
    auto level0_chunks = [&](bool all) {
        const int n_take = all ? cq_len : cq_len / n_here * n_here;
        // few entries: 2, 4 or 8 lanes share an entry, each taking a part of the chunk's words (the mixed 4K frame queues 13 entries
        // per scan: a round in which 51 lanes idle through 16 slab tests would cost what a full one does)
        static_assert(kItemChunkWords == 4 || kItemChunkWords == 8, "a chunk's batches are one 32-bit mask per lane, shared by up to kItemChunkWords lanes");
        const int lshift = n_take * 8 <= n_here && kItemChunkWords >= 8 ? 3 : (n_take * 4 <= n_here ? 2 : (n_take * 2 <= n_here ? 1 : 0));
        const int per = kItemChunkWords >> lshift;             // words per lane
        for (int base = 0; base < (n_take << lshift); base += n_here) {
            const int u = base + rank_here;
            const int e = u >> lshift, part = u & ((1 << lshift) - 1);
            const bool work = e < n_take;
            const uint32_t ent = work ? cq[(e >> 6) * kScanBlock + (e & 63)] : uint32_t(lane);
            const int owner = int(ent & 63u), ci = int(ent >> 6);
            FilterRay fo = fr;                                 // (t_lo depends on t_min alone: the same for every lane)
            fo.I01 = f2{__shfl(fr.I01.x, owner), __shfl(fr.I01.y, owner)}; fo.I2 = f2{__shfl(fr.I2.x, owner), 0.0f};
            fo.A01 = f2{__builtin_fabsf(fo.I01.x), __builtin_fabsf(fo.I01.y)}; fo.A2 = f2{__builtin_fabsf(fo.I2.x), 0.0f};
            fo.KX = f2{__shfl(fr.KX.x, owner), __shfl(fr.KX.y, owner)};
            fo.KY = f2{__shfl(fr.KY.x, owner), __shfl(fr.KY.y, owner)};
            fo.KZ = f2{__shfl(fr.KZ.x, owner), __shfl(fr.KZ.y, owner)};
            const float t_hi_o = __shfl(__double2float_ru(h.t), owner);
            uint32_t cur = 0u;                                 // bit 4 wi + bb: the owner needs batch bb of word wi of the chunk
            if (work) {
                const int fwc = ci * kItemChunkWords;
                const int nwc = n_fwords - fwc < kItemChunkWords ? n_fwords - fwc : kItemChunkWords;
                const float4* __restrict__ bn = reinterpret_cast<const float4*>(tg.bnd) + size_t(fwc) * 6;
                const int w_end = (part + 1) * per < nwc ? (part + 1) * per : nwc;
#pragma unroll 2
                for (int wi = part * per; wi < w_end; ++wi) {
                    const float4 q0 = bn[6 * wi], q1 = bn[6 * wi + 1], q2 = bn[6 * wi + 2], q3 = bn[6 * wi + 3], q4 = bn[6 * wi + 4], q5 = bn[6 * wi + 5];
                    bool a0, a1, a2, a3;
                    float e0, e1;
                    box_pair_v(fo, t_hi_o, f2{q0.x, q0.y}, f2{q1.x, q1.y}, f2{q2.x, q2.y}, f2{q3.x, q3.y}, f2{q4.x, q4.y}, f2{q5.x, q5.y}, a0, a1, e0, e1);
                    box_pair_v(fo, t_hi_o, f2{q0.z, q0.w}, f2{q1.z, q1.w}, f2{q2.z, q2.w}, f2{q3.z, q3.w}, f2{q4.z, q4.w}, f2{q5.z, q5.w}, a2, a3, e0, e1);
                    const int real = tg.n_fbatches - (fwc + wi) * 4;
                    const uint32_t bits = (a0 && 0 < real ? 1u : 0u) | (a1 && 1 < real ? 2u : 0u) | (a2 && 2 < real ? 4u : 0u) | (a3 && 3 < real ? 8u : 0u);
                    cur |= bits << (4 * wi);
                }
            }
            for (;;) {
                const unsigned long long pushing = __ballot(cur != 0u);
                if (pushing == 0ull) break;
                const int cnt = __popcll(pushing);
                if (bq_len + cnt > kBatchCap) filter_batches(false);
                if (cur != 0u) {
                    const int p = __builtin_ctz(cur);
                    cur &= cur - 1u;
                    const int slot = bq_len + __popcll(pushing & ((1ull << lane) - 1ull));
                    wq[(slot >> 6) * kScanBlock + (slot & 63)] = (uint32_t(ci * kItemChunkWords * 4 + p) << 6) | uint32_t(owner);
                }
                bq_len += cnt;
            }
        }
        keep_rest(cq, n_take, cq_len);
    };
    {
        // Which way a scan goes is decided from the chunk boxes (tested wave-uniformly for 32 chunks at a time, a bit per chunk and
        // lane): a round of the shared form costs about as much as 2 chunks of the wave-uniform form (16 slab tests, 768 bytes of
        // vector loads in four dependent groups, the pushes), so it is taken when its rounds replace more chunks than that - the
        // mixed 4K frame: 13 (chunk, lane) needs spread over 4.1 chunks per scan - and not when many lanes want the same few
        // chunks - the 9.8k-triangle mesh alone: 88 needs over 4.2 chunks.
        const bool one_chunk_s = n_fwords <= kItemChunkWords;
        const int n_chunks = (n_fwords + kItemChunkWords - 1) / kItemChunkWords;
        auto chunk_uniform = [&](int ci) {                     // level 0 of one chunk, wave-uniform
            const int fw0 = ci * kItemChunkWords;
            const int nw = n_fwords - fw0 < kItemChunkWords ? n_fwords - fw0 : kItemChunkWords;
            filter_words_level0(tg, fr, __double2float_ru(h.t), fw0, nw, [&](int fw, const bool (&nd)[4], int real) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const bool need = nd[bb] && bb < real;
                    const unsigned long long pushing = __ballot(need);
                    if (pushing == 0ull) continue;
                    const int cnt = __popcll(pushing);
                    if (bq_len + cnt > kBatchCap) filter_batches(false);
                    if (need) {
                        const int slot = bq_len + __popcll(pushing & ((1ull << lane) - 1ull));
                        wq[(slot >> 6) * kScanBlock + (slot & 63)] = (uint32_t(fw * 4 + bb) << 6) | uint32_t(lane);
                    }
                    bq_len += cnt;
                }
            });
        };
        for (int cg = 0; cg < n_chunks; cg += 32) {
            const int ng = n_chunks - cg < 32 ? n_chunks - cg : 32;
            uint32_t my_chunks = one_chunk_s ? 1u : 0u;        // bit c: this lane needs chunk cg + c
            int n_needs = one_chunk_s ? n_here : 0, n_union = one_chunk_s ? 1 : 0;   // wave-uniform
            if (!one_chunk_s) {
                const float t_hi = __double2float_ru(h.t);
                for (int q = 0; 2 * q < ng; ++q) {
                    cf2p cb = RT_CONST_F2(tg.cbnd) + ((cg >> 1) + q) * 6;
                    bool n0, n1;
                    box_pair(fr, t_hi, cb[0], cb[1], cb[2], cb[3], cb[4], cb[5], n0, n1);
                    n1 = n1 && 2 * q + 1 < ng;
                    const unsigned long long b0 = __ballot(n0), b1 = __ballot(n1);
                    my_chunks |= (n0 ? 1u : 0u) << (2 * q) | (n1 ? 2u : 0u) << (2 * q);
                    n_needs += __popcll(b0) + __popcll(b1);
                    n_union += (b0 != 0ull ? 1 : 0) + (b1 != 0ull ? 1 : 0);
                }
            }
            RT_STAMP(1);
            if (n_union == 0) continue;
            // (a round in which 2, 4 or 8 lanes share an entry is priced at 0.6, 0.4, 0.3 of a full one)
            const int rounds = (n_needs + n_here - 1) / n_here;
            const int tenths = n_needs * 8 <= n_here ? 3 : (n_needs * 4 <= n_here ? 4 : (n_needs * 2 <= n_here ? 6 : 10 * rounds));
            const bool shared = !one_chunk_s && tenths * kLevel0RoundCost < n_union * 50;
#ifdef RT_STAMP_LEVEL0
            if (shared) RT_EVENT(7);                           // (diagnostic: scans that take the shared form)
            if (rank_here == 0) { stp.cnt[1] += n_union; stp.cnt[2] += n_needs; }   // (diagnostic: chunks some lane needs, (chunk, lane) needs: per scan, on top of the per-ray counts)
#endif
            for (int c = 0; c < ng; ++c) {
                const unsigned long long needs = __ballot((my_chunks >> c) & 1u);
                if (needs == 0ull) continue;
                if (shared) {
                    const int cnt = __popcll(needs);
                    if (cq_len + cnt > kChunkCap) level0_chunks(false);
                    if ((needs >> lane) & 1ull) {
                        const int slot = cq_len + __popcll(needs & ((1ull << lane) - 1ull));
                        cq[(slot >> 6) * kScanBlock + (slot & 63)] = (uint32_t(cg + c) << 6) | uint32_t(lane);
                    }
                    cq_len += cnt;
                    if (cq_len >= 64) level0_chunks(false);
                } else {
                    chunk_uniform(cg + c);
                }
                RT_STAMP(1);
                if (bq_len >= 64) { filter_batches(false); RT_STAMP(2); }
                if (q_len >= kItemFlush) { walk_queue(false); RT_STAMP(3); }
            }
            level0_chunks(true);
            RT_STAMP(1);
        }
        filter_batches(true);
        RT_STAMP(2);
        walk_queue(true);
        RT_STAMP(3);
    }
#endif
}

// Triangles of one ray through the BVH (the acceleration structure the reference planned over its
// aabb / bounding_box hooks, shapes.h:14-44, hitPosition.cpp:24-53, and never shipped). Every lane walks the tree on its
// own: a node holds the boxes of its two children, both tested with one packed slab test (box_pair_v); a leaf is one
// item of up to two triangles, tested with tri_t. The walk is conservative, so the result is still hitFromList's: a leaf
// box encloses its triangles' bounding spheres, every accepted hit point lies inside its triangle's sphere by definition of
// the primitive (ray_amd_tri.h), and the boxes are widened by m = 2e-5 (|o| + extent), far above the FP32 rounding of the
// hit point, of the slab arithmetic and of the clamp of tiny direction components (< 1e-6 of that scale).
// tri_t keeps the list-order tie rule through the flattened index, so the visiting order does not matter.
// The stack (node indices) is the lane's column of the flag words in LDS: at most kMaskWords entries (the builder
// balances the tree and refuses scenes that would need more).
__device__ __forceinline__ void scan_triangles_bvh(const DevScene& sc, const Ray& r, double t_min, Hit& h, uint32_t* words) {
    auto visit = [&](int item) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int slot = 2 * item + hh;
            const int idx = sc.tri_prim[slot];
            double t;
            if (idx >= 0 && tri_t(r, sc.tri + 16 * slot, t_min, h.t, idx, h.prim, &t)) { h.t = t; h.prim = idx; }
        }
    };
    const V3 orel = r.o - v3(sc.tg.gx, sc.tg.gy, sc.tg.gz);
    const float omax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(orel.x), __builtin_fabsf(orel.y)), __builtin_fabsf(orel.z));
    FilterRay br;
    set_box_ray(br, orel, r.d, r.dl2, 2.0e-5f * (omax + sc.tri_extent) + 1.0e-30f, t_min);
    int node = 0, sp = 0;
    for (;;) {
        const float4* __restrict__ q = sc.tbvh + 4 * node;
        const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        bool n0, n1;
        float e0, e1;
        box_pair_v(br, __double2float_ru(h.t), f2{q0.x, q0.y}, f2{q0.z, q0.w}, f2{q1.x, q1.y}, f2{q1.z, q1.w}, f2{q2.x, q2.y}, f2{q2.z, q2.w}, n0, n1, e0, e1);
        const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
        if (n0 && c0 < 0) { for (int k = 0; k < sc.tbvh_leaf_items; ++k) visit(~c0 + k); n0 = false; }
        if (n1 && c1 < 0) { for (int k = 0; k < sc.tbvh_leaf_items; ++k) visit(~c1 + k); n1 = false; }   // tested against the t range before the first leaf: a superset
        if (n0 && n1) {                                    // nearer child first, the other one on the stack
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
}

template <class T>
__device__ __forceinline__ T load_kernarg(size_t offset) {
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    T v;   // copied through a pointer to T: the copy is known to be aligned, which scalar loads need
    __builtin_memcpy(&v, (const T __attribute__((address_space(4)))*)(p + offset), sizeof(T));
    return v;
}


// Fast scan: spheres through rt_scan.hpp (filter + exact), then rects and triangles. Lanes whose ray
// is outside the fast range (non-finite, huge, zero direction), a scene that is, or a NaN t from a
// rect, fall back to the list-order scan, which is the reference loop itself.
// RECTS: 0 = the rects through the direct loop (scan_rects), 1 = through the flag pass + walk (flag_rects; the
// launcher picks this instance for scenes with >= kRectFlagMin rects), 2 = choose at run time (the one-ray kernels).
// A compile-time choice for the render kernel: the second path's code costs the 485-sphere frame 3 % when it is
// merely present.
// MEDIA: 0 = no constant_medium in the scene (its code is compiled out), 1 = there is one: list-order scan only,
// 2 = decide at run time (the one-ray kernels). A compile-time choice for the render kernel for the same reason as RECTS:
// the medium's code merely being present costs the other workloads registers (16 VGPRs spilled to scratch).
// TRIS: 0 = the triangle group through the flat box / filter / walk levels, 1 = through the BVH (the launcher picks it
// TRIS 3 = the launcher knows that the scene has no triangles: their scan is compiled out (fewer registers for the sphere scenes)
// when rt_upload_scene built one), 2 = decide at run time.
// KARG: the caller is render_kernel, whose first argument is the scene: every group's fields are read from the kernarg segment
// where its scan starts (load_kernarg), so that they occupy scalar registers during that scan only.
template <int RECTS = 2, int MEDIA = 2, int TRIS = 2, bool KARG = false>
__device__ __forceinline__ Hit closest_hit(const DevScene& sc, const Ray& r, double t_min, double t_max, Rng& g, uint32_t* words RT_STAMP_PARAM) {
    Hit h; h.t = t_max; h.prim = -1;
    if (MEDIA == 1) return closest_hit_listorder<true>(sc, r, t_min, t_max, g);
    bool ok = sc.fast_ok && ray_in_fast_range(r.o, r.d, r.dl2) && t_min == t_min && t_max == t_max;
    if (ok) {
        HitState hs;
        hit_state_init(hs, t_max);
        bool sph_redo = false;
        {
            const FilterGroup sg_here = KARG ? load_kernarg<FilterGroup>(offsetof(DevScene, sg)) : sc.sg;
            scan_spheres(sg_here, sc.sph_prim, r.o, r.d, r.dl2, t_min, t_max, hs, words, sph_redo RT_STAMP_ARG);
        }
        h.t = hs.bt; h.prim = hs.idx;
        bool saw_nan = sph_redo;
        if (sc.n_rect > 0) {
        const DevScene scr = KARG ? load_kernarg<DevScene>(0) : sc;   // the rects' arrays, read only by scenes that have rects
        if (RECTS == 1 || (RECTS == 2 && scr.n_rect >= kRectFlagMin)) {
            if (r.d.x == 0.0f || r.d.y == 0.0f || r.d.z == 0.0f) saw_nan = true;   // axis-parallel: the list-order loop
            FilterRay br;
            const float omax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(r.o.x), __builtin_fabsf(r.o.y)), __builtin_fabsf(r.o.z));
            set_box_ray(br, r.o, r.d, r.dl2, 2.0e-5f * (omax + scr.rect_extent) + 1.0e-30f, t_min);
            for (int c0 = 0; c0 < scr.n_rect && !saw_nan; c0 += 32 * kMaskWords) {
                const int c1 = c0 + 32 * kMaskWords < scr.n_rect ? c0 + 32 * kMaskWords : scr.n_rect;
                const uint32_t summary = flag_rects(scr, br, __double2float_ru(h.t), c0, c1, words);
                RT_STAMP(4);
                walk_flagged(summary, c0 >> 5, words, [&](int j) {
                    // the ray's components along the rect's normal and in its plane, picked with bit masks: a select
                    // chain over ray components is turned into an indexed load from a scratch copy of the ray
                    const int axis = scr.rect_axis[j];
                    uint32_t mxy = axis == 0 ? 0xffffffffu : 0u, mxz = axis <= 1 ? 0xffffffffu : 0u;
                    asm("" : "+v"(mxy), "+v"(mxz));
                    auto pick = [](uint32_t m, float a, float b) { return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m)); };
                    const float ok = pick(mxy, r.o.z, pick(mxz, r.o.y, r.o.x)), dk = pick(mxy, r.d.z, pick(mxz, r.d.y, r.d.x));
                    const float oa = pick(mxz, r.o.x, r.o.y), da = pick(mxz, r.d.x, r.d.y);
                    const float ob = pick(mxy, r.o.y, r.o.z), db = pick(mxy, r.d.y, r.d.z);
                    const double* q = scr.rect + 5 * j;
                    const int idx = scr.rect_prim[j];
                    double t;
                    if (rect_core(ok, dk, oa, da, ob, db, q[0], q[1], q[2], q[3], q[4], t_min, h.t, idx, h.prim, &t)) {
                        if (t != t) saw_nan = true; else { h.t = t; h.prim = idx; }
                    }
                } RT_STAMP_ARG);
            }
        } else if (RECTS != 1 && scr.n_rect > 0) {
            const float tminf = __double2float_rd(t_min);
            float btf = __double2float_ru(h.t);
            scan_rects<0>(scr.rect, scr.rect_f, scr.rect_prim, 0, scr.n_rect_xy, r, t_min, tminf, h.t, h.prim, btf, saw_nan);
            scan_rects<1>(scr.rect, scr.rect_f, scr.rect_prim, scr.n_rect_xy, scr.n_rect_xy + scr.n_rect_xz, r, t_min, tminf, h.t, h.prim, btf, saw_nan);
            scan_rects<2>(scr.rect, scr.rect_f, scr.rect_prim, scr.n_rect_xy + scr.n_rect_xz, scr.n_rect, r, t_min, tminf, h.t, h.prim, btf, saw_nan);
        }
        }
        if (TRIS != 3 && !saw_nan && (sc.tg.n_epairs | sc.tg.n_fpairs) != 0) {
            const DevScene sc_tri = KARG ? load_kernarg<DevScene>(0) : sc;
            if (TRIS == 1 || (TRIS == 2 && sc.tbvh != nullptr)) scan_triangles_bvh(sc_tri, r, t_min, h, words);
            else scan_triangles(sc_tri, r, t_min, h, words, saw_nan RT_STAMP_ARG);
        }
        ok = !saw_nan;
    }
    if (!ok) h = closest_hit_listorder<MEDIA != 0>(sc, r, t_min, t_max, g);
    RT_STAMP(4);
    return h;
}

// hit_record fields of the winning primitive (shapes.h:98-108, 182-191; hitPosition.h:104-108)
struct HitRec {
    V3 p, normal;
    bool front_face;
};
__device__ __forceinline__ HitRec hit_record(const DevScene& sc, const Ray& r, const Hit& h) {
    HitRec rec;
    int k = sc.prim_kind[h.prim], j = sc.prim_local[h.prim];
    rec.p = ray_at(r, h.t);
    V3 outward;
    if (k == 0) {
        float4 s = sc.sph[j];
        outward = div3(rec.p - v3(s.x, s.y, s.z), s.w);     // (at - center) / radius, shapes.h:99
    } else if (k == 1) {
        int axis = sc.rect_axis[j];
        outward = axis == 0 ? v3(0, 0, 1) : (axis == 1 ? v3(0, 1, 0) : v3(1, 0, 0));
    } else if (k == 3) {                         // constant_medium: "arbitrary" normal and face, shapes.h:439-440
        rec.normal = v3(1, 0, 0); rec.front_face = true;
        return rec;
    } else {
        const float* q = sc.tri + 16 * j;
        outward = v3(q[9], q[10], q[11]);
    }
    rec.front_face = dot(r.d, outward) < 0.0f;
    rec.normal = rec.front_face ? outward : -outward;
    return rec;
}

// hittable_pdf over the `lights` shape (hitPosition.cpp:55-61): sphere::pdf_value / random
// (shapes.h:117-135), rectangle / xz_rect (shapes.h:202-219, 268-284), base class otherwise.
__device__ __forceinline__ double light_pdf_value(const DevScene& sc, V3 o, V3 v) {
    if (sc.light_kind == RT_PRIM_SPHERE) {
        Ray r = make_ray(o, v);
        float4 s = make_float4(sc.light_c[0], sc.light_c[1], sc.light_c[2], sc.light_r);
        // sphere::hit(Ray(o, v), 0.001, infinity) is only asked WHETHER it hits (shapes.h:119-121): FP32
        // bounds on the roots decide it unless a root is within rounding of 0.001
        V3 oc = r.o - v3(s.x, s.y, s.z);
        float a = r.dl2, b = dot(oc, r.d), c = length_squared(oc) - s.w * s.w;
        float D = b * b - a * c;
        int verdict = -1;   // 1 hit, 0 miss, -1 undecided
        if (a >= kMinA && a <= kMaxA && __builtin_fabsf(b) <= 1.0e18f && __builtin_fabsf(D) <= 1.0e36f) {
            if (D < 0.0f) verdict = 0;
            else {
                float nb = -b, ia = __builtin_amdgcn_rcpf(a), sf = __builtin_amdgcn_sqrtf(D);
                float e = (__builtin_fabsf(nb) + sf) * ia * 2.0e-6f;
                if ((nb - sf) * ia - e > 0.0010001f) verdict = 1;          // near root certainly >= t_min
                else if ((nb + sf) * ia + e < 0.0009999f) verdict = 0;     // far root certainly < t_min
            }
        }
        if (verdict < 0) { double t; verdict = sphere_t(r, s, 0.001, kInf, 0, -1, &t) ? 1 : 0; }
        if (verdict == 0) return 0.0;
        V3 co = v3(s.x, s.y, s.z) - o;
        double cos_theta_max = sqrt64(double(1 - s.w * s.w / length_squared(co)));
        double solid_angle = 2 * kPi * (1 - cos_theta_max);
        if (recip_range_ok(solid_angle)) return div64(1.0, recip64(solid_angle));
        return 1 / solid_angle;
    }
    if (sc.light_kind == RT_PRIM_XY_RECT || sc.light_kind == RT_PRIM_XZ_RECT) {
        Ray r = make_ray(o, v);
        double t;
        int axis = sc.light_kind == RT_PRIM_XY_RECT ? 0 : 1;
        if (!rect_t(r, axis, sc.light_rect, 0.001, kInf, 0, -1, &t)) return 0.0;
        V3 outward = axis == 0 ? v3(0, 0, 1) : v3(0, 1, 0);
        V3 normal = (dot(r.d, outward) < 0.0f) ? outward : -outward;
        const double* q = sc.light_rect;
        double area = (q[1] - q[0]) * (q[3] - q[2]);
        double distance_squared = t * t * double(length_squared(v));
        double cosine = __builtin_fabs(double(dot(v, normal) / length(v)));
        return distance_squared / (cosine * area);
    }
    return 0.0;
}
__device__ __forceinline__ V3 light_random(const DevScene& sc, Rng& g, V3 o) {
    if (sc.light_kind == RT_PRIM_SPHERE) {
        V3 direction = v3(sc.light_c[0], sc.light_c[1], sc.light_c[2]) - o;
        float distance_squared = length_squared(direction);
        Onb uvw;
        uvw.build_from_w(direction);
        return uvw.local(random_to_sphere(g, double(sc.light_r), double(distance_squared)));
    }
    if (sc.light_kind == RT_PRIM_XY_RECT || sc.light_kind == RT_PRIM_XZ_RECT) {
        // QVector3D(random_double(a0,a1), k, random_double(b0,b1)) for both kinds (reference quirk:
        // the XY rect also puts k in the y slot); the (b0,b1) draw comes first in the reference binary
        const double* q = sc.light_rect;
        float zc = g.range(float(q[2]), float(q[3]));
        float xc = g.range(float(q[0]), float(q[1]));
        return v3(xc, float(q[4]), zc) - o;
    }
    return v3(1, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// calc::ray_color (calc.cpp:90-131) as a loop. The reference recursion computes
//     L_k = emitted + (attenuation * float(scattering_pdf) * L_{k+1}) / float(pdf)      (diffuse)
//     L_k = attenuation * L_{k+1}                                                      (specular)
// so each level pushes (w = attenuation [* float(spdf)], pdf) on a per-thread stack in HBM and the
// radiance is unwound from the terminal value in the reference's own order of operations:
// bit-identical per-sample colours, not just "within tolerance". A specular level is stored with
// pdf = -1 (a mixture pdf value is >= 0 or NaN, never -1).
// ------------------------------------------------------------------------------------------------
// The first kLdsLevels levels of every thread's stack live in LDS ([level][thread] float4, 4 KB per
// level and block), deeper levels in HBM ([level][thread] over the whole grid): with depth <= 6 a frame
// writes nothing but its accumulators to HBM.
constexpr int kLdsLevels = 6;
struct PathStack {
    float4* lds;                 // this thread's column of the block's LDS stack (stride kScanBlock)
    float4* __restrict__ base;   // HBM: [depth - kLdsLevels][stride] entries (w.xyz, pdf)
    uint32_t stride, tid;
    RT_DEV void push(int level, V3 w, float pdf) {
        const float4 e = make_float4(w.x, w.y, w.z, pdf);
        if (level < kLdsLevels) lds[level * kScanBlock] = e;
        else base[size_t(level - kLdsLevels) * stride + tid] = e;
    }
    RT_DEV float4 get(int level) const {
        if (level < kLdsLevels) return lds[level * kScanBlock];
        return base[size_t(level - kLdsLevels) * stride + tid];
    }
};
constexpr float kSpecularMark = -1.0f;

// double(x) / pi with the reciprocal of pi refined once per kernel (recip64); `/` for inf and NaN
RT_DEV double div_by_pi(float x, const Recip64& rpi) {
    if (__builtin_fabsf(x) <= 3.0e38f) return div64(double(x), rpi);
    return double(x) / kPi;
}

// One bounce: the body of ray_color after hitFromList returned `h` for `ray`. Returns true when the
// path ends here with terminal radiance L (miss, light, normal view); otherwise pushes this level's
// weight and replaces `ray` by the scattered / specular ray.
// ENDS_ONLY: the caller knows that the path ends here (miss, normal view, light) and wants only that part of the code.
template <bool ENDS_ONLY = false>
__device__ __forceinline__ bool shade_bounce(const DevScene& sc, Ray& ray, const Hit& h, Rng& g, bool normals, PathStack& st, int level,
                                             const Recip64& rpi, V3& L) {
    if (h.prim < 0) { L = v3(sc.bg[0], sc.bg[1], sc.bg[2]); return true; }      // calc.cpp:100-103
    if (normals) {                                                          // calc.cpp:105-109
        V3 N = normalized(ray_at(ray, h.t) - v3(0.0f, 0.0f, -1.0f));
        L = 0.5f * v3(N.x + 1.0f, N.y + 1.0f, N.z + 1.0f);
        return true;
    }
    HitRec rec = hit_record(sc, ray, h);
    int m = sc.prim_mat[h.prim];
    int mk = sc.mat_kind[m];
    V3 rgb = v3(sc.mat_rgb[3 * m], sc.mat_rgb[3 * m + 1], sc.mat_rgb[3 * m + 2]);
    if (mk == RT_MAT_DIFFUSE_LIGHT || ENDS_ONLY) { L = rgb; return true; }   // hitPosition.h:301-310, calc.cpp:115-116
    V3 w;
    float pdf = kSpecularMark;
    V3 unit_direction = v3(0.0f, 0.0f, 0.0f);
    if (mk == RT_MAT_METAL || mk == RT_MAT_DIELECTRIC) unit_direction = normalized(ray.d);   // hitPosition.h:233,261
    if (mk == RT_MAT_ISOTROPIC) {                                           // hitPosition.h:327-333, read as specular
        ray = make_ray(rec.p, random_in_unit_sphere(g));
        w = rgb;
    } else if (mk == RT_MAT_METAL) {                                        // hitPosition.h:230-239
        V3 reflected = reflect(unit_direction, rec.normal);
        float fuzz = sc.mat_fuzz[m];
        ray = make_ray(rec.p, reflected + fuzz * random_in_unit_sphere(g));
        w = rgb;
    } else if (mk == RT_MAT_DIELECTRIC) {                                   // hitPosition.h:253-287
        const double inv_ir = sc.mat_inv_ir[m], ir = sc.mat_ir[m];   // both loaded: a select of the two POINTERS becomes a table in scratch
        double refraction_ratio = rec.front_face ? inv_ir : ir;
        float cos_theta = float(__builtin_fmin(double(dot(-unit_direction, rec.normal)), 1.0));
        float sin_theta = float(sqrt64(double(1.0f - cos_theta * cos_theta)));
        bool cannot_refract = refraction_ratio * double(sin_theta) > 1.0;
        bool do_reflect = cannot_refract;
        if (!do_reflect) {
            float ref_idx = float(refraction_ratio);
            float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
            r0 = r0 * r0;
            float refl = r0 + (1.0f - r0) * float(pow5_portable(double(1.0f - cos_theta)));
            do_reflect = refl > g.u01();
        }
        V3 direction = do_reflect ? reflect(unit_direction, rec.normal)
                                  : refract(unit_direction, rec.normal, refraction_ratio);
        ray = make_ray(rec.p, direction);
        w = v3(1.0f, 1.0f, 1.0f);
    } else {                                                                // lambertian, hitPosition.h:202-216
        const V3 wn = normalized(rec.normal);                               // cosine_pdf's onb.w, hitPosition.h:143,129
        const bool pick_light = g.u01() < 0.5f;                             // mixture_pdf::generate, hitPosition.cpp:67-72
        V3 dir;
        if (sc.light_kind == RT_PRIM_SPHERE) {
            // Both branches (sphere::random, shapes.h:129-135 + calc.h:75-86; cosine_pdf::generate,
            // hitPosition.h:152-155 + calc.h:62-73) draw r1, r2, take sin/cos of 2*pi*r1, two square
            // roots and an orthonormal basis: one copy of that code, operands selected per lane.
            const float r1 = g.u01(), r2 = g.u01();
            const V3 to_light = v3(sc.light_c[0], sc.light_c[1], sc.light_c[2]) - rec.p;
            Onb sb;
            sb.build_from_w(pick_light ? to_light : rec.normal);
            const double radius = double(sc.light_r), distance_squared = double(length_squared(to_light));
            const double s1 = sqrt64(pick_light ? 1 - radius * radius / distance_squared : double(1 - r2));
            const double z = pick_light ? 1 + r2 * (s1 - 1) : s1;
            const double s2 = sqrt64(pick_light ? 1 - z * z : double(r2));
            const double phi = 2 * kPi * r1;
            double sn, cs;
            sincos_portable(phi, &sn, &cs);
            dir = sb.local(v3(float(cs * s2), float(sn * s2), float(z)));
        } else {
            if (pick_light) dir = light_random(sc, g, rec.p);
            else { Onb uvw; uvw.build_from_w(rec.normal); dir = uvw.local(random_cosine_direction(g)); }
        }
        V3 ndir = normalized(dir);
        float cosine_s = dot(rec.normal, ndir);                             // lambertian::scattering_pdf
        double spdf = cosine_s < 0 ? 0.0 : div_by_pi(cosine_s, rpi);
        float cosine_c = dot(ndir, wn);                                     // cosine_pdf::value
        double cval = (cosine_c <= 0) ? 0.0 : div_by_pi(cosine_c, rpi);
        double pval = 0.5 * light_pdf_value(sc, rec.p, dir) + 0.5 * cval;   // mixture_pdf::value
        ray = make_ray(rec.p, dir);
        w = rgb * float(spdf); pdf = float(pval);
        if (pdf == kSpecularMark) pdf = -pdf;   // unreachable (pval >= 0 or NaN); keeps the mark unambiguous
    }
    st.push(level, w, pdf);
    return false;
}

// Unwind `level` pushed bounces from the terminal radiance L, innermost first. The stack entries are
// fetched four at a time so that a path pays one HBM round trip per four levels, not one per level.
RT_DEV V3 unwind_step(float4 e, V3 L) {
    V3 w = v3(e.x, e.y, e.z);
    if (e.w == kSpecularMark) return w * L;                                  // calc.cpp:119-120
    return div3_zero_plus(w * L, e.w);                                       // calc.cpp:129-130: emitted (= 0) + attenuation * spdf * L / pdf
}
__device__ __forceinline__ V3 unwind_path(const PathStack& st, int level, V3 L) {
    while (level > 0) {
        const float4 e0 = st.get(level - 1);
        const float4 e1 = st.get(level > 1 ? level - 2 : 0);
        const float4 e2 = st.get(level > 2 ? level - 3 : 0);
        const float4 e3 = st.get(level > 3 ? level - 4 : 0);
        L = unwind_step(e0, L);
        if (level > 1) L = unwind_step(e1, L);
        if (level > 2) L = unwind_step(e2, L);
        if (level > 3) L = unwind_step(e3, L);
        level -= 4;
    }
    return L;
}

// ray_color for one ray, simple per-thread loop (rt_trace_rays).
__device__ __forceinline__ V3 trace_path(const DevScene& sc, Ray ray, Rng& g, int depth, bool normals, PathStack& st, uint32_t* words) {
#ifdef RT_STAMPS
    Stamps stp{};
#endif
    int level = 0;
    const Recip64 rpi = recip64(kPi);
    V3 L = v3(0.0f, 0.0f, 0.0f);
    for (int depth_left = depth; depth_left > 0; --depth_left) {            // depth_left <= 0: black, calc.cpp:96-97
        Hit h = closest_hit(sc, ray, double(0.001f), kInf, g, words RT_STAMP_ARG);   // calc.cpp:99
        if (shade_bounce(sc, ray, h, g, normals, st, level, rpi, L)) break;
        ++level;
        L = v3(0.0f, 0.0f, 0.0f);
    }
    return unwind_path(st, level, L);
}

// ================================================================================================
// Kernels
// ================================================================================================
constexpr int kBlock = kScanBlock;   // 4 waves
constexpr int kTile = 8;             // work unit handed to a wave: one 8x8 pixel tile
constexpr long long kMaxWorkItems = 0x70000000ll;   // the work counter is a 32-bit int; the rest is room for every resident wave's last fetch
constexpr int kMaxSlices = 16;        // sample slices per pixel and call (render_kernel)

struct RenderArgs {
    int W, H, y_begin, y_count, spp_begin, spp_count, depth;
    int band_first, band_stride, tiles_y;          // tile row k covers rows y_begin + (band_first + k * band_stride) * 8 + [0, 8)
    uint64_t seed;
    uint32_t flags;
    float4* __restrict__ accum;
    float4* __restrict__ stack;
    unsigned long long* __restrict__ rays_out;
    unsigned int* __restrict__ work_counter;       // next tile to hand out, zeroed before the launch
    unsigned long long* __restrict__ mismatches;   // VERIFY builds: scans whose fast result differs from the list-order scan
    unsigned long long* stamps;                    // RT_STAMPS diagnostic builds: [12] section cycles and pass counts
    const int* __restrict__ pixels;                // PIXELS kernels: [n_pixels] pixel indices y * W + x to render
    int n_pixels;
    double inv_w, inv_h;                           // 1 / double(float(W)), 1 / double(float(H)), correctly rounded (host): see jitter_div
    // Sample slices (render_kernel "slices"): the samples [spp_begin, spp_begin + spp_count) of a pixel are cut into n_slices
    // consecutive slices, slice k = [spp_begin + slice_begin[k], spp_begin + slice_begin[k + 1]); the units of work are (slice, tile)
    // pairs. n_slices = 1: whole pixels (the PIXELS kernels, RT_RENDER_ACCUMULATE, frames of few samples).
    int n_tiles, n_slices;
    int slice_begin[kMaxSlices + 1];
    // The tail: n_tail_tiles tiles, every tail_stride-th of the frame (tile tt * tail_stride + tail_stride - 1), are not among the n_tiles
    // above. They come after all of those, as tail_chunks units of tail_chunk samples per pixel whose colours go to `tail` and are
    // summed in sample order by tail_sum_kernel: work that is parallel over a pixel's samples too, for the end of the frame.
    int n_tail_tiles, tail_chunk, tail_chunks, tail_stride;
    float* __restrict__ tail;                      // [n_tail_tiles][spp_count][64][3] colour (r, g, b: 12 bytes) of sample s of pixel `in` of tail tile t
    unsigned int* __restrict__ tile_state;         // [n_tiles] low byte: slices of the tile that are finished; bit 8 + k: unit (k, tile) came up before
                                                   // slice k - 1 was finished and belongs to whoever finishes it (zeroed before the launch; null when n_slices == 1)
    int sort_guard;                                // SORT kernels: the scene's scans write flag words (a fourth barrier per trip keeps them off the results)
    // Tiles in order of cost (rt_set_tile_order, render_impl): tile_order[b] is the frame tile of bulk unit tile b, tile_order[n_tiles + t]
    // that of tail tile t (null: the arithmetic order above); tile_cost, in the one-sample pass that measures, receives the rays of
    // every frame tile's samples.
    const int* __restrict__ tile_order;
    unsigned int* __restrict__ tile_cost;
};

// (x + xi) / float(W) of rayview.cpp:80-81 without the FP32 division sequence: float(double(n) * inv) with inv = RN(1 / double(float(W))).
// This IS the IEEE quotient float(n) / float(W): the double product is within 2^-52 (relative) of the exact quotient Q of the two
// floats, and Q is either a float itself or at least 2^-49 Q away from every midpoint between adjacent floats (Q = X / Y 2^e with
// 24-bit integers X, Y: |X / Y - (2 j + 1) / 2^k| = |X 2^k - (2 j + 1) Y| / (Y 2^k) >= 2^-49 unless the numerator is 0, which needs
// 2^k | Y, i.e. Y a power of two, where the product is exact). n is 0 or in [2^-24, 2^31), W in [1, 2^31]: no result is subnormal.
RT_DEV float jitter_div(float n, float w, double inv) {
#ifdef __HIP_DEVICE_COMPILE__
    (void)w;
    return float(double(n) * inv);
#else
    (void)inv;
    return n / w;
#endif
}

// 16 bytes written through to memory (sc1): visible to every XCD once the store has completed (s_waitcnt vmcnt(0)), without a release fence.
// The compiler does not count a store inside an asm statement; the kernel waits for it explicitly before it publishes the unit
// (render_kernel, complete_unit), and an uncounted store in flight only makes the compiler's own vmcnt waits longer, never shorter.
__device__ __forceinline__ void store_float4_sc1(float4* p, float4 v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}
constexpr uint32_t kTailSlot = 4u;   // render_kernel: the `slot` of a lane that renders a pixel of a tail unit
constexpr uint32_t kUnitTileBits = 26, kUnitTileMask = (1u << kUnitTileBits) - 1u;   // a unit as one word: slice << 26 | tile

// The render kernel's arguments as they lie in the kernarg segment. Values that one section of the main loop needs are read from
// there (scalar loads through the constant address space) where they are used, behind a pointer the compiler cannot see through:
// left to itself it loads every argument once at kernel entry and then keeps ~280 scalars alive across the whole loop by spilling
// them to VGPR lanes, and every use costs a v_readlane - a VALU slot in a kernel that is bound by VALU issue; the scalar memory
// pipe is idle.
struct RenderKernArgs {
    DevScene sc;
    DevCamera cam;
    RenderArgs ra;
};
static_assert(offsetof(RenderKernArgs, sc) == 0, "closest_hit<..., KARG> reads the scene at offset 0 of the kernarg segment");

// Persistent waves with path regeneration. Every lane owns one pixel at a time and walks its samples
// in order (the reference sums a pixel's samples in order, rayview.cpp:79-84); each trip of the main
// loop does exactly ONE closest-hit scan for every lane that has a pixel, whatever bounce its path is
// at, so the 64 lanes of a wave stay in the scan together instead of waiting for the longest path.
// A lane whose pixel is finished takes the next pixel of the wave's tile queue; an empty queue is
// refilled with one 8x8 tile from a global counter (one atomic per wave and tile).
// PIXELS: the work items are the entries of a caller-supplied pixel list (RayView::renderOneRay's random
// pixels, rayview.cpp:69-70) instead of the tiles of a row range.
// Sample slices. A lane's unit of work used to be a whole pixel, so a frame ended with every lane finishing the pixel it happened
// to hold: for about one pixel's time lanes ran dry one by one (10 - 20 % of a 1080p frame of 64 samples, and 38 % of the 1/8 band
// share of a 4K frame of 1024 samples, where a tile is 100 ms of a wave's time and a wave gets four). Now the samples of a pixel
// are cut into slices (ra.slice_begin: each half of what is left, the last ones 8 samples) and the units of work are (slice, tile)
// pairs. A lane that takes a pixel of slice k > 0 CONTINUES the pixel's sum: it starts from the accumulator's contents (the sum
// of slices 0 .. k - 1, written by whichever lanes rendered them) and adds its own samples in sample order, so the pixel's sum is
// the same chain of additions ((c0 + c1) + c2) + ... as one lane walking all samples (rayview.cpp:79-84), and the frame is
// bit-identical to the whole-pixel one: a sample's colour depends on (seed, pixel, sample) only.
// Order between the slices of a tile, without anybody ever waiting. The counter hands the units out slice by slice in tile order
// (all tiles' slice 0, then all tiles' slice 1, ...: the pixels a wave holds stay neighbours, which keeps the union of what its
// lanes need in a scan small). All 64 pixels of a unit are taken by lanes of the ONE wave that got it, so "the unit is done" is
// wave-local bookkeeping on ballots (up to four unfinished units per wave, a byte each in left_word), no atomics per pixel.
// tile_state[tile] holds the number of finished slices of the tile. A wave that gets unit (k, tile), k > 0, from the counter
// ORs bit 8 + k into that word (one returning atomic): if the count it reads back is k the unit is ready and the wave renders
// it; if not it has thereby LEFT the unit to whoever finishes slice k - 1, and takes the next unit from the counter. The wave that
// finishes (k - 1, tile) adds 1 to the word (one returning atomic) and, if it reads bit 8 + k back, renders (k, tile) itself, before
// anything else from the counter (its pixels are that tile's anyway). Both are atomics on one word, so exactly one of the two
// sees the other: no unit is lost, none is rendered twice, nobody polls or spins, every wave exits when the counter has run out
// and its own units are done. The XCDs' L2 caches are not coherent and a CU's L1 is never refreshed by another CU's stores, so the
// hand-off follows the write-through recipe: the pixel-slices are stored with 16-byte sc1 stores (no release fence, which would
// write back the whole L2, 2 - 6 us), the finishing wave waits for all its stores (s_waitcnt vmcnt(0)) before its atomic add;
// the wave that takes the unit over does ONE agent-scope acquire (invalidates its CU's L1) before its lanes load the tile's pixels.
// RT_RENDER_CONTINUE does the same across calls: slice 0 then also starts from the accumulator.
#ifdef RT_STAMPS
__device__ unsigned long long g_wave_times[8192][2];
#endif
// CHAIN: false = the launcher knows that every pixel is rendered as one slice (n_slices == 1): the slices' bookkeeping is compiled out
// SORT: the bounces of the block's 256 rays are shaded in order of material (below, "block-sorted shading"): for scenes whose scan is
// cheap against their shading (no filter group, few rects, no triangles, no medium); not with VERIFY or PIXELS
template <bool VERIFY, bool PIXELS = false, int RECTS = 0, int MEDIA = 0, int TRIS = 0, bool CHAIN = true, bool SORT = false>
__global__ __launch_bounds__(kBlock, 4) void render_kernel(DevScene sc, DevCamera cam, RenderArgs ra_entry) {
    static_assert(!SORT || (!VERIFY && !PIXELS && MEDIA == 0 && TRIS == 3 && RECTS == 0), "block-sorted shading: plain frames of scenes without triangles only");
    __shared__ RT_WORDS_ALIGN uint32_t s_words[kMaskWords * kBlock];
    uint32_t* words = s_words + threadIdx.x;
    const int lane = threadIdx.x & 63;
    __shared__ float4 s_stack[kLdsLevels * kBlock];
    PathStack st; st.lds = s_stack + threadIdx.x; st.base = ra_entry.stack; st.stride = gridDim.x * kBlock; st.tid = blockIdx.x * kBlock + threadIdx.x;
    const Recip64 rpi = recip64(kPi);

    bool have = false, path = false;
    int x = 0, y = 0, s = 0, s_end = 0, level = 0, depth_left = 0;
    uint32_t pix = 0, n_rays = 0;
    float wn = 0.0f;               // the accumulator's .w (samples summed so far) of this lane's pixel
    uint32_t slot = 0u;            // which of the wave's unfinished units this lane's pixel belongs to
    V3 sum = v3(0.0f, 0.0f, 0.0f);
    Ray ray = make_ray(v3(0, 0, 0), v3(0, 0, 0));
    Rng g; g.state = 0;
    int q_cur = 0, q_end = 0;      // wave-uniform: the wave's queue of work indices (one unit of 64 at a time), and what the unit is:
    int q_x0 = 0, q_y0 = 0, q_s0 = 0, q_s1 = 0;   // its tile's first pixel, its samples [q_s0, q_s1)
    uint32_t q_slot = 0u;          // the slot of the unit being handed out (kTailSlot: a tail unit, whose colours go to ra.tail at q_at)
    uint32_t q_at = 0u;
    uint32_t mine = 0u;            // bit u: slot u's unit is finished and the tile's NEXT slice (left to this wave by the wave that got it from the
                                   // counter too early) has still to be begun; uid[u] already names it
    // the wave's unfinished units: slot u's pixels still to finish in byte u of left_word (0: the slot is free), what it is in uid[u]
    uint32_t left_word = 0u, uid0 = 0u, uid1 = 0u, uid2 = 0u, uid3 = 0u;
    bool exhausted = false;        // wave-uniform: the global counter has run past the last unit
    uint32_t sort_turn = 0u;       // SORT: this trip's rotation of the slots over the block's waves
#ifdef RT_STAMPS
    Stamps stp{};
    stp.last = __builtin_amdgcn_s_memtime();
    const unsigned long long t_wave_start = stp.last;
#endif

    if (SORT) {                    // the two sets of counters of the block-sorted shading
        if (threadIdx.x < 6u) s_words[2 * 4 * kBlock + 3 + 4u * threadIdx.x] = 0u;
        __syncthreads();
    }
    for (;;) {
        bool finished = false;     // this lane's pixel-slice ended in this trip
        // ---- hand pixels to idle lanes ------------------------------------------------------------
        for (;;) {
            const unsigned long long need = __ballot(!have);
            if (need == 0ull) break;
            const RenderArgs ra = load_kernarg<RenderArgs>(offsetof(RenderKernArgs, ra));   // (every section reads its arguments where it uses them)
            const bool chained = CHAIN && !PIXELS && ra.n_slices > 1;
            // a unit of tiles: its samples, its tile, a slot for the wave's bookkeeping
            auto begin_unit = [&](int k, int tile) {
                const int tiles_x = (ra.W + kTile - 1) / kTile;
                // (indexed through the kernarg pointer: an index into the struct's copy would put the array in scratch memory)
                const size_t sb = offsetof(RenderKernArgs, ra) + offsetof(RenderArgs, slice_begin) + size_t(k) * sizeof(int);
                q_s0 = ra.spp_begin + load_kernarg<int>(sb); q_s1 = ra.spp_begin + load_kernarg<int>(sb + sizeof(int));
                // the b-th tile that is not a tail tile
                int ftile = tile < ra.n_tail_tiles * (ra.tail_stride - 1) ? tile + tile / (ra.tail_stride - 1) : tile + ra.n_tail_tiles;
                if (ra.tile_order) ftile = ra.tile_order[tile];
                const int trow = ftile / tiles_x, tcol = ftile - trow * tiles_x;
                q_x0 = tcol * kTile;
                q_y0 = ra.y_begin + (ra.band_first + trow * ra.band_stride) * kTile;
                q_cur = 0; q_end = 64; q_slot = 0u;
#ifndef RT_EXP_NOSYNC
                if (chained)
#else
                if (false)
#endif
                {
                    const int nx = ra.W - q_x0 < kTile ? ra.W - q_x0 : kTile;
                    const int ny = ra.y_begin + ra.y_count - q_y0 < kTile ? ra.y_begin + ra.y_count - q_y0 : kTile;
                    // a free slot: no pixel of it unfinished, and not kept for an inherited unit that has still to be begun
                    const uint32_t busy = ((left_word & 0xffu) != 0u ? 1u : 0u) | ((left_word & 0xff00u) != 0u ? 2u : 0u) |
                                          ((left_word & 0xff0000u) != 0u ? 4u : 0u) | ((left_word & 0xff000000u) != 0u ? 8u : 0u) | mine;
                    q_slot = uint32_t(__builtin_ctz(~busy));
                    left_word |= uint32_t(nx * ny) << (8u * q_slot);
                    const uint32_t id = (uint32_t(k) << kUnitTileBits) | uint32_t(tile);
                    uid0 = q_slot == 0u ? id : uid0; uid1 = q_slot == 1u ? id : uid1; uid2 = q_slot == 2u ? id : uid2; uid3 = q_slot == 3u ? id : uid3;
                }
            };
            if (q_cur >= q_end) {
                if (chained && mine == 0u && (left_word & 0xffu) != 0u && (left_word & 0xff00u) != 0u && (left_word & 0xff0000u) != 0u && (left_word & 0xff000000u) != 0u)
                { RT_EVENT(1); break; }                              // four units of this wave are unfinished: their lanes go on, the idle ones wait
                if (!PIXELS && mine != 0u) {                         // a unit this wave inherited by finishing the slice before it
                    const uint32_t u = uint32_t(__builtin_ctz(mine));
                    mine &= mine - 1u;
                    const uint32_t id = u == 0u ? uid0 : (u == 1u ? uid1 : (u == 2u ? uid2 : uid3));
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (its pixels were written through by this wave; L1 is not trusted)
                    begin_unit(int(id >> kUnitTileBits), int(id & kUnitTileMask));
                } else {
                    if (exhausted) break;
                    const int bulk_units = ra.n_tiles * ra.n_slices;
                    const int total = PIXELS ? ra.n_pixels : (bulk_units + ra.n_tail_tiles * ra.tail_chunks) * 64;
                    int base = 0;
                    if (lane == 0) base = int(atomicAdd(ra.work_counter, 64u));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= total) { exhausted = true; break; }
                    if (PIXELS) { q_cur = base; q_end = base + 64; }
                    else {
                        const int unit = base >> 6;
                        if (unit >= bulk_units) {   // tail_chunk samples of tail tile tt = every tail_stride-th tile of the frame
                            const int tu = unit - bulk_units;
                            const int tt = tu / ra.tail_chunks, ch = tu - tt * ra.tail_chunks;
                            const int tiles_x = (ra.W + kTile - 1) / kTile;
                            int ftile = tt * ra.tail_stride + ra.tail_stride - 1;
                            if (ra.tile_order) ftile = ra.tile_order[ra.n_tiles + tt];
                            const int trow = ftile / tiles_x, tcol = ftile - trow * tiles_x;
                            q_x0 = tcol * kTile;
                            q_y0 = ra.y_begin + (ra.band_first + trow * ra.band_stride) * kTile;
                            q_s0 = ra.spp_begin + ch * ra.tail_chunk;
                            q_s1 = q_s0 + ra.tail_chunk < ra.spp_begin + ra.spp_count ? q_s0 + ra.tail_chunk : ra.spp_begin + ra.spp_count;
                            q_at = uint32_t(tt) * uint32_t(ra.spp_count) * 64u;
                            q_cur = 0; q_end = 64; q_slot = kTailSlot;
                            continue;
                        }
                        const int k = unit / ra.n_tiles, tile = unit - k * ra.n_tiles;   // (the units come slice by slice)
#ifdef RT_EXP_NOSYNC
                        if (false) {   // timing experiment only: WRONG frames (no order between a pixel's slices)
#else
                        if (k > 0) {
#endif
                            uint32_t old = 0u;
                            if (lane == 0) old = atomicOr(ra.tile_state + tile, 1u << (8 + k));
                            old = uint32_t(__builtin_amdgcn_readfirstlane(int(old)));
                            if (int(old & 0xffu) < k) { RT_EVENT(7); continue; }   // not ready: left to whoever finishes the slice before it
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this CU's L1 may hold the tile's lines as they were slices ago
                        }
                        begin_unit(k, tile);
                    }
                }
            }
            const int rank = __popcll(need & ((1ull << lane) - 1ull));
            const int avail = q_end - q_cur;
            if (!have && rank < avail) {
                const int idx = q_cur + rank;
                bool inside;
                if (PIXELS) {
                    inside = idx < ra.n_pixels;
                    const int p = inside ? ra.pixels[idx] : 0;
                    inside = inside && uint32_t(p) < uint32_t(ra.W) * uint32_t(ra.H);   // an index outside the frame is skipped, never written
                    y = p / ra.W; x = p - y * ra.W;
                } else {
                    const int in = idx & 63;
                    x = q_x0 + (in & 7);
                    y = q_y0 + (in >> 3);
                    inside = x < ra.W && y < ra.y_begin + ra.y_count;
                }
                if (inside) {
                    have = true; path = false;
                    pix = uint32_t(y) * uint32_t(ra.W) + uint32_t(x);
                    s = PIXELS ? ra.spp_begin : q_s0;
                    s_end = PIXELS ? ra.spp_begin + ra.spp_count : q_s1;
                    slot = q_slot;
                    sum = v3(0.0f, 0.0f, 0.0f);
                    wn = 0.0f;
                    if (!PIXELS && q_slot == kTailSlot) {
                        wn = __uint_as_float(q_at + uint32_t(idx & 63));   // (a tail pixel keeps no sums: the register holds its place in ra.tail)
                    } else if (!PIXELS && !(q_s0 == ra.spp_begin && (ra.flags & RT_RENDER_CONTINUE) == 0u)) {   // continue the pixel's sum where the slice before left it. No
                        const float4 o = ra.accum[pix];              // arithmetic on the loaded values here: their first use is the end
                        sum = v3(o.x, o.y, o.z);                     // of the lane's first sample, so nobody waits for this load (the
                        wn = o.w;                                    // line comes from HBM: the slice before wrote it through)
                    }
                }
            }
            const int want = __popcll(need);
            q_cur += want < avail ? want : avail;
        }
        if (!SORT && __ballot(have) == 0ull) break;   // (SORT: the block leaves together, below)
        // ---- start the next sample of the lane's pixel (rayview.cpp:80-83) ----------------------------
        if (have && !path) {
            const RenderArgs ra = load_kernarg<RenderArgs>(offsetof(RenderKernArgs, ra));
            g.key(ra.seed, pix, uint32_t(s));
            float u = jitter_div(float(x) + g.u01(), float(ra.W), ra.inv_w);
            float v = jitter_div(float(y) + g.u01(), float(ra.H), ra.inv_h);
            const DevCamera cam_here = load_kernarg<DevCamera>(offsetof(RenderKernArgs, cam));
            ray = get_ray(cam_here, g, u, v);
            level = 0; depth_left = ra.depth; path = true;
        }
        RT_STAMP(0);
        RT_COUNT(3);
        // ---- one closest-hit scan per lane, then one bounce ------------------------------------------------
        const DevScene sc_scan = load_kernarg<DevScene>(offsetof(RenderKernArgs, sc));    // the groups' arrays and constants (wave-uniform code: scalars)
        bool done = false;
        V3 L = v3(0.0f, 0.0f, 0.0f);
        if (SORT) {
            // Block-sorted shading. shade_bounce is a chain of branches, one per material (lambertian with its light sampling and FP64 sin /
            // cos by far the longest), and a wave executes every branch that one of its 64 rays takes: on the built-in scene all of them in
            // nearly every trip, with a third of the lanes on. Here every lane scans its own ray as before, then the block's rays are
            // sorted by class: a path that ends here (miss, light, normal view) is finished by its own lane; the bounce of every other ray
            // goes as a 12-dword item (ray, t, primitive, RNG state, level, owner) to a slot of the flag-word array (free between scans),
            // lambertian items from one end, the others from the other end; thread p shades the item of slot p -- so a wave's 64 items
            // are of one class, two where the classes meet -- pushing the level's weight on the OWNER's path stack, and writes the new
            // ray, the RNG state, L and `done` back to the slot, where the owner collects them. The arithmetic of a bounce does not
            // depend on the lane that executes it: frames are bit-identical.
            // Two barriers per trip; the four waves of a block therefore run trip by trip together and leave together.
            const RenderArgs ra = load_kernarg<RenderArgs>(offsetof(RenderKernArgs, ra));
            const bool normals = (ra.flags & RT_RENDER_NORMALS) != 0;
            Hit h; h.t = 0.0; h.prim = -1;
            int cls = -1;                                            // no ray in this lane
            if (path) {
                ++n_rays;
                h = closest_hit<RECTS, MEDIA, TRIS, true>(sc_scan, ray, double(0.001f), kInf, g, words RT_STAMP_ARG);   // calc.cpp:99
                cls = 2;
                if (h.prim >= 0 && !normals) {
                    const int mk = sc_scan.mat_kind[sc_scan.prim_mat[h.prim]];
                    cls = mk == RT_MAT_LAMBERTIAN ? 0 : (mk == RT_MAT_DIFFUSE_LIGHT ? 2 : 1);
                }
                if (cls == 2) {                                      // the path ends here (miss, light, normal view): no hand-over for three loads
                    const DevScene sc_shade = load_kernarg<DevScene>(offsetof(RenderKernArgs, sc));
                    done = shade_bounce<true>(sc_shade, ray, h, g, normals, st, level, rpi, L);
                }
            }
            RT_STAMP(4);
            // slots (rows 0 - 11 of the array as three quads per slot): lambertian items from slot 0 upwards, the others (dielectric, metal,
            // isotropic) from slot 255 downwards; a wave takes its run of slots with one LDS atomic per class. The last dword of a
            // slot's third quad is not part of the item: those of slots 0 - 5 are the counters, two sets used in turn, {lambertian,
            // others, lanes that have a ray}. Results: one quad per OWNER (rows 12 - 15), so that the next trip's items, which a wave
            // may write while another still collects, never touch them.
            if (ra.sort_guard) __syncthreads();                      // scenes whose scans write flag words: every scan of the block must be over
            const unsigned long long b0 = __ballot(cls == 0), b1 = __ballot(cls == 1);
            uint32_t* cnt = s_words + 2 * 4 * kBlock + 3 + (sort_turn & 64u ? 12u : 0u);   // counter k of the set at cnt[4 k]
            uint32_t base0 = 0u, base1 = 0u;
            if (lane == 0) {
                if (b0 != 0ull) base0 = atomicAdd(cnt, uint32_t(__popcll(b0)));
                if (b1 != 0ull) base1 = atomicAdd(cnt + 4, uint32_t(__popcll(b1)));
            }
            const unsigned long long with_ray = __ballot(cls >= 0);
            if (lane == 0 && with_ray != 0ull) atomicAdd(cnt + 8, uint32_t(__popcll(with_ray)));
            base0 = uint32_t(__builtin_amdgcn_readfirstlane(int(base0)));
            base1 = uint32_t(__builtin_amdgcn_readfirstlane(int(base1)));
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t dest = cls == 0 ? base0 + uint32_t(__popcll(b0 & below)) : 255u - (base1 + uint32_t(__popcll(b1 & below)));
            float4* xq = reinterpret_cast<float4*>(s_words);         // slot p: quads xq[p], xq[kBlock + p], xq[2 kBlock + p]; owner p's result: xq[3 kBlock + p]
            if (cls == 0 || cls == 1) {
                xq[dest] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
                xq[kBlock + dest] = make_float4(ray.d.y, ray.d.z, __int_as_float(__double2loint(h.t)), __int_as_float(__double2hiint(h.t)));
                uint32_t* third = s_words + 4 * (2 * kBlock + dest);   // (three dwords: the fourth may be a counter)
                third[0] = uint32_t(h.prim); third[1] = g.state; third[2] = uint32_t(level) | (threadIdx.x << 8);
            }
            __syncthreads();
            RT_STAMP(2);
            const uint32_t n_lamb = cnt[0], n_other = cnt[4], n_with_ray = cnt[8];
            if (n_with_ray == 0u) break;                             // no wave of the block has a ray, and none will get one
            if (threadIdx.x < 3u) s_words[2 * 4 * kBlock + 3 + (sort_turn & 64u ? 0u : 12u) + 4u * threadIdx.x] = 0u;   // the next trip's set
            // which wave shades which 64 slots rotates from trip to trip: the first slots hold the expensive class, and a wave stays on its SIMD
            sort_turn = (sort_turn + 64u) & 255u;
            const uint32_t my = (threadIdx.x + sort_turn) & 255u;
            if (my < n_lamb || my + n_other > 255u) {
                const float4 q0 = xq[my], q1 = xq[kBlock + my], q2 = xq[2 * kBlock + my];
                Ray r2 = make_ray(v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y));
                Hit h2; h2.t = __hiloint2double(__float_as_int(q1.w), __float_as_int(q1.z)); h2.prim = __float_as_int(q2.x);
                Rng g2; g2.state = __float_as_uint(q2.y);
                const uint32_t lw = __float_as_uint(q2.z), owner = lw >> 8;
                PathStack so; so.lds = s_stack + owner; so.base = st.base; so.stride = st.stride; so.tid = blockIdx.x * kBlock + owner;
                const DevScene sc_shade = load_kernarg<DevScene>(offsetof(RenderKernArgs, sc));   // materials, lights, per-primitive records
                V3 L2 = v3(0.0f, 0.0f, 0.0f);
                // (none of these materials ends a path: the bounce leaves the new ray -- from the hit point, which the owner computes
                // itself -- the RNG state, and the level's weight on the owner's stack)
                (void)shade_bounce(sc_shade, r2, h2, g2, false, so, int(lw & 0xffu), rpi, L2);
                xq[3 * kBlock + owner] = make_float4(r2.d.x, r2.d.y, r2.d.z, __uint_as_float(g2.state));
            }
            RT_STAMP(5);
            __syncthreads();
            if (cls == 0 || cls == 1) {
                const float4 q = xq[3 * kBlock + threadIdx.x];
                ray = make_ray(ray_at(ray, h.t), v3(q.x, q.y, q.z));   // hit_record's p (shapes.h:98, Ray.h:11) as shade_bounce computes it
                g.state = __float_as_uint(q.w);
            }
            if (ra.sort_guard) __syncthreads();                      // ... and no scan of the next trip before every result has been collected
            RT_STAMP(7);
        } else if (path) {
            const RenderArgs ra = load_kernarg<RenderArgs>(offsetof(RenderKernArgs, ra));
            const bool normals = (ra.flags & RT_RENDER_NORMALS) != 0;
            ++n_rays;
            Hit h = closest_hit<RECTS, MEDIA, TRIS, true>(sc_scan, ray, double(0.001f), kInf, g, words RT_STAMP_ARG);   // calc.cpp:99
            if (VERIFY) {   // scenes with a constant_medium (which draws inside the scan) are rejected for VERIFY by the host
                Hit h2 = closest_hit_listorder<false>(sc, ray, double(0.001f), kInf, g);
                if (h2.prim != h.prim || __double_as_longlong(h2.t) != __double_as_longlong(h.t)) atomicAdd(ra.mismatches, 1ull);
            }
            const DevScene sc_shade = load_kernarg<DevScene>(offsetof(RenderKernArgs, sc));   // materials, lights, per-primitive records
            done = shade_bounce(sc_shade, ray, h, g, normals, st, level, rpi, L);
        }
        if (path) {
            const RenderArgs ra = load_kernarg<RenderArgs>(offsetof(RenderKernArgs, ra));
            if (!done) {
                ++level; --depth_left;
                if (depth_left <= 0) { L = v3(0.0f, 0.0f, 0.0f); done = true; }   // calc.cpp:96-97
            }
            RT_STAMP(5);
            if (done) {
                if (!PIXELS && ra.tile_cost) {   // (the one-sample pass of rt_set_tile_order: this sample's rays, to its frame tile)
                    const int tiles_x = (ra.W + kTile - 1) / kTile;
                    const int trow = (((y - ra.y_begin) >> 3) - ra.band_first) / ra.band_stride;
                    atomicAdd(ra.tile_cost + (trow * tiles_x + (x >> 3)), uint32_t(level) + 1u);
                }
                const V3 colour = unwind_path(st, level, L);
                const bool tail_px = !PIXELS && slot == kTailSlot;
                if (tail_px) {
                    float* tc = ra.tail + 3u * size_t(__float_as_uint(wn) + uint32_t(s - ra.spp_begin) * 64u);
                    tc[0] = colour.x; tc[1] = colour.y; tc[2] = colour.z;
                } else {
                    sum = sum + colour;                                           // rayview.cpp:83
                    wn += 1.0f;                                                   // (samples summed: exact up to 2^24)
                }
                path = false;
                ++s;
                if (s == s_end && tail_px) have = false;
                if (s == s_end && !tail_px) {
                    float4 a = make_float4(sum.x, sum.y, sum.z, wn);
                    if (PIXELS && (ra.flags & RT_RENDER_ACCUMULATE)) {
                        // a pixel may be listed more than once (the reference draws its pixels at random, rayview.cpp:69-70):
                        // every listing adds its (identical) sums with one IEEE add each, in whatever order
                        float* q = reinterpret_cast<float*>(ra.accum + pix);
                        atomicAdd(q, a.x); atomicAdd(q + 1, a.y); atomicAdd(q + 2, a.z); atomicAdd(q + 3, a.w);
                    } else {
                        if (ra.flags & RT_RENDER_ACCUMULATE) {                    // (one slice: the pass's own sum is added to what was there)
                            float4 o = ra.accum[pix];
                            a = make_float4(o.x + sum.x, o.y + sum.y, o.z + sum.z, o.w + float(ra.spp_count));
                        }
#ifdef RT_EXP_NOSYNC   // (tools/: timing experiment without any order between a pixel's slices; WRONG frames)
                        if (false) {}
#else
                        if (CHAIN && !PIXELS && ra.n_slices > 1) store_float4_sc1(ra.accum + pix, a);   // another wave continues this pixel: write-through
#endif
                        else ra.accum[pix] = a;
                    }
                    have = false; finished = true;
                }
            }
            RT_STAMP(6);
        }
        // ---- units whose last pixel-slice ended in this trip: hand the tile's next slice to whoever pops it ------------
        if (!PIXELS && CHAIN) {
            const unsigned long long fin = __ballot(finished);
#ifdef RT_EXP_NOSYNC
            const int n_slices = 1;
#else
            const int n_slices = fin != 0ull ? load_kernarg<int>(offsetof(RenderKernArgs, ra) + offsetof(RenderArgs, n_slices)) : 1;
#endif
            if (n_slices > 1) {
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const int cnt = __popcll(fin & __ballot(slot == u));
                    if (cnt == 0) continue;
                    left_word -= uint32_t(cnt) << (8u * u);
                    if (((left_word >> (8u * u)) & 0xffu) != 0u) continue;
                    const uint32_t id = u == 0u ? uid0 : (u == 1u ? uid1 : (u == 2u ? uid2 : uid3));
                    const int k_next = int(id >> kUnitTileBits) + 1;
                    if (k_next < n_slices) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every pixel-slice of the unit (stores of this wave's lanes) is in memory
                        unsigned int* tile_state = load_kernarg<unsigned int*>(offsetof(RenderKernArgs, ra) + offsetof(RenderArgs, tile_state));
                        uint32_t old = 0u;
                        if (lane == 0) old = atomicAdd(tile_state + (id & kUnitTileMask), 1u);
                        old = uint32_t(__builtin_amdgcn_readfirstlane(int(old)));
                        if (old & (1u << (8 + k_next))) {                      // the counter has handed the next slice out already: it is this wave's now
                            const uint32_t nid = id + (1u << kUnitTileBits);
                            uid0 = u == 0u ? nid : uid0; uid1 = u == 1u ? nid : uid1; uid2 = u == 2u ? nid : uid2; uid3 = u == 3u ? nid : uid3;
                            mine |= 1u << u;                                   // (the slot stays taken until that unit has been begun)
                        }
                    }
                }
            }
        }
    }
#ifdef RT_STAMPS
    if (lane == 0) {   // when this wave started and ended (s_memtime), for the frame's timeline (tools/wave_times.py)
        const unsigned int wid = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
        if (wid < 8192u) { g_wave_times[wid][0] = t_wave_start; g_wave_times[wid][1] = __builtin_amdgcn_s_memtime(); }
    }
    if (ra_entry.stamps) {
        for (int k = 0; k < 8; ++k) if (lane == 0) atomicAdd(ra_entry.stamps + k, stp.t[k]);
        for (int k = 0; k < 4; ++k) atomicAdd(ra_entry.stamps + 8 + k, (unsigned long long)stp.cnt[k]);
    }
#endif
    if (ra_entry.rays_out) {
        unsigned long long v = n_rays;   // per-wave sum, one atomic per wave
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && v) atomicAdd(ra_entry.rays_out, v);
    }
}

// The tail tiles of a frame (render_kernel): one thread per pixel adds the stored colours of its samples in sample order, as
// the lane that renders a whole pixel does (rayview.cpp:79-84), and writes the accumulator the same way. A wave is one tile:
// the 64 lanes read 768 bytes of consecutive colours per sample.
__global__ __launch_bounds__(256) void tail_sum_kernel(RenderArgs ra) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int tt = idx >> 6, in = idx & 63;
    if (tt >= ra.n_tail_tiles) return;
    const int tiles_x = (ra.W + kTile - 1) / kTile;
    const int tile = ra.tile_order ? ra.tile_order[ra.n_tiles + tt] : tt * ra.tail_stride + ra.tail_stride - 1;
    const int x = (tile % tiles_x) * kTile + (in & 7);
    const int y = ra.y_begin + (ra.band_first + (tile / tiles_x) * ra.band_stride) * kTile + (in >> 3);
    if (!(x < ra.W && y < ra.y_begin + ra.y_count)) return;
    const float* __restrict__ c = ra.tail + 3u * (size_t(tt) * size_t(ra.spp_count) * 64u + size_t(in));
    const uint32_t pix = uint32_t(y) * uint32_t(ra.W) + uint32_t(x);
    V3 sum = v3(0.0f, 0.0f, 0.0f);
    float wn = 0.0f;
    if (ra.flags & RT_RENDER_CONTINUE) { const float4 o = ra.accum[pix]; sum = v3(o.x, o.y, o.z); wn = o.w; }
#pragma unroll 8
    for (int s = 0; s < ra.spp_count; ++s) {
        const float* v = c + size_t(s) * 192u;
        sum = sum + v3(v[0], v[1], v[2]);
        wn += 1.0f;
    }
    float4 a = make_float4(sum.x, sum.y, sum.z, wn);
    if (ra.flags & RT_RENDER_ACCUMULATE) {
        const float4 o = ra.accum[pix];
        a = make_float4(o.x + sum.x, o.y + sum.y, o.z + sum.z, o.w + float(ra.spp_count));
    }
    ra.accum[pix] = a;
}

// rt_render_pixels: the 8x8 tile of every listed pixel (the key the list is sorted by before it is rendered); a pixel
// outside the frame gets the key after the last tile, so it ends up at the end of the list, where the render kernel skips it
__global__ void pixel_tile_keys_kernel(int n, int W, int H, int tiles_x, int n_tiles, const int* __restrict__ pixels, uint32_t* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pixels[i];
    uint32_t k = uint32_t(n_tiles);
    if (uint32_t(p) < uint32_t(W) * uint32_t(H)) { const int y = p / W, x = p - y * W; k = uint32_t((y >> 3) * tiles_x + (x >> 3)); }
    keys[i] = k;
}

// rt_set_tile_order: 0, 1, 2, ... (the values the radix sort carries along with the tiles' costs)
__global__ void tile_iota_kernel(int n, int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = i;
}
// ... and the order render_kernel reads, from the tiles sorted by falling cost: every stride-th of them (n_tail in all) is a tail tile,
// as every stride-th tile of the frame is without the sort, so that the tail keeps its share of the frame's work; the others are
// the bulk tiles, heaviest first. order[0 .. n - n_tail): bulk, order[n - n_tail .. n): tail.
__global__ void tile_arrange_kernel(int n, int n_tail, int stride, const int* __restrict__ sorted, int* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int q = (i + 1) / stride;
    if (n_tail > 0 && (i + 1) % stride == 0 && q <= n_tail) { order[n - n_tail + q - 1] = sorted[i]; return; }
    const int before = i / stride < n_tail ? i / stride : n_tail;   // tail tiles among the sorted positions before this one
    order[i - before] = sorted[i];
}

// depth <= 0 or no samples: every sample is black and no ray is traced (calc.cpp:96-97)
__global__ void render_black_kernel(int W, int y_begin, int y_count, int band_first, int band_stride, int tiles_y, int spp_count, uint32_t flags,
                                    float4* __restrict__ accum) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * tiles_y * kTile) return;
    const int r = i / W, x = i - r * W;
    const int y = y_begin + (band_first + (r / kTile) * band_stride) * kTile + r % kTile;
    if (y >= y_begin + y_count) return;
    size_t pix = size_t(y) * W + x;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, float(spp_count));
    if (flags & RT_RENDER_ACCUMULATE) {
        float4 o = accum[pix];
        a = make_float4(o.x + 0.0f, o.y + 0.0f, o.z + 0.0f, o.w + float(spp_count));
    }
    accum[pix] = a;
}

__global__ void render_black_pixels_kernel(int n, int n_frame, const int* __restrict__ pixels, int spp_count, uint32_t flags, float4* __restrict__ accum) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int pix = pixels[i];
    if (uint32_t(pix) >= uint32_t(n_frame)) return;
    if (flags & RT_RENDER_ACCUMULATE) {   // duplicates in the list: one add per listing (see render_kernel)
        float* q = reinterpret_cast<float*>(accum + pix);
        atomicAdd(q, 0.0f); atomicAdd(q + 1, 0.0f); atomicAdd(q + 2, 0.0f); atomicAdd(q + 3, float(spp_count));
    } else {
        accum[pix] = make_float4(0.0f, 0.0f, 0.0f, float(spp_count));
    }
}

__global__ __launch_bounds__(kBlock) void primary_kernel(DevScene sc, DevCamera cam, int W, int H, uint64_t seed,
                                                         int* __restrict__ prim_out, double* __restrict__ t_out) {
    __shared__ RT_WORDS_ALIGN uint32_t s_words[kMaskWords * kBlock];
    uint32_t* words = s_words + threadIdx.x;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
        const int t = threadIdx.x;
        const int x = (tile % tiles_x) * 16 + (t & 7) + 8 * ((t >> 6) & 1);
        const int y = (tile / tiles_x) * 16 + ((t & 63) >> 3) + 8 * (t >> 7);
        if (x >= W || y >= H) continue;
        const uint32_t pix = uint32_t(y) * uint32_t(W) + uint32_t(x);
        Rng g; g.key(seed, pix, 0u);
        float u = (float(x) + 0.5f) / float(W), v = (float(y) + 0.5f) / float(H);
        Ray ray = get_ray(cam, g, u, v);
#ifdef RT_STAMPS
        Stamps stp{};
#endif
        Hit h = closest_hit(sc, ray, double(0.001f), kInf, g, words RT_STAMP_ARG);
        if (prim_out) prim_out[pix] = h.prim;
        if (t_out) t_out[pix] = h.prim >= 0 ? h.t : 0.0;
    }
}

__global__ __launch_bounds__(kBlock) void hit_rays_kernel(DevScene sc, long long n, const float* __restrict__ ro,
                                                          const float* __restrict__ rd, double t_min, double t_max,
                                                          int* __restrict__ prim_out, double* __restrict__ t_out,
                                                          float* __restrict__ p_out, float* __restrict__ n_out,
                                                          uint8_t* __restrict__ ff_out) {
    __shared__ RT_WORDS_ALIGN uint32_t s_words[kMaskWords * kBlock];
    uint32_t* words = s_words + threadIdx.x;
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        Ray ray = make_ray(v3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), v3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]));
#ifdef RT_STAMPS
        Stamps stp{};
#endif
        Rng g; g.key(0ull, uint32_t(i), 0u);   // a constant_medium in the list draws from the ray's own stream
        Hit h = closest_hit(sc, ray, t_min, t_max, g, words RT_STAMP_ARG);
        prim_out[i] = h.prim;
        if (t_out) t_out[i] = h.prim >= 0 ? h.t : 0.0;
        HitRec rec; rec.p = v3(0, 0, 0); rec.normal = v3(0, 0, 0); rec.front_face = false;
        if (h.prim >= 0) rec = hit_record(sc, ray, h);
        if (p_out) { p_out[3 * i] = rec.p.x; p_out[3 * i + 1] = rec.p.y; p_out[3 * i + 2] = rec.p.z; }
        if (n_out) { n_out[3 * i] = rec.normal.x; n_out[3 * i + 1] = rec.normal.y; n_out[3 * i + 2] = rec.normal.z; }
        if (ff_out) ff_out[i] = rec.front_face ? 1 : 0;
    }
}

__global__ __launch_bounds__(kBlock, 4) void trace_rays_kernel(DevScene sc, long long n, const float* __restrict__ ro,
                                                            const float* __restrict__ rd, int depth, uint64_t seed, uint32_t flags,
                                                            float* __restrict__ rgb, float4* __restrict__ stack) {
    __shared__ RT_WORDS_ALIGN uint32_t s_words[kMaskWords * kBlock];
    uint32_t* words = s_words + threadIdx.x;
    __shared__ float4 s_stack[kLdsLevels * kBlock];
    PathStack st; st.lds = s_stack + threadIdx.x; st.base = stack; st.stride = gridDim.x * kBlock; st.tid = blockIdx.x * kBlock + threadIdx.x;
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        Ray ray = make_ray(v3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), v3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]));
        Rng g; g.key(seed, uint32_t(i), 0u);
        V3 c = trace_path(sc, ray, g, depth, (flags & RT_RENDER_NORMALS) != 0, st, words);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}

// calc::rgbPerSamples (calc.cpp:133-154) + RayView::writeToImg (rayview.cpp:57-65,86)
__global__ void tonemap_kernel(int W, int H, const float4* __restrict__ accum, int n_samples, uint32_t* __restrict__ rgb32) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int x = i % W, y = i / W;
    float4 a = accum[i];
    int n = n_samples > 0 ? n_samples : int(a.w);
    float scale = 1.0f / float(n);
    float c[3] = {a.x, a.y, a.z};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float r = float(__builtin_sqrt(double(scale * c[k])));
        r = (r < 0.0f) ? 0.0f : (0.999f < r) ? 0.999f : r;
        float v = 255.0f * r;
        q[k] = (v != v) ? 0u : (uint32_t(int(v)) & 255u);   // NaN -> 0 (undefined in the reference)
    }
    rgb32[size_t(H - 1 - y) * W + (W - 1 - x)] = 0xff000000u | (q[0] << 16) | (q[1] << 8) | q[2];
}

// ================================================================================================
// Host side: context + C ABI
// ================================================================================================
// Scratch of rt_render_noalloc (include/ray_amd.h): the 256-byte counter block and the HBM part of the path
// stack (levels kLdsLevels .. kNoAllocMaxDepth - 1 of every thread of at most kNoAllocMaxGrid blocks), in a
// file-scope device array instead of hipMalloc'ed buffers. Every GPU the library runs on holds its own copy
// (40 MiB + 256 B) until the process ends. Host code never uses this NAME as an address: rt_init asks the
// runtime for the array's device address on the bound device (hipGetSymbolAddress) and keeps it per device.
constexpr int kNoAllocMaxDepth = RT_NOALLOC_MAX_DEPTH;
constexpr int kNoAllocMaxGrid = 1024;
constexpr size_t kScratchCounterBytes = 256;
constexpr size_t kScratchStackBytes = size_t(kNoAllocMaxDepth - kLdsLevels) * size_t(kNoAllocMaxGrid) * kBlock * sizeof(float4);
constexpr size_t kScratchDoneTiles = size_t(1) << 20;   // tile_state words of render_kernel: frames of up to 2^20 tiles of 8x8 pixels (67 Mpixel); larger ones: whole pixels
constexpr size_t kScratchBytes = kScratchCounterBytes + kScratchStackBytes + kScratchDoneTiles * sizeof(unsigned int);
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {

thread_local std::string g_err;
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};   // device address of `scratch` on each device rt_init has bound

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return 1;
}
#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct Context {
    bool ready = false;
    int device = -1;
    hipDeviceProp_t prop{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.0f;
    int last_launches = 0;
    std::vector<void*> scene_bufs;
    DevScene scene{};
    bool have_scene = false;
    DevCamera cam{};
    bool have_cam = false;
    void* stack = nullptr;
    size_t stack_bytes = 0;
    void* counters = nullptr;   // [0] unsigned int: tile counter of render_kernel; [8] unsigned long long: verify mismatches
    int render_blocks_per_cu = 0;
    void* queue = nullptr;              // render_kernel's tile_state words (grow-only, 4 bytes per 8x8 tile)
    size_t queue_bytes = 0;
    int slice_min = 8, slice_max = kMaxSlices;   // rt_set_sample_slices
    void* tail = nullptr;               // colours of the frame's tail tiles (render_kernel, tail_sum_kernel); grown on demand up to tail_limit_mb,
    size_t tail_bytes = 0;              // released by rt_set_tail_limit / rt_upload_scene / rt_shutdown
    long long tail_limit_mb = 1400;     // rt_set_tail_limit
    bool slices_always = false;         // RT_SLICES_ALWAYS=1 (tests, tools): sample slices also where the tail has its full size
    int sort_mode = 0;                  // rt_set_shading_sort: 1 = block-sorted shading wherever its kernel exists (measured slower: off by default)
    void* pix_sort = nullptr;           // rt_render_pixels: tile keys in / out, sorted pixel indices, radix-sort workspace (grow-only)
    size_t pix_sort_bytes = 0;
    int order_mode = 1;                 // rt_set_tile_order: 1 = frames of many samples and thin filler hand their tiles out in order of measured cost (2: every frame of 64 samples and more)
    void* order_buf = nullptr;          // rays per tile, tile indices, both sorted, the order the kernel reads, radix-sort workspace (grow-only)
    size_t order_bytes = 0;
    int phase_launches = 0;             // launches of the measuring pass of such a frame (its time is part of the frame's)
    hipStream_t own_stream = nullptr;   // created by rt_init; `stream` is this one unless rt_set_stream handed in the caller's
    bool pending = false;               // a RT_RENDER_NOWAIT launch whose end event has not been waited for yet
    int pending_launches = 0;
};
// One context per HIP device, created by rt_init(device). Every host thread has a CURRENT context (the device of
// its last rt_init / rt_set_device); all other entry points act on it.
Context g_ctx[kMaxDevices];
Context g_none;                              // never ready: what a thread sees before its first rt_init
thread_local Context* g_cur = &g_none;
inline Context& G() { return *g_cur; }

int need_ready() {
    if (!G().ready) return fail("rt_init() has not bound a HIP device (no CPU fallback exists)");
    hipError_t e = hipSetDevice(G().device);     // the thread may have used another device since
    if (e != hipSuccess) return fail("hipSetDevice(%d) failed: %s", G().device, hipGetErrorString(e));
    return 0;
}

// RT_RENDER_NOWAIT: at most one launch per context is left in flight; rt_sync() or the next call that launches
// on the context waits for it and takes its time
int settle() {
    if (!G().pending) return 0;
    G().pending = false;
    HIP_OK(hipStreamSynchronize(G().stream));
    HIP_OK(hipEventElapsedTime(&G().last_ms, G().ev0, G().ev1));
    G().last_launches = G().pending_launches;
    return 0;
}
int begin_timed() {
    if (settle()) return 1;
    HIP_OK(hipEventRecord(G().ev0, G().stream));
    return 0;
}
int finish_timed(int launches, bool nowait = false) {
    HIP_OK(hipEventRecord(G().ev1, G().stream));
    HIP_OK(hipGetLastError());
    if (nowait) { G().pending = true; G().pending_launches = launches; return 0; }
    HIP_OK(hipStreamSynchronize(G().stream));
    HIP_OK(hipEventElapsedTime(&G().last_ms, G().ev0, G().ev1));
    G().last_launches = launches;
    return 0;
}

template <class T, class P>
int upload(const std::vector<T>& v, P* out) {
    *out = nullptr;
    size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(T);
    void* d = nullptr;
    HIP_OK(hipMalloc(&d, bytes));
    G().scene_bufs.push_back(d);
    if (!v.empty()) HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(d);
    return 0;
}

void free_scene() {
    for (void* p : G().scene_bufs) (void)hipFree(p);
    G().scene_bufs.clear();
    G().have_scene = false;
}

// Partition of one kind's items (bounding spheres: centre xyz + radius w; r2k: the radius^2 the filter
// uses) into the exact group and the filter group of rt_scan.hpp, and the slot order of the arrays:
// exact group in pairs, padded to whole 32-pair words; then the filter group in spatially compact
// batches of 8 pairs with their bounding balls, padded to whole words. slot_item[slot] = item index, or -1 for a pad.
struct GroupBuild {
    FilterGroup fg{};
    std::vector<int> slot_item;
    std::vector<float> fx, fy, fz, fk, fi, bnd, wbnd, cbnd;
};
// Order `order` (indices into a point set) into spatially compact runs of 16: median splits along the widest
// axis, every part but the last a whole number of runs.
template <class Coord>
void kd_order16(std::vector<int>& order, Coord coord) {
    std::vector<std::pair<int, int>> todo;
    todo.push_back({0, int(order.size())});
    while (!todo.empty()) {
        const int lo = todo.back().first, hi = todo.back().second;
        todo.pop_back();
        if (hi - lo <= 16) continue;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int q = lo; q < hi; ++q) for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], coord(order[q], k)); mx[k] = std::max(mx[k], coord(order[q], k)); }
        int ax = 0;
        for (int k = 1; k < 3; ++k) if (mx[k] - mn[k] > mx[ax] - mn[ax]) ax = k;
        const int nb = (hi - lo + 15) / 16, mid = lo + 16 * (nb / 2);
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](int a, int b) {
            const float ca = coord(a, ax), cb = coord(b, ax);
            return ca < cb || (ca == cb && a < b);
        });
        todo.push_back({lo, mid}); todo.push_back({mid, hi});
    }
}

// The device's slab test (box_pair, rt_scan.hpp) takes boxes as centre C and half extent E. Converts an array of lo / hi boxes
// in place: groups of `group` floats hold `half` lo values followed by `half` hi values; [C - E, C + E] encloses [lo, hi]
// (C the rounded midpoint, E rounded up); an empty box (lo > hi) gets E = -inf, which no ray needs.
void boxes_to_centre_half(std::vector<float>& v, size_t group, size_t half) {
    for (size_t g = 0; g + 2 * half <= v.size(); g += group)
        for (size_t k = 0; k < half; ++k) {
            const float lo = v[g + k], hi = v[g + half + k];
            float c = 0.0f, e = -INFINITY;
            if (lo <= hi) {
                c = float(0.5 * (double(lo) + double(hi)));
                const double ed = std::max(double(hi) - double(c), double(c) - double(lo));
                e = float(ed);
                if (double(e) < ed) e = std::nextafterf(e, INFINITY);
            }
            v[g + k] = c; v[g + half + k] = e;
        }
}

void build_group(const std::vector<float4>& bound, const std::vector<float>& r2k, float margin_k, bool finite, int chunk_words, GroupBuild& out) {
    const int n = int(bound.size());
    std::vector<char> in_filter(n, 0);
    double gc[3] = {0, 0, 0};
    if (finite && n >= 16) {
        std::vector<double> tmp(n);
        for (int k = 0; k < 3; ++k) {
            for (int i = 0; i < n; ++i) tmp[i] = k == 0 ? bound[i].x : (k == 1 ? bound[i].y : bound[i].z);
            std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
            gc[k] = double(float(tmp[n / 2]));
        }
        std::vector<double> ext(n);
        for (int i = 0; i < n; ++i) {
            double dx = bound[i].x - gc[0], dy = bound[i].y - gc[1], dz = bound[i].z - gc[2];
            ext[i] = std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs(double(bound[i].w));
        }
        tmp = ext;
        std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
        const double thr = std::min(8.0 * tmp[n / 2], 0.5e6);
        int n_filter = 0;
        for (int i = 0; i < n; ++i) if (ext[i] <= thr) { in_filter[i] = 1; ++n_filter; }
        if (n_filter < 16 || std::fabs(gc[0]) > 0.5e6 || std::fabs(gc[1]) > 0.5e6 || std::fabs(gc[2]) > 0.5e6)
            std::fill(in_filter.begin(), in_filter.end(), 0);
    }
    FilterGroup& fg = out.fg;
    for (int i = 0; i < n; ++i) if (!in_filter[i]) out.slot_item.push_back(i);
    fg.n_epairs = int((out.slot_item.size() + 1) / 2);
    fg.n_ewords = (fg.n_epairs + 31) / 32;
    out.slot_item.resize(size_t(fg.n_ewords) * 64, -1);
    // filter group in spatially compact batches of 16 items: median splits along the widest axis, every
    // part but the last a whole number of batches
    std::vector<int> order;
    for (int i = 0; i < n; ++i) if (in_filter[i]) order.push_back(i);
    auto coord = [&](int i, int k) { return k == 0 ? bound[i].x : (k == 1 ? bound[i].y : bound[i].z); };
    kd_order16(order, coord);
    double cmax = 0, r2max = 0;
    std::vector<double> rel;   // centre - g as the device sees it (FP32 values)
    for (int i : order) {
        out.slot_item.push_back(i);
        const float rx = float(bound[i].x - gc[0]), ry = float(bound[i].y - gc[1]), rz = float(bound[i].z - gc[2]);
        const double c2 = double(rx) * rx + double(ry) * ry + double(rz) * rz;
        out.fx.push_back(rx); out.fy.push_back(ry); out.fz.push_back(rz); out.fk.push_back(float(c2 - double(r2k[i])));
        rel.push_back(rx); rel.push_back(ry); rel.push_back(rz); rel.push_back(std::sqrt(double(r2k[i])));
        cmax = std::max(cmax, std::sqrt(c2)); r2max = std::max(r2max, double(r2k[i]));
    }
    while (out.fx.size() % 64 != 0) {   // pads to whole words: k huge, never flagged
        out.fx.push_back(0.0f); out.fy.push_back(0.0f); out.fz.push_back(0.0f); out.fk.push_back(3.0e38f);
        out.slot_item.push_back(-1);
    }
    fg.n_fpairs = int(out.fx.size() / 2);
    out.fi.resize(out.fx.size() * 4);   // the device's copy: interleaved per pair
    for (size_t pr = 0; pr < out.fx.size() / 2; ++pr)
        for (int e = 0; e < 2; ++e) {
            out.fi[8 * pr + e] = out.fx[2 * pr + e]; out.fi[8 * pr + 2 + e] = out.fy[2 * pr + e];
            out.fi[8 * pr + 4 + e] = out.fz[2 * pr + e]; out.fi[8 * pr + 6 + e] = out.fk[2 * pr + e];
        }
    // box of every batch around its items' spheres (radius: the one the filter uses), relative to g, rounded outwards
    const int n_fwords = fg.n_fpairs / 32, n_items = int(order.size());
    fg.n_fbatches = (n_items + 15) / 16;
    out.bnd.assign(size_t(n_fwords) * 24, 0.0f);
    auto down = [](double v) { float f = float(v); return double(f) > v ? std::nextafterf(f, -INFINITY) : f; };
    auto up = [](double v) { float f = float(v); return double(f) < v ? std::nextafterf(f, INFINITY) : f; };
    for (int b = 0; b < fg.n_fbatches; ++b) {
        float* w = &out.bnd[size_t(b / 4) * 24];
        const int h = b % 4, i0 = b * 16, i1 = std::min(n_items, i0 + 16);
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        for (int i = i0; i < i1; ++i) for (int k = 0; k < 3; ++k) {
            const double rr = rel[4 * i + 3] * 1.000001;
            mn[k] = std::min(mn[k], rel[4 * i + k] - rr); mx[k] = std::max(mx[k], rel[4 * i + k] + rr);
        }
        for (int k = 0; k < 3; ++k) { w[4 * k + h] = down(mn[k]); w[12 + 4 * k + h] = up(mx[k]); }
    }
    fg.chunk_words = chunk_words;
    const int n_chunks = (n_fwords + chunk_words - 1) / chunk_words;
    out.cbnd.assign(size_t((n_chunks + 1) / 2) * 12, 0.0f);
    for (int ci = 0; ci < n_chunks; ++ci) {
        float* q = &out.cbnd[size_t(ci / 2) * 12];
        for (int k = 0; k < 3; ++k) {
            float lo = INFINITY, hi = -INFINITY;
            for (int b = ci * chunk_words * 4; b < std::min(fg.n_fbatches, (ci + 1) * chunk_words * 4); ++b) {
                const float* w = &out.bnd[size_t(b / 4) * 24];
                lo = std::min(lo, w[4 * k + b % 4]); hi = std::max(hi, w[12 + 4 * k + b % 4]);
            }
            q[2 * k + ci % 2] = lo; q[6 + 2 * k + ci % 2] = hi;
        }
    }
    out.wbnd.assign(size_t((n_fwords + 1) / 2) * 12, 0.0f);
    for (int fw = 0; fw < n_fwords; ++fw) {
        const float* w = &out.bnd[size_t(fw) * 24];
        float* q = &out.wbnd[size_t(fw / 2) * 12];
        const int real = std::min(4, fg.n_fbatches - 4 * fw);
        for (int k = 0; k < 3; ++k) {
            float lo = INFINITY, hi = -INFINITY;
            for (int h = 0; h < real; ++h) { lo = std::min(lo, w[4 * k + h]); hi = std::max(hi, w[12 + 4 * k + h]); }
            q[2 * k + fw % 2] = lo; q[6 + 2 * k + fw % 2] = hi;
        }
    }
    boxes_to_centre_half(out.bnd, 24, 12); boxes_to_centre_half(out.wbnd, 12, 6); boxes_to_centre_half(out.cbnd, 12, 6);
    fg.gx = float(gc[0]); fg.gy = float(gc[1]); fg.gz = float(gc[2]);
    fg.cmax = float(cmax * 1.000001) + 1.0e-30f; fg.r2max = float(r2max * 1.000001);
    fg.margin_k = margin_k;
}

// The sample slices of one call (render_kernel): begin[0] = 0 < begin[1] < ... < begin[n] = spp, returns n <= smax.
// The LAST slice has smin samples (a lane's last unit of work is small, so no lane waits long for another's at the end of the
// frame) and every slice is at most twice its successor, as large as that allows (large units while the frame is in full swing:
// a lane asks for work rarely, and a tile's units are handed on rarely). If smax slices of that shape cannot hold spp samples
// the smallest slice grows. Frames with fewer tiles than waves gain nothing from slices beyond the first few (a tile's slices
// run one after the other whatever their size), but lose nothing either.
int make_slices(int spp, long long n_tiles, long long waves, int smin, int smax, int* begin) {
    (void)n_tiles; (void)waves;
    const double r = 0.5;
    int sizes[kMaxSlices];   // from the last slice backwards
    int n = 0;
    for (double unit = double(smin);; unit *= 1.25) {
        long long sum = 0;
        double cur = unit;
        n = 0;
        while (n < smax && sum < spp) {
            sizes[n] = int(cur) < smin ? smin : int(cur);
            sum += sizes[n++];
            cur /= r;
            if (cur > 2.0e9) cur = 2.0e9;
        }
        if (sum < spp) continue;
        int excess = int(sum - spp);                           // the first slice (sizes[n - 1]) takes what is left
        while (n > 1 && excess >= sizes[n - 1]) excess -= sizes[--n];
        sizes[n - 1] -= excess;
        if (n > 1 && (sizes[n - 1] < smin || 2 * sizes[n - 1] <= sizes[n - 2])) { sizes[n - 2] += sizes[n - 1]; --n; }   // a small first slice joins the second
        break;
    }
    begin[0] = 0;
    for (int k = 0; k < n; ++k) begin[k + 1] = begin[k] + sizes[n - 1 - k];
    return n;
}

int grid_for(int n_tiles) {
    int resident = G().prop.multiProcessorCount * 8;   // 8 blocks of 256 threads per CU = 32 waves/CU
    return n_tiles < resident ? (n_tiles > 0 ? n_tiles : 1) : resident;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

// Everything rt_init creates, on a context that is not yet visible to anyone: a failure part-way leaves nothing behind
// (rt_init releases what was created) and a retry starts from scratch.
static int init_context(Context& c, int device) {
    HIP_OK(hipGetDeviceProperties(&c.prop, device));
    HIP_OK(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.stream = c.own_stream;
    HIP_OK(hipEventCreate(&c.ev0));
    HIP_OK(hipEventCreate(&c.ev1));
    HIP_OK(hipMalloc(&c.counters, 128));
    HIP_OK(hipMemset(c.counters, 0, 128));
    if (!g_scratch_ptr[device]) {   // once per device, with that device current
        void* sp = nullptr;
        HIP_OK(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
        if (!sp) return fail("hipGetSymbolAddress(scratch) returned a null address on device %d", device);
        g_scratch_ptr[device] = sp;
    }
    int per_cu = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_kernel<false, false, 0>, kBlock, 0));
    c.render_blocks_per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    if (const char* e = getenv("RT_SLICES_ALWAYS")) c.slices_always = atoi(e) != 0;
    if (const char* e = getenv("RT_TILE_ORDER")) { const int v = atoi(e); if (v >= 0 && v <= 2) c.order_mode = v; }   // tools/: rt_set_tile_order
    if (const char* e = getenv("RT_TAIL_MB")) { const long long v = atoll(e); if (v >= 0) c.tail_limit_mb = v; }
    if (const char* e = getenv("RT_SLICE_MIN")) { const int v = atoi(e); if (v >= 0) c.slice_min = v; }   // tools/: 0 = whole pixels
    c.device = device;
    return 0;
}
// releases what a context holds (the device must be current); the context can be initialised again afterwards
static void destroy_context(Context& c) {
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    for (void* p : c.scene_bufs) (void)hipFree(p);
    c.scene_bufs.clear();
    if (c.stack) (void)hipFree(c.stack);
    if (c.counters) (void)hipFree(c.counters);
    if (c.pix_sort) (void)hipFree(c.pix_sort);
    if (c.order_buf) (void)hipFree(c.order_buf);
    c.order_buf = nullptr; c.order_bytes = 0;
    if (c.queue) (void)hipFree(c.queue);
    if (c.tail) (void)hipFree(c.tail);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
    c = Context{};
}

int rt_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail("no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= n || device >= kMaxDevices) return fail("device %d out of range (0..%d)", device, (n < kMaxDevices ? n : kMaxDevices) - 1);
    HIP_OK(hipSetDevice(device));
    if (!g_ctx[device].ready) {
        Context fresh;
        if (init_context(fresh, device)) {   // (the error message is set) nothing of the attempt stays behind
            const std::string msg = g_err;
            destroy_context(fresh);
            g_err = msg;
            return 1;
        }
        fresh.ready = true;
        g_ctx[device] = fresh;
    }
    g_cur = &g_ctx[device];   // only a context that is ready becomes a thread's current one
    return 0;
}

int rt_device_count(int* count) {
    if (!count) return fail("rt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail("hipGetDeviceCount failed: %s", hipGetErrorString(e)); }
    *count = n < kMaxDevices ? n : kMaxDevices;
    return 0;
}

int rt_set_device(int device) {
    if (device < 0 || device >= kMaxDevices || !g_ctx[device].ready) return fail("rt_set_device: device %d has no context (rt_init it first)", device);
    g_cur = &g_ctx[device];
    HIP_OK(hipSetDevice(device));
    return 0;
}

int rt_set_stream(void* hip_stream) {
    if (need_ready()) return 1;
    if (settle()) return 1;
    HIP_OK(hipStreamSynchronize(G().stream));    // nothing of ours is left behind on the stream we leave
    G().stream = static_cast<hipStream_t>(hip_stream);   // NULL is a stream too: HIP's default stream (torch's default)
    return 0;
}

int rt_reset_stream(void) {
    if (need_ready()) return 1;
    if (settle()) return 1;
    HIP_OK(hipStreamSynchronize(G().stream));
    G().stream = G().own_stream;
    return 0;
}

int rt_set_sample_slices(int min_samples, int max_slices) {
    if (need_ready()) return 1;
    if (min_samples < 0 || max_slices < 1) return fail("rt_set_sample_slices: bad arguments %d, %d", min_samples, max_slices);
    G().slice_min = min_samples;
    G().slice_max = max_slices < kMaxSlices ? max_slices : kMaxSlices;
    return 0;
}

int rt_set_tile_order(int mode) {
    if (need_ready()) return 1;
    if (mode < 0 || mode > 2) return fail("rt_set_tile_order: mode %d outside 0..2", mode);
    G().order_mode = mode;
    return 0;
}

int rt_set_tail_limit(int64_t megabytes) {
    if (need_ready()) return 1;
    if (megabytes < 0 || megabytes > (int64_t(1) << 20)) return fail("rt_set_tail_limit: %lld MiB outside 0..2^20", (long long)megabytes);
    if (settle()) return 1;
    HIP_OK(hipStreamSynchronize(G().stream));
    G().tail_limit_mb = megabytes;
    if (G().tail && G().tail_bytes > size_t(megabytes) << 20) {   // what is held beyond the new limit goes back at once
        HIP_OK(hipFree(G().tail));
        G().tail = nullptr; G().tail_bytes = 0;
    }
    return 0;
}

int rt_set_shading_sort(int mode) {
    if (need_ready()) return 1;
    if (mode < 0 || mode > 1) return fail("rt_set_shading_sort: mode %d outside 0..1", mode);
    G().sort_mode = mode;
    return 0;
}

void rt_shutdown(void) {
    if (!G().ready) return;
    (void)hipSetDevice(G().device);
    destroy_context(G());
}

int rt_device_info(char* buf, int buflen) {
    if (need_ready()) return 1;
    if (!buf || buflen <= 0) return fail("rt_device_info: no buffer");
    snprintf(buf, size_t(buflen), "%s|%s|%d|%d", G().prop.name, G().prop.gcnArchName, G().prop.multiProcessorCount, G().prop.clockRate / 1000);
    return 0;
}

int rt_malloc(void** d_ptr, uint64_t bytes) { if (need_ready()) return 1; HIP_OK(hipMalloc(d_ptr, bytes ? bytes : 1)); return 0; }
int rt_free(void* d_ptr) { if (need_ready()) return 1; HIP_OK(hipFree(d_ptr)); return 0; }
int rt_malloc_host(void** ptr, uint64_t bytes) {
    if (need_ready()) return 1;
    if (!ptr) return fail("rt_malloc_host: null argument");
    HIP_OK(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));   // coherent, mapped into every device's address space
    return 0;
}
int rt_free_host(void* ptr) { if (need_ready()) return 1; HIP_OK(hipHostFree(ptr)); return 0; }
int rt_memcpy_d2h(void* dst, const void* d_src, uint64_t bytes) {
    if (need_ready()) return 1;
    if (settle()) return 1;
    HIP_OK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, G().stream));
    HIP_OK(hipStreamSynchronize(G().stream));
    return 0;
}
int rt_memcpy_h2d(void* d_dst, const void* src, uint64_t bytes) {
    if (need_ready()) return 1;
    if (settle()) return 1;
    HIP_OK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, G().stream));
    HIP_OK(hipStreamSynchronize(G().stream));
    return 0;
}
int rt_memset(void* d_ptr, int value, uint64_t bytes) {
    if (need_ready()) return 1;
    if (settle()) return 1;
    HIP_OK(hipMemsetAsync(d_ptr, value, bytes, G().stream));
    HIP_OK(hipStreamSynchronize(G().stream));
    return 0;
}
int rt_sync(void) { if (need_ready()) return 1; if (settle()) return 1; HIP_OK(hipStreamSynchronize(G().stream)); return 0; }

int rt_upload_scene(const RtScene* s) {
    if (need_ready()) return 1;
    if (!s) return fail("rt_upload_scene: null scene");
    if (s->n_prims < 0 || s->n_mats <= 0) return fail("rt_upload_scene: n_prims=%d n_mats=%d", s->n_prims, s->n_mats);
    if (s->n_prims > 0 && (!s->prim_kind || !s->prim_mat || !s->prim_data)) return fail("rt_upload_scene: null primitive arrays");
    if (!s->mat_kind || !s->mat_rgb || !s->mat_fuzz || !s->mat_ir) return fail("rt_upload_scene: null material arrays");
    std::vector<float4> sph; std::vector<int> sph_prim;
    std::vector<double> rect; std::vector<int> rect_axis, rect_prim;
    std::vector<float> tri; std::vector<int> tri_prim;
    std::vector<float> med_bound; std::vector<double> med_niv; std::vector<int> med_is_box;
    std::vector<float> tbvh_nodes; float tri_extent = 0.0f; int tbvh_leaf_items = 1;
    std::vector<float> sbvh_nodes;
    std::vector<int> prim_kind(s->n_prims), prim_local(s->n_prims), prim_mat(s->n_prims);
    for (int i = 0; i < s->n_prims; ++i) {
        const double* d = s->prim_data + size_t(i) * RT_PRIM_STRIDE;
        int m = s->prim_mat[i];
        if (m < 0 || m >= s->n_mats) return fail("rt_upload_scene: primitive %d has material %d (n_mats=%d)", i, m, s->n_mats);
        prim_mat[i] = m;
        switch (s->prim_kind[i]) {
        case RT_PRIM_SPHERE:
            prim_kind[i] = 0; prim_local[i] = int(sph.size());
            sph.push_back(make_float4(float(d[0]), float(d[1]), float(d[2]), float(d[3]))); sph_prim.push_back(i);
            break;
        case RT_PRIM_XY_RECT: case RT_PRIM_XZ_RECT: case RT_PRIM_YZ_RECT:
            prim_kind[i] = 1; prim_local[i] = int(rect_axis.size());
            rect.insert(rect.end(), d, d + 5); rect_axis.push_back(s->prim_kind[i] - RT_PRIM_XY_RECT); rect_prim.push_back(i);
            break;
        case RT_PRIM_TRIANGLE: {
            prim_kind[i] = 2; prim_local[i] = int(tri_prim.size());
            V3 v0 = v3(float(d[0]), float(d[1]), float(d[2])), v1 = v3(float(d[3]), float(d[4]), float(d[5])),
               v2 = v3(float(d[6]), float(d[7]), float(d[8]));
            V3 e1 = v1 - v0, e2 = v2 - v0, nn = normalized(cross(e1, e2));
            const float fa[3] = {v0.x, v0.y, v0.z}, fb[3] = {v1.x, v1.y, v1.z}, fc[3] = {v2.x, v2.y, v2.z};
            float cb[3], rb2;
            rt_triangle_bound(fa, fb, fc, cb, &rb2);
            const float q[16] = {v0.x, v0.y, v0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, nn.x, nn.y, nn.z, cb[0], cb[1], cb[2], rb2};
            tri.insert(tri.end(), q, q + 16); tri_prim.push_back(i);
            break; }
        case RT_PRIM_MEDIUM_SPHERE: case RT_PRIM_MEDIUM_BOX: {
            if (s->mat_kind[m] != RT_MAT_ISOTROPIC)
                return fail("rt_upload_scene: primitive %d is a constant_medium but its material %d is not RT_MAT_ISOTROPIC (shapes.h:376-388)", i, m);
            const bool is_box = s->prim_kind[i] == RT_PRIM_MEDIUM_BOX;
            prim_kind[i] = 3; prim_local[i] = int(med_niv.size());
            for (int k = 0; k < 6; ++k) med_bound.push_back(k < (is_box ? 6 : 4) ? float(d[k]) : 0.0f);
            med_niv.push_back(d[is_box ? 6 : 4]); med_is_box.push_back(is_box ? 1 : 0);
            break; }
        default: return fail("rt_upload_scene: primitive %d has unknown kind %d", i, s->prim_kind[i]);
        }
    }
    for (int i = 0; i < s->n_mats; ++i)
        if (s->mat_kind[i] < RT_MAT_LAMBERTIAN || s->mat_kind[i] > RT_MAT_ISOTROPIC)
            return fail("rt_upload_scene: material %d has unknown kind %d", i, s->mat_kind[i]);
    if (s->light_kind != RT_LIGHT_BASE && s->light_kind != RT_PRIM_SPHERE && s->light_kind != RT_PRIM_XY_RECT &&
        s->light_kind != RT_PRIM_XZ_RECT)
        return fail("rt_upload_scene: light kind %d has no pdf_value/random in the reference; pass RT_LIGHT_BASE", s->light_kind);

    // ---- filter groups (rt_scan.hpp) of the spheres and of the triangles' bounding spheres ------------------
    bool finite = true;
    GroupBuild sgb, tgb;
    std::vector<float4> fq;
    {
        std::vector<float4> bound(sph.size());
        std::vector<float> r2k(sph.size());
        for (size_t i = 0; i < sph.size(); ++i) {
            bound[i] = sph[i]; r2k[i] = sph[i].w * sph[i].w;   // radiusSquared, shapes.h:73
            if (!(std::fabs(sph[i].x) <= 1.0e7f && std::fabs(sph[i].y) <= 1.0e7f && std::fabs(sph[i].z) <= 1.0e7f && std::fabs(sph[i].w) <= 1.0e7f)) finite = false;
        }
        build_group(bound, r2k, kMarginK, finite, kMaskWords, sgb);
        const float nanf_ = std::numeric_limits<float>::quiet_NaN();
        std::vector<float4> sph2(sgb.slot_item.size(), make_float4(nanf_, nanf_, nanf_, 1.0f));
        std::vector<int> prim2(sgb.slot_item.size(), -1);
        for (size_t sl = 0; sl < sgb.slot_item.size(); ++sl) {
            const int it = sgb.slot_item[sl];
            if (it >= 0) { sph2[sl] = sph[it]; prim2[sl] = sph_prim[it]; prim_local[sph_prim[it]] = int(sl); }
        }
        for (size_t pr = 0; pr < sph2.size() / 2; ++pr) {
            const float4 s0 = sph2[2 * pr], s1 = sph2[2 * pr + 1];
            fq.push_back(make_float4(s0.x, s1.x, s0.y, s1.y));
            fq.push_back(make_float4(s0.z, s1.z, s0.w * s0.w, s1.w * s1.w));
        }
        sph.swap(sph2); sph_prim.swap(prim2);
#ifdef RT_EXPERIMENT_SPHERE_BVH
        {   // BVH over the sphere pairs (experiment): balanced median splits, leaf = pair, box around both spheres, relative to g
            std::vector<int> leaves;
            for (size_t pr = size_t(sgb.fg.n_ewords) * 32; pr < sph.size() / 2; ++pr)   // the filter group's pairs; the exact group is visited first as always
                if (sph_prim[2 * pr] >= 0 || sph_prim[2 * pr + 1] >= 0) leaves.push_back(int(pr));
            const double g3[3] = {double(sgb.fg.gx), double(sgb.fg.gy), double(sgb.fg.gz)};
            struct Box { float lo[3], hi[3]; };
            auto down = [](double v) { float f = float(v); return double(f) > v ? std::nextafterf(f, -INFINITY) : f; };
            auto up = [](double v) { float f = float(v); return double(f) < v ? std::nextafterf(f, INFINITY) : f; };
            auto leaf_box = [&](int pr) {
                Box bx;
                for (int k = 0; k < 3; ++k) { bx.lo[k] = INFINITY; bx.hi[k] = -INFINITY; }
                for (int hh = 0; hh < 2; ++hh) {
                    if (sph_prim[2 * pr + hh] < 0) continue;
                    const float4 b = sph[2 * pr + hh];
                    const double c[3] = {b.x, b.y, b.z}, rr = std::fabs(double(b.w)) * 1.000001;
                    for (int k = 0; k < 3; ++k) { bx.lo[k] = std::min(bx.lo[k], down(c[k] - g3[k] - rr)); bx.hi[k] = std::max(bx.hi[k], up(c[k] - g3[k] + rr)); }
                }
                return bx;
            };
            auto centre = [&](int pr, int k) { const float4 b = sph[2 * pr + (sph_prim[2 * pr] >= 0 ? 0 : 1)]; return k == 0 ? b.x : (k == 1 ? b.y : b.z); };
            std::vector<float> nodes;
            float extent = 0.0f;
            std::function<int(int, int, Box&)> build = [&](int lo, int hi, Box& out) -> int {
                if (hi - lo == 1) { out = leaf_box(leaves[lo]); return ~leaves[lo]; }
                float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int q = lo; q < hi; ++q) for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], centre(leaves[q], k)); mx[k] = std::max(mx[k], centre(leaves[q], k)); }
                int ax = 0;
                for (int k = 1; k < 3; ++k) if (mx[k] - mn[k] > mx[ax] - mn[ax]) ax = k;
                const int mid = lo + (hi - lo) / 2;
                std::nth_element(leaves.begin() + lo, leaves.begin() + mid, leaves.begin() + hi, [&](int a, int b) {
                    const float ca = centre(a, ax), cb = centre(b, ax);
                    return ca < cb || (ca == cb && a < b);
                });
                const int me = int(nodes.size() / 16);
                nodes.resize(nodes.size() + 16, 0.0f);
                Box b0, b1;
                const int c0 = build(lo, mid, b0), c1 = build(mid, hi, b1);
                float* n = &nodes[size_t(me) * 16];
                n[0] = b0.lo[0]; n[1] = b1.lo[0]; n[2] = b0.lo[1]; n[3] = b1.lo[1]; n[4] = b0.lo[2]; n[5] = b1.lo[2];
                n[6] = b0.hi[0]; n[7] = b1.hi[0]; n[8] = b0.hi[1]; n[9] = b1.hi[1]; n[10] = b0.hi[2]; n[11] = b1.hi[2];
                memcpy(&n[12], &c0, 4); memcpy(&n[13], &c1, 4);
                for (int k = 0; k < 3; ++k) {
                    out.lo[k] = std::min(b0.lo[k], b1.lo[k]); out.hi[k] = std::max(b0.hi[k], b1.hi[k]);
                    extent = std::max(extent, std::max(std::fabs(out.lo[k]), std::fabs(out.hi[k])));
                }
                return me;
            };
            sgb.fg.sbvh = nullptr; sgb.fg.sbvh_extent = 0.0f;
            if (finite && leaves.size() >= 8 && leaves.size() <= (size_t(1) << (kMaskWords - 1))) {
                Box root;
                build(0, int(leaves.size()), root);
                boxes_to_centre_half(nodes, 16, 6);
                sbvh_nodes.swap(nodes); sgb.fg.sbvh_extent = extent;
            }
        }
#endif
    }
    {
        // Filter items of the triangle group: ONE bounding sphere per item of up to two triangles. Two
        // consecutive triangles share an item when the sphere around both bounding spheres is at most 15 %
        // wider than the larger of the two (the two halves of a quad have the same minimal sphere: the cell's).
        const size_t nt = tri_prim.size();
        std::vector<int> item_a, item_b;   // triangle indices of each item (item_b = -1: single)
        std::vector<float4> bound;
        std::vector<float> r2k;
        auto joint = [&](size_t i, size_t j, double c[3]) {   // sphere around the bounding spheres of triangles i and j
            const float *p = &tri[16 * i], *q = &tri[16 * j];
            const double ri = std::sqrt(double(p[15])), rj = std::sqrt(double(q[15]));
            for (int k = 0; k < 3; ++k) { c[k] = double(float(0.5 * (double(p[12 + k]) + double(q[12 + k])))); }
            double ra = 0, rb_ = 0;
            for (int k = 0; k < 3; ++k) { ra += (p[12 + k] - c[k]) * (p[12 + k] - c[k]); rb_ += (q[12 + k] - c[k]) * (q[12 + k] - c[k]); }
            return std::max(std::sqrt(ra) + ri, std::sqrt(rb_) + rj);
        };
        for (size_t i = 0; i < nt; ++i) {
            const float* q = &tri[16 * i];
            for (int k = 0; k < 9; ++k) if (!(std::fabs(q[k]) <= 1.0e7f)) finite = false;
            if (!(std::sqrt(q[15]) <= 1.0e7f)) finite = false;
        }
        for (size_t i = 0; i < nt;) {
            const float* q = &tri[16 * i];
            const double ri = std::sqrt(double(q[15]));
            double c[3];
            if (finite && i + 1 < nt) {
                const double rj = std::sqrt(double(tri[16 * (i + 1) + 15]));
                const double r = joint(i, i + 1, c);
                if (r <= 1.15 * std::max(ri, rj)) {
                    const float rf = float(r * 1.000001) + 1.0e-30f;
                    item_a.push_back(int(i)); item_b.push_back(int(i + 1));
                    bound.push_back(make_float4(float(c[0]), float(c[1]), float(c[2]), rf));
                    r2k.push_back(rf * rf * 1.0001f);   // the FP32 containment test of tri_t passes only inside this sphere
                    i += 2;
                    continue;
                }
            }
            item_a.push_back(int(i)); item_b.push_back(-1);
            bound.push_back(make_float4(q[12], q[13], q[14], float(ri)));
            r2k.push_back(q[15] * 1.0001f);
            ++i;
        }
        build_group(bound, r2k, kMarginKTri, finite, kItemChunkWords, tgb);
        if (tgb.slot_item.size() >= (size_t(1) << 26))   // scan_triangles packs an item slot and a lane into 32 bits
            return fail("rt_upload_scene: %zu triangle item slots exceed the limit of 2^26", tgb.slot_item.size());

        const float nanf_ = std::numeric_limits<float>::quiet_NaN();
        std::vector<float> tri2(tgb.slot_item.size() * 2 * 16, nanf_);   // two triangle slots per item slot
        std::vector<int> prim2(tgb.slot_item.size() * 2, -1);
        for (size_t sl = 0; sl < tgb.slot_item.size(); ++sl) {
            const int it = tgb.slot_item[sl];
            if (it < 0) continue;
            const int members[2] = {item_a[it], item_b[it]};
            for (int h = 0; h < 2; ++h) if (members[h] >= 0) {
                const size_t ts = 2 * sl + h;
                std::copy(&tri[16 * members[h]], &tri[16 * members[h]] + 16, &tri2[16 * ts]);
                prim2[ts] = tri_prim[members[h]]; prim_local[tri_prim[members[h]]] = int(ts);
            }
        }
        tri.swap(tri2); tri_prim.swap(prim2);
        // BVH over the items (scan_triangles_bvh): balanced median splits along the widest axis of the items' centres, a
        // leaf per item, the box of a leaf around the item's bounding sphere (the radius the filter uses, a hair wider
        // than the one tri_t tests against), relative to the group's reference point, rounded outwards.
        // leaf = an aligned run of L consecutive item slots (consecutive slots are spatially compact: exact group in list
        // order, filter group in kd order); L = 1 unless the tree would need more than kMaskWords - 1 levels
        size_t n_real = 0;
        for (int it : tgb.slot_item) n_real += it >= 0 ? 1 : 0;
        int L = 1;
        while ((tgb.slot_item.size() + L - 1) / L > (size_t(1) << (kMaskWords - 1)) && L < 16) L *= 2;
        std::vector<int> leaves;
        for (size_t sl = 0; sl < tgb.slot_item.size(); sl += L) {
            bool any = false;
            for (size_t k = sl; k < std::min(tgb.slot_item.size(), sl + L); ++k) any = any || tgb.slot_item[k] >= 0;
            if (any) leaves.push_back(int(sl));
        }
        // OFF by default: measured against the flat chunk / word / batch levels the per-lane walk is 1.1x to 2.4x slower from
        // 2.4k to 400k triangles (tools/bvh_crossover.py). RT_TRI_BVH_MIN=<items> switches it on for scenes
        // of at least that many items (the tests force it onto tiny scenes with 1).
        int bvh_min = 0x7fffffff;
        if (const char* e = getenv("RT_TRI_BVH_MIN")) bvh_min = atoi(e);
        if (finite && int(n_real) >= bvh_min && !leaves.empty() && leaves.size() <= (size_t(1) << (kMaskWords - 1))) {
            const double g3[3] = {double(tgb.fg.gx), double(tgb.fg.gy), double(tgb.fg.gz)};
            auto down = [](double v) { float f = float(v); return double(f) > v ? std::nextafterf(f, -INFINITY) : f; };
            auto up = [](double v) { float f = float(v); return double(f) < v ? std::nextafterf(f, INFINITY) : f; };
            struct Box { float lo[3], hi[3]; };
            auto leaf_box = [&](int sl0) {
                Box bx;
                for (int k = 0; k < 3; ++k) { bx.lo[k] = INFINITY; bx.hi[k] = -INFINITY; }
                for (size_t sl = size_t(sl0); sl < std::min(tgb.slot_item.size(), size_t(sl0) + L); ++sl) {
                    if (tgb.slot_item[sl] < 0) continue;
                    const float4 b = bound[tgb.slot_item[sl]];
                    const double rr = std::sqrt(double(r2k[tgb.slot_item[sl]])) * 1.000001;
                    const double c[3] = {b.x, b.y, b.z};
                    for (int k = 0; k < 3; ++k) { bx.lo[k] = std::min(bx.lo[k], down(c[k] - g3[k] - rr)); bx.hi[k] = std::max(bx.hi[k], up(c[k] - g3[k] + rr)); }
                }
                return bx;
            };
            auto leaf_centre = [&](int sl0, int k) {     // centre of the run's first real item: a sort key, nothing more
                for (size_t sl = size_t(sl0); sl < std::min(tgb.slot_item.size(), size_t(sl0) + L); ++sl) {
                    if (tgb.slot_item[sl] < 0) continue;
                    const float4 b = bound[tgb.slot_item[sl]];
                    return k == 0 ? b.x : (k == 1 ? b.y : b.z);
                }
                return 0.0f;
            };
            std::vector<float> nodes;   // 16 floats per node
            float extent = 0.0f;
            // returns the child reference of the subtree over leaves[lo, hi) and its box
            std::function<int(int, int, Box&)> build = [&](int lo, int hi, Box& out) -> int {
                if (hi - lo == 1) { out = leaf_box(leaves[lo]); return ~leaves[lo]; }
                float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
                auto centre = [&](int sl, int k) { return leaf_centre(sl, k); };
                for (int q = lo; q < hi; ++q) for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], centre(leaves[q], k)); mx[k] = std::max(mx[k], centre(leaves[q], k)); }
                int ax = 0;
                for (int k = 1; k < 3; ++k) if (mx[k] - mn[k] > mx[ax] - mn[ax]) ax = k;
                const int mid = lo + (hi - lo) / 2;
                std::nth_element(leaves.begin() + lo, leaves.begin() + mid, leaves.begin() + hi, [&](int a, int b) {
                    const float ca = centre(a, ax), cb = centre(b, ax);
                    return ca < cb || (ca == cb && a < b);
                });
                const int me = int(nodes.size() / 16);
                nodes.resize(nodes.size() + 16, 0.0f);
                Box b0, b1;
                const int c0 = build(lo, mid, b0), c1 = build(mid, hi, b1);
                float* n = &nodes[size_t(me) * 16];
                n[0] = b0.lo[0]; n[1] = b1.lo[0]; n[2] = b0.lo[1]; n[3] = b1.lo[1]; n[4] = b0.lo[2]; n[5] = b1.lo[2];
                n[6] = b0.hi[0]; n[7] = b1.hi[0]; n[8] = b0.hi[1]; n[9] = b1.hi[1]; n[10] = b0.hi[2]; n[11] = b1.hi[2];
                memcpy(&n[12], &c0, 4); memcpy(&n[13], &c1, 4);
                for (int k = 0; k < 3; ++k) {
                    out.lo[k] = std::min(b0.lo[k], b1.lo[k]); out.hi[k] = std::max(b0.hi[k], b1.hi[k]);
                    extent = std::max(extent, std::max(std::fabs(out.lo[k]), std::fabs(out.hi[k])));
                }
                return me;
            };
            Box root;
            if (leaves.size() == 1) {       // a root needs two children: the one leaf twice (the second visit changes nothing)
                root = leaf_box(leaves[0]);
                const int c = ~leaves[0];
                nodes.assign(16, 0.0f);
                for (int k = 0; k < 3; ++k) { nodes[2 * k] = nodes[2 * k + 1] = root.lo[k]; nodes[6 + 2 * k] = nodes[7 + 2 * k] = root.hi[k];
                                              extent = std::max(extent, std::max(std::fabs(root.lo[k]), std::fabs(root.hi[k]))); }
                memcpy(&nodes[12], &c, 4); memcpy(&nodes[13], &c, 4);
            } else {
                build(0, int(leaves.size()), root);
            }
            boxes_to_centre_half(nodes, 16, 6);
            tbvh_nodes.swap(nodes); tri_extent = extent; tbvh_leaf_items = L;
        }
    }

    int n_rect_axis[3] = {0, 0, 0};
    {   // few rects: sorted by axis (stable), the direct scan runs one loop per axis. Many (>= kRectFlagMin): list order,
        // in batches of 16 for the flag pass, whose walk reads each rect's axis from rect_axis. The list order of a
        // scene keeps the six sides of a `box` together (rayview.cpp:273-375); on box_scene it gives tighter batch
        // boxes than a median split over the rects' centres (17.9 vs 22.5 ms at 16 spp).
        std::vector<int> order;
        if (int(rect_axis.size()) >= kRectFlagMin) {
            for (size_t j = 0; j < rect_axis.size(); ++j) order.push_back(int(j));
        } else {
            for (int ax = 0; ax < 3; ++ax)
                for (size_t j = 0; j < rect_axis.size(); ++j) if (rect_axis[j] == ax) order.push_back(int(j));
        }
        std::vector<double> rect2; std::vector<int> axis2, prim2;
        for (int j : order) {
            prim_local[rect_prim[j]] = int(axis2.size());
            rect2.insert(rect2.end(), &rect[5 * j], &rect[5 * j] + 5); axis2.push_back(rect_axis[j]); prim2.push_back(rect_prim[j]);
            ++n_rect_axis[rect_axis[j]];
        }
        rect.swap(rect2); rect_axis.swap(axis2); rect_prim.swap(prim2);
    }
    std::vector<float> rect_f(rect.size());
    {
        auto down = [](double v) { float f = float(v); return double(f) > v ? std::nextafterf(f, -INFINITY) : f; };
        auto up = [](double v) { float f = float(v); return double(f) < v ? std::nextafterf(f, INFINITY) : f; };
        for (size_t j = 0; j < rect_axis.size(); ++j) {
            const double* q = &rect[5 * j];
            float* f = &rect_f[5 * j];
            f[0] = float(q[4]); f[1] = down(q[0]); f[2] = up(q[1]); f[3] = down(q[2]); f[4] = up(q[3]);
        }
    }

    std::vector<float> rect_box(((rect_axis.size() + 31) / 32) * 16 * 12, 0.0f);   // whole words of 16 pairs; pads: a box at the origin, masked by index
    float rect_extent = 0.0f;
    for (size_t j = 0; j < rect_axis.size(); ++j) {
        const float* f = &rect_f[5 * j];   // k nearest, a0 down, a1 up, b0 down, b1 up
        const double k = rect[5 * j + 4];
        for (int c = 0; c < 5; ++c) if (!(std::fabs(rect[5 * j + c]) <= 1.0e7)) finite = false;   // inf / NaN / huge: list-order scan only
        const float kd = double(f[0]) > k ? std::nextafterf(f[0], -INFINITY) : f[0], ku = double(f[0]) < k ? std::nextafterf(f[0], INFINITY) : f[0];
        float lo[3], hi[3];
        const int ax = rect_axis[j];       // 0: XY (z = k; a = x, b = y), 1: XZ (y = k; a = x, b = z), 2: YZ (x = k; a = y, b = z)
        if (ax == 0) { lo[0] = f[1]; hi[0] = f[2]; lo[1] = f[3]; hi[1] = f[4]; lo[2] = kd; hi[2] = ku; }
        else if (ax == 1) { lo[0] = f[1]; hi[0] = f[2]; lo[2] = f[3]; hi[2] = f[4]; lo[1] = kd; hi[1] = ku; }
        else { lo[1] = f[1]; hi[1] = f[2]; lo[2] = f[3]; hi[2] = f[4]; lo[0] = kd; hi[0] = ku; }
        float* q = &rect_box[(j / 2) * 12];
        for (int c = 0; c < 3; ++c) {
            q[2 * c + j % 2] = lo[c]; q[6 + 2 * c + j % 2] = hi[c];
            rect_extent = std::max(rect_extent, std::max(std::fabs(lo[c]), std::fabs(hi[c])));
        }
    }
    if (!(rect_extent <= 1.0e7f)) finite = false;
    // box around every batch of 16 rects (8 pairs of rect_box), two batches per 12 floats
    const int n_rbatches = int((rect_axis.size() + 15) / 16);
    std::vector<float> rect_bbox(size_t((n_rbatches + 1) / 2) * 12, 0.0f);
    for (int b = 0; b < n_rbatches; ++b) {
        float* q = &rect_bbox[size_t(b / 2) * 12];
        for (int c = 0; c < 3; ++c) {
            float lo = INFINITY, hi = -INFINITY;
            for (size_t j = size_t(b) * 16; j < std::min(rect_axis.size(), size_t(b + 1) * 16); ++j) {
                lo = std::min(lo, rect_box[(j / 2) * 12 + 2 * c + j % 2]); hi = std::max(hi, rect_box[(j / 2) * 12 + 6 + 2 * c + j % 2]);
            }
            q[2 * c + b % 2] = lo; q[6 + 2 * c + b % 2] = hi;
        }
    }

    if (settle()) return 1;
    HIP_OK(hipStreamSynchronize(G().stream));
    free_scene();
    if (G().tail) { (void)hipFree(G().tail); G().tail = nullptr; G().tail_bytes = 0; }   // a new scene: the colour buffer comes back with the next frame that wants it
    DevScene ds{};
    ds.n_rect_xy = n_rect_axis[0]; ds.n_rect_xz = n_rect_axis[1]; ds.rect_extent = rect_extent;
    ds.n_sph = int(sph.size()); ds.n_rect = int(rect_axis.size()); ds.n_tri = int(tri_prim.size()); ds.n_prims = s->n_prims;
    boxes_to_centre_half(rect_box, 12, 6); boxes_to_centre_half(rect_bbox, 12, 6);
    if (upload(sph, &ds.sph) || upload(sph_prim, &ds.sph_prim) || upload(rect, &ds.rect) || upload(rect_f, &ds.rect_f) || upload(rect_box, &ds.rect_box) || upload(rect_bbox, &ds.rect_bbox) || upload(rect_axis, &ds.rect_axis) ||
        upload(rect_prim, &ds.rect_prim) || upload(tri, &ds.tri) || upload(tri_prim, &ds.tri_prim) ||
        upload(prim_kind, &ds.prim_kind) || upload(prim_local, &ds.prim_local) || upload(prim_mat, &ds.prim_mat))
        return 1;
    std::vector<int> mk(s->mat_kind, s->mat_kind + s->n_mats);
    std::vector<float> mrgb(s->mat_rgb, s->mat_rgb + 3 * size_t(s->n_mats)), mf(s->mat_fuzz, s->mat_fuzz + s->n_mats);
    std::vector<double> mir(s->mat_ir, s->mat_ir + s->n_mats), minv(s->n_mats);
    for (int i = 0; i < s->n_mats; ++i) minv[i] = 1.0 / mir[i];
    if (upload(mk, &ds.mat_kind) || upload(mrgb, &ds.mat_rgb) || upload(mf, &ds.mat_fuzz) || upload(mir, &ds.mat_ir) ||
        upload(minv, &ds.mat_inv_ir)) return 1;
    if (upload(sgb.fi, &sgb.fg.fi) ||
        upload(sgb.bnd, &sgb.fg.bnd) || upload(sgb.wbnd, &sgb.fg.wbnd) || upload(sgb.cbnd, &sgb.fg.cbnd) || upload(fq, &sgb.fg.fq)) return 1;
    if (upload(tgb.fi, &tgb.fg.fi) ||
        upload(tgb.bnd, &tgb.fg.bnd) || upload(tgb.wbnd, &tgb.fg.wbnd) || upload(tgb.cbnd, &tgb.fg.cbnd)) return 1;
    tgb.fg.fq = nullptr;
#ifdef RT_EXPERIMENT_SPHERE_BVH
    tgb.fg.sbvh = nullptr; tgb.fg.sbvh_extent = 0.0f;
    if (!sbvh_nodes.empty() && finite) {
        const float* dn = nullptr;
        if (upload(sbvh_nodes, &dn)) return 1;
        sgb.fg.sbvh = reinterpret_cast<const float4*>(dn);
    }
#endif
    if (!tbvh_nodes.empty() && finite) {
        const float* dn = nullptr;
        if (upload(tbvh_nodes, &dn)) return 1;
        ds.tbvh = reinterpret_cast<const float4*>(dn); ds.tri_extent = tri_extent; ds.tbvh_leaf_items = tbvh_leaf_items;
    }
    ds.n_med = int(med_niv.size());
    if (upload(med_bound, &ds.med_bound) || upload(med_niv, &ds.med_niv) || upload(med_is_box, &ds.med_is_box)) return 1;
    // a constant_medium draws a random number INSIDE hitFromList's loop, for every ray that crosses its boundary within
    // the closest hit so far: which rays draw depends on list order, so such scenes are scanned in list order only
    ds.sg = sgb.fg; ds.tg = tgb.fg; ds.fast_ok = finite && ds.n_med == 0 ? 1 : 0;
    ds.light_kind = s->light_kind;
    for (int k = 0; k < 3; ++k) ds.light_c[k] = float(s->light_data[k]);
    ds.light_r = float(s->light_data[3]);
    for (int k = 0; k < 5; ++k) ds.light_rect[k] = s->light_data[k];
    for (int k = 0; k < 3; ++k) ds.bg[k] = s->background[k];
    G().scene = ds;
    G().have_scene = true;
    return 0;
}

// camera::resetCam, camera.h:32-54 (host arithmetic, as in the reference)
int rt_camera_from_lookat(const float lookfrom[3], const float lookat[3], const float vup[3], double focus_dist, double vfov,
                          double aspect_ratio, double aperture, RtCamera* out) {
    if (!lookfrom || !lookat || !vup || !out) return fail("rt_camera_from_lookat: null argument");
    V3 from = v3(lookfrom[0], lookfrom[1], lookfrom[2]), at = v3(lookat[0], lookat[1], lookat[2]), up = v3(vup[0], vup[1], vup[2]);
    double theta = vfov * kPi / 180.0;
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h;
    double viewport_width = aspect_ratio * viewport_height;
    V3 w = normalized(from - at);
    V3 u = normalized(cross(up, w));
    V3 v = cross(w, u);
    V3 horizontal = float(focus_dist * viewport_width) * u;
    V3 vertical = float(focus_dist * viewport_height) * v;
    V3 llc = from - horizontal / 2.0f - vertical / 2.0f - float(focus_dist) * w;
    auto put = [](float* o, V3 a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; };
    put(out->origin, from); put(out->lower_left_corner, llc); put(out->horizontal, horizontal); put(out->vertical, vertical);
    put(out->u, u); put(out->v, v); put(out->w, w);
    out->lens_radius = aperture / 2.0f;
    return 0;
}

int rt_set_camera(const RtCamera* c) {
    if (need_ready()) return 1;
    if (!c) return fail("rt_set_camera: null camera");
    auto get = [](const float* p) { return v3(p[0], p[1], p[2]); };
    G().cam.origin = get(c->origin); G().cam.llc = get(c->lower_left_corner); G().cam.horizontal = get(c->horizontal);
    G().cam.vertical = get(c->vertical); G().cam.u = get(c->u); G().cam.v = get(c->v); G().cam.w = get(c->w);
    G().cam.lens_radius = c->lens_radius;
    G().have_cam = true;
    return 0;
}

static int need_scene_cam() {
    if (need_ready()) return 1;
    if (!G().have_scene) return fail("no scene uploaded (rt_upload_scene)");
    if (!G().have_cam) return fail("no camera set (rt_set_camera)");
    return 0;
}

int rt_primary_hits(int w, int h, uint64_t seed, int32_t* d_prim, double* d_t) {
    if (need_scene_cam()) return 1;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return fail("rt_primary_hits: bad size %dx%d", w, h);
    int n_tiles = ((w + 15) / 16) * ((h + 15) / 16);
    if (begin_timed()) return 1;
    hipLaunchKernelGGL(primary_kernel, dim3(grid_for(n_tiles)), dim3(kBlock), 0, G().stream, G().scene, G().cam, w, h, seed, d_prim, d_t);
    return finish_timed(1);
}

int rt_hit_rays(int64_t n, const float* d_origin, const float* d_dir, double t_min, double t_max, int32_t* d_prim, double* d_t,
                float* d_p, float* d_normal, uint8_t* d_front) {
    if (need_ready()) return 1;
    if (!G().have_scene) return fail("no scene uploaded (rt_upload_scene)");
    if (n < 0 || (n > 0 && (!d_origin || !d_dir || !d_prim))) return fail("rt_hit_rays: bad arguments");
    if (n == 0) { G().last_ms = 0; G().last_launches = 0; return 0; }
    int blocks = int((n + kBlock - 1) / kBlock);
    if (begin_timed()) return 1;
    hipLaunchKernelGGL(hit_rays_kernel, dim3(grid_for(blocks)), dim3(kBlock), 0, G().stream, G().scene, (long long)n, d_origin, d_dir,
                       t_min, t_max, d_prim, d_t, d_p, d_normal, d_front);
    return finish_timed(1);
}

static int ensure_stack(int depth, int grid) {
    size_t need = size_t(depth > kLdsLevels ? depth - kLdsLevels : 1) * size_t(grid) * kBlock * sizeof(float4);
    if (need > G().stack_bytes) {
        HIP_OK(hipStreamSynchronize(G().stream));
        if (G().stack) HIP_OK(hipFree(G().stack));
        G().stack = nullptr; G().stack_bytes = 0;
        HIP_OK(hipMalloc(&G().stack, need));
        G().stack_bytes = need;
    }
    return 0;
}

int rt_trace_rays(int64_t n, const float* d_origin, const float* d_dir, int depth, uint64_t seed, uint32_t flags, float* d_rgb) {
    if (need_ready()) return 1;
    if (!G().have_scene) return fail("no scene uploaded (rt_upload_scene)");
    if (n < 0 || (n > 0 && (!d_origin || !d_dir || !d_rgb))) return fail("rt_trace_rays: bad arguments");
    if (depth < 0 || depth > 128) return fail("rt_trace_rays: depth %d outside 0..128", depth);
    if (n == 0) { G().last_ms = 0; G().last_launches = 0; return 0; }
    if (settle()) return 1;
    int grid = grid_for(int((n + kBlock - 1) / kBlock));
    if (ensure_stack(depth, grid)) return 1;
    if (begin_timed()) return 1;
    hipLaunchKernelGGL(trace_rays_kernel, dim3(grid), dim3(kBlock), 0, G().stream, G().scene, (long long)n, d_origin, d_dir, depth, seed,
                       flags, d_rgb, static_cast<float4*>(G().stack));
    return finish_timed(1);
}

// scratch_dev: nullptr = the library's own buffers (counter block from rt_init, path stack grown by ensure_stack);
// otherwise the device address of the file-scope `scratch` array (rt_render_noalloc): no allocation on this path.
// How many of a frame's tiles go through the colour buffer of the frame's end (render_pass has the reasons), and whether that is
// less than half of what the frame would like (then the other tiles are rendered in sample slices instead of as whole pixels).
static long long tail_tiles_wanted(long long n_tiles, long long waves, int spp_count, bool* tail_short) {
    // all tiles: frames of up to four tiles per wave (1280 x 720 at 64 samples: 16.2 against 20.2 ms with half of them), and frames
    // whose whole tiles are large against the frame (n_tiles x 64 <= waves x samples: the 1/8 band share of a 4K frame of 1024
    // samples takes 403 ms with all 16200 tiles in the tail, 1.02 x its share of the whole frame, and 525 ms with half of them)
    const bool all = n_tiles <= 4 * waves || n_tiles * 64 <= waves * (long long)spp_count;
    // (six per wave where the limit allows: frames whose tiles differ much in cost - the mixed 4K frame in passes of 32 .. 128 samples -
    // gain 1 - 2 % over four, the 485-sphere frame loses at most 0.3 %; whether the other tiles are rendered as whole pixels or in
    // slices is still decided against four per wave, which is what the 256-sample frames were tuned with)
    const long long t_need = all ? n_tiles : (n_tiles / 2 < 4 * waves ? n_tiles / 2 : 4 * waves);
    long long t = all ? n_tiles : (n_tiles / 2 < 6 * waves ? n_tiles / 2 : 6 * waves);
    const long long per_tile = (long long)spp_count * 64 * 12;   // three floats per colour
    long long cap = (G().tail_limit_mb << 20) / per_tile;
    if (cap * spp_count * 64 > 0xffffffffll) cap = 0xffffffffll / (spp_count * 64ll);   // (a colour's index in the buffer is 32 bits wide: limits beyond 48 GiB)
    *tail_short = false;
    // (4 / 9 of the tail it would like still beats slices: 4K / 256 spp in order of cost 378.9 ms with 1400 MiB as with 1536, 389.3 in slices
    // with 1344; the mixed frame at 1 024 samples per call is better off in slices below that)
    if (t > cap) { *tail_short = 9 * cap < 4 * t_need; t = cap; }
    return t;
}

// phase: 0 = a frame in one pass; 1 / 2 = the measuring pass (one sample per pixel, rays per tile to order_buf) and the rest of a frame
// whose tiles are handed out in order of cost (render_impl below): one timed region, the first pass neither ends it nor waits.
struct OrderBufs { unsigned int* cost; int* idx; unsigned int* cost_sorted; int* idx_sorted; int* order; void* tmp; size_t tmp_bytes; };
static OrderBufs order_bufs(int n_tiles, size_t tmp_bytes) {
    const size_t arr = (size_t(n_tiles) * 4 + 255) / 256 * 256;
    char* b = static_cast<char*>(G().order_buf);
    return OrderBufs{reinterpret_cast<unsigned int*>(b), reinterpret_cast<int*>(b + arr), reinterpret_cast<unsigned int*>(b + 2 * arr),
                     reinterpret_cast<int*>(b + 3 * arr), reinterpret_cast<int*>(b + 4 * arr), b + 5 * arr, tmp_bytes};
}
static int render_pass(const char* who, int w, int h, int y_begin, int y_count, int band_first, int band_stride, int tiles_y, int spp_begin,
                       int spp_count, int depth, uint64_t seed, uint32_t flags, float* d_accum, uint64_t* d_rays, void* scratch_dev, int phase) {
    if (spp_begin < 0 || spp_count < 0) return fail("%s: bad sample range [%d,+%d)", who, spp_begin, spp_count);
    if (depth < 0 || depth > 128) return fail("%s: depth %d outside 0..128", who, depth);
    if (!d_accum) return fail("%s: null accumulator", who);
    if (tiles_y == 0) { G().last_ms = 0; G().last_launches = 0; return 0; }
    if (spp_count == 0 || depth == 0) {   // no ray is traced: every sample is black (calc.cpp:96-97)
        int n = w * tiles_y * kTile;
        if (begin_timed()) return 1;
        hipLaunchKernelGGL(render_black_kernel, dim3((n + 255) / 256), dim3(256), 0, G().stream, w, y_begin, y_count, band_first, band_stride, tiles_y,
                           spp_count, (flags & RT_RENDER_CONTINUE) ? (flags | RT_RENDER_ACCUMULATE) : flags,   // (continuing with black samples adds +0)
                           reinterpret_cast<float4*>(d_accum));
        return finish_timed(1, (flags & RT_RENDER_NOWAIT) != 0);
    }
    const int n_tiles = ((w + kTile - 1) / kTile) * tiles_y;   // one 8x8 tile per wave at a time
    int resident = G().prop.multiProcessorCount * G().render_blocks_per_cu;
#ifdef RT_STAMPS
    if (const char* e = getenv("RT_BPC")) resident = G().prop.multiProcessorCount * atoi(e);   // diagnostic: blocks per CU
#endif
    if (settle()) return 1;   // a launch still in flight owns the counters and the stack
    // Sample slices (render_kernel): make_slices. RT_RENDER_ACCUMULATE adds the pass's own sum to what was there, which a chain of
    // slices through the accumulator cannot express: one slice. rt_set_sample_slices changes the smallest slice (0: whole pixels)
    // and the slice limit.
    if ((flags & RT_RENDER_ACCUMULATE) && (flags & RT_RENDER_CONTINUE)) return fail("%s: RT_RENDER_ACCUMULATE and RT_RENDER_CONTINUE exclude each other", who);
    if ((long long)n_tiles * 64 > kMaxWorkItems)
        return fail("%s: %d tiles of 8x8 pixels exceed what one call hands out (33.5 million); split the rows over several calls", who, n_tiles);
    RenderArgs ra{};
    const long long waves = (long long)resident * (kBlock / 64);
    // Two ways to keep every lane fed until a frame ends, both leaving the bits of a pixel summed by one thread:
    // * the tail (render_kernel): n_tail tiles spread evenly over the frame are kept for its end and rendered in units of 8 samples
    //   per pixel whose colours go through a buffer (16 bytes per sample), parallel over a pixel's samples as well. Measured best
    //   (tools/tail_sweep.py, round 2): half the frame's tiles, at most six per resident wave; all tiles of small frames (below). That costs 768 bytes per tile and sample (1 KiB before the colours were stored as three floats): 1 GiB for a 1080p frame of 64 samples, 4 GiB for 16384 tiles of
    //   a 4K frame of 256, 16 GiB at 1024 samples - so it is bounded (rt_set_tail_limit, default 2 GiB);
    // * sample slices (make_slices): where that limit leaves less than half the tail the frame would like, the other tiles are rendered slice by slice through
    //   the accumulator instead of as whole pixels: no memory to speak of, units that shrink towards the frame's end, but a
    //   tile's slices come one after the other, so the slowest tiles' last slices keep a few waves busy after the others have
    //   left (15 % of a 1080p frame of the triangle mesh without a tail), and the bookkeeping costs 1 - 2 %: second choice.
    int chunk = 8, n_tail = 0, chunks = 1;
    bool tail_short = true;             // the tail is smaller than the frame would like (or there is none)
    if (!scratch_dev && G().tail_limit_mb > 0 && spp_count >= 2 * chunk) {
        const long long per_tile = (long long)spp_count * 64 * 12;   // three floats per colour
        long long t = tail_tiles_wanted(n_tiles, waves, spp_count, &tail_short);
        chunks = (spp_count + chunk - 1) / chunk;
        if (t * chunks * 64 > kMaxWorkItems / 2) t = 0;
        n_tail = int(t);
        if (n_tail > 0 && size_t(n_tail * per_tile) > G().tail_bytes) {
            if (G().tail) HIP_OK(hipFree(G().tail));
            G().tail = nullptr; G().tail_bytes = 0;
            // a device that is short of memory gets a smaller tail, in the end none: the frame is the same, only its last part slower
            while (n_tail > 0 && hipMalloc(&G().tail, size_t(n_tail * per_tile)) != hipSuccess) {
                (void)hipGetLastError();
                G().tail = nullptr;
                n_tail /= 2;
            }
            if (2 * (long long)n_tail < t) tail_short = true;
            G().tail_bytes = size_t(n_tail * per_tile);
        }
    }
    if (n_tail == 0) chunks = 1;
    const int n_bulk = n_tiles - n_tail;
    {
        const int smin = G().slice_min;
        int n = 0;
        ra.slice_begin[0] = 0;
        const bool one = !(tail_short || G().slices_always) || n_bulk == 0 || smin <= 0 || spp_count < 2 * smin || (flags & RT_RENDER_ACCUMULATE) || G().slice_max < 2;
        if (!one) {
            int smax = G().slice_max < kMaxSlices ? G().slice_max : kMaxSlices;
            if (scratch_dev && (long long)n_bulk > (long long)kScratchDoneTiles) smax = 1;   // (the scratch array holds the state words of that many tiles)
            while (smax > 1 && (long long)n_bulk * smax * 64 > kMaxWorkItems / 2) --smax;   // the work counter is 32 bits wide
            if (n_bulk > int(kUnitTileMask)) smax = 1;                                       // (a unit word holds the tile in 26 bits)
            if (smax >= 2) n = make_slices(spp_count, n_bulk, waves, smin, smax, ra.slice_begin);
        }
        if (n == 0) { n = 1; ra.slice_begin[1] = spp_count; }
        ra.n_slices = n; ra.n_tiles = n_bulk;
    }
    ra.n_tail_tiles = n_tail; ra.tail_chunk = chunk; ra.tail_chunks = chunks;
    ra.tail_stride = n_tail > 0 ? n_tiles / n_tail : 1;
    ra.tail = static_cast<float*>(G().tail);
    const size_t queue_bytes = (size_t(n_bulk > 0 ? n_bulk : 1) * sizeof(unsigned int) + 255) / 256 * 256;   // tile_state: one word per tile
    const long long units = (long long)n_bulk * ra.n_slices + (long long)n_tail * chunks;
    const long long want = (units + 3) / 4;
    int grid = want < resident ? int(want) : resident;
    void* counters = G().counters;
    float4* stack = nullptr;
    unsigned int* queue = nullptr;
    if (scratch_dev) {   // the stack part of `scratch` holds kNoAllocMaxDepth - kLdsLevels levels of kNoAllocMaxGrid blocks
        if (grid > kNoAllocMaxGrid) grid = kNoAllocMaxGrid;
        counters = scratch_dev;
        stack = reinterpret_cast<float4*>(static_cast<char*>(scratch_dev) + kScratchCounterBytes);
        queue = reinterpret_cast<unsigned int*>(static_cast<char*>(scratch_dev) + kScratchCounterBytes + kScratchStackBytes);
    } else {
        if (ensure_stack(depth, grid)) return 1;
        stack = static_cast<float4*>(G().stack);
        if (ra.n_slices > 1 && queue_bytes > G().queue_bytes) {
            if (G().queue) HIP_OK(hipFree(G().queue));
            G().queue = nullptr; G().queue_bytes = 0;
            HIP_OK(hipMalloc(&G().queue, queue_bytes));
            G().queue_bytes = queue_bytes;
        }
        queue = static_cast<unsigned int*>(G().queue);
    }
    ra.W = w; ra.H = h; ra.y_begin = y_begin; ra.y_count = y_count; ra.spp_begin = spp_begin; ra.spp_count = spp_count; ra.depth = depth;
    ra.band_first = band_first; ra.band_stride = band_stride; ra.tiles_y = tiles_y;
    ra.seed = seed; ra.flags = flags;
    ra.inv_w = 1.0 / double(float(w)); ra.inv_h = 1.0 / double(float(h));
    ra.accum = reinterpret_cast<float4*>(d_accum); ra.stack = stack;
    ra.tile_state = ra.n_slices > 1 ? queue : nullptr;
    ra.rays_out = reinterpret_cast<unsigned long long*>(d_rays);
    ra.work_counter = static_cast<unsigned int*>(counters);
    ra.mismatches = reinterpret_cast<unsigned long long*>(static_cast<char*>(counters) + 8);
    const bool verify = (flags & RT_RENDER_VERIFY) != 0;
    if (verify && G().scene.n_med > 0) return fail("%s: RT_RENDER_VERIFY has nothing to compare in a scene with a constant_medium (list-order scan only)", who);
    // the counter block keeps the last call's contents: the tile counter (and the verify count) must start at zero
#ifdef RT_STAMPS
    ra.stamps = reinterpret_cast<unsigned long long*>(static_cast<char*>(counters) + 16);
    HIP_OK(hipMemsetAsync(counters, 0, 128, G().stream));
#else
    ra.stamps = nullptr;
    HIP_OK(hipMemsetAsync(counters, 0, 16, G().stream));
#endif
    if (ra.n_slices > 1) {   // no slice of any tile is finished, none has been left to another wave
        HIP_OK(hipMemsetAsync(queue, 0, queue_bytes, G().stream));
    }
    int extra_launches = 0;
    if (phase == 1) ra.tile_cost = order_bufs(n_tiles, 0).cost;
    if (phase != 2 && begin_timed()) return 1;
    if (phase == 2) {
        const OrderBufs ob = order_bufs(n_tiles, 0);
        hipLaunchKernelGGL(tile_arrange_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, G().stream, n_tiles, n_tail, ra.tail_stride, ob.idx_sorted, ob.order);
        ra.tile_order = ob.order;
        extra_launches = G().phase_launches + 3;   // (the measuring pass, the sort's kernels counted as one, the index fill, this one)
    }
    const bool many_rects = G().scene.n_rect >= kRectFlagMin;
    const bool bvh = G().scene.tbvh != nullptr && !many_rects;   // (a scene with many rects AND a triangle BVH takes the flat triangle path)
    const bool chain = ra.n_slices > 1;
    // block-sorted shading (render_kernel, SORT): on request only. It raises the lane utilisation of the built-in scene's
    // frame from 37 to 48 % and executes 16 % fewer instructions, but the two barriers per trip leave issue slots empty: a tie at 14.0 ms
    const bool has_tris = (G().scene.tg.n_epairs | G().scene.tg.n_fpairs) != 0;
    const bool sort = G().sort_mode > 0 && !verify && G().scene.n_med == 0 && !has_tris && !many_rects;
    ra.sort_guard = G().scene.sg.n_fpairs != 0 ? 1 : 0;
#define RT_LAUNCH(...) hipLaunchKernelGGL((render_kernel<__VA_ARGS__>), dim3(grid), dim3(kBlock), 0, G().stream, G().scene, G().cam, ra)
    if (G().scene.n_med > 0) { if (chain) RT_LAUNCH(false, false, 0, 1, 0, true); else RT_LAUNCH(false, false, 0, 1, 0, false); }
    else if (verify && bvh) RT_LAUNCH(true, false, 0, 0, 1, true);
    else if (bvh) { if (chain) RT_LAUNCH(false, false, 0, 0, 1, true); else RT_LAUNCH(false, false, 0, 0, 1, false); }
    else if (verify && many_rects) RT_LAUNCH(true, false, 1, 0, 0, true);
    else if (verify) RT_LAUNCH(true, false, 0, 0, 0, true);
    else if (many_rects && !has_tris) { if (chain) RT_LAUNCH(false, false, 1, 0, 3, true); else RT_LAUNCH(false, false, 1, 0, 3, false); }
    else if (many_rects) { if (chain) RT_LAUNCH(false, false, 1, 0, 0, true); else RT_LAUNCH(false, false, 1, 0, 0, false); }
    else if (sort) { if (chain) RT_LAUNCH(false, false, 0, 0, 3, true, true); else RT_LAUNCH(false, false, 0, 0, 3, false, true); }
    else if (!has_tris) { if (chain) RT_LAUNCH(false, false, 0, 0, 3, true); else RT_LAUNCH(false, false, 0, 0, 3, false); }
    else { if (chain) RT_LAUNCH(false, false, 0, 0, 0, true); else RT_LAUNCH(false, false, 0, 0, 0, false); }
#undef RT_LAUNCH
    if (n_tail > 0) hipLaunchKernelGGL(tail_sum_kernel, dim3((n_tail * 64 + 255) / 256), dim3(256), 0, G().stream, ra);
    if (phase == 1) { HIP_OK(hipGetLastError()); G().phase_launches = n_tail > 0 ? 2 : 1; return 0; }
    return finish_timed((n_tail > 0 ? 2 : 1) + extra_launches, (flags & RT_RENDER_NOWAIT) != 0);
}

// Tiles in order of cost (rt_set_tile_order(1)). A frame ends when its last unit does, and the units are handed out in
// the order of the tiles in the image, whatever they cost: the last whole-pixel tiles of a frame of many samples finish anywhere in
// the time of one expensive tile, and the colour buffer's filler has to cover that. Here the frame's first sample is rendered as a
// pass of its own, which also counts the rays of every tile; the tiles are sorted by that count (device radix sort, same stream), and
// the other samples CONTINUE the sums (RT_RENDER_CONTINUE: the bits of one call) with the heaviest tiles first and every
// tail_stride-th tile of the sorted order as the tail. No sample is rendered twice and none is skipped.
static int render_impl(const char* who, int w, int h, int y_begin, int y_count, int band_first, int band_stride, int tiles_y, int spp_begin,
                       int spp_count, int depth, uint64_t seed, uint32_t flags, float* d_accum, uint64_t* d_rays, void* scratch_dev = nullptr) {
    const long long n_tiles_ll = (long long)((w + kTile - 1) / kTile) * tiles_y;
    const long long waves = (long long)G().prop.multiProcessorCount * G().render_blocks_per_cu * (kBlock / 64);
    // Where it pays (measured): calls of 64 samples and more whose filler is thin - under five tail tiles per resident wave,
    // i.e. 1080p frames and 4K frames whose colour buffer is capped by the limit. Frames with more filler end evenly as they are and
    // would only pay for the extra pass (the 485-sphere 4K frame at 32 .. 128 samples: + 0.3 .. 0.5 %).
    // (mode 2, for tests and tools: every call of two samples and more, whatever the frame)
    const bool always = G().order_mode == 2;
    bool ordered = G().order_mode > 0 && !scratch_dev && spp_begin >= 0 && spp_count >= (always ? 2 : 64) && depth > 0 && depth <= 128 && d_accum && tiles_y > 0 &&
                   (flags & (RT_RENDER_VERIFY | RT_RENDER_ACCUMULATE)) == 0u && (always || n_tiles_ll >= 2 * waves) && n_tiles_ll * 64 <= kMaxWorkItems;
    if (ordered && G().order_mode == 1) {
        bool tail_short = false;
        const long long t = G().tail_limit_mb > 0 ? tail_tiles_wanted(n_tiles_ll, waves, spp_count - 1, &tail_short) : 0;
        ordered = t < 5 * waves && t < n_tiles_ll;   // (a frame that is all tail has no long units to put first: the 1/8 band share of the mixed 4K frame, 313 against 317 ms)
    }
    if (!ordered) return render_pass(who, w, h, y_begin, y_count, band_first, band_stride, tiles_y, spp_begin, spp_count, depth, seed, flags, d_accum, d_rays, scratch_dev, 0);
    const int n_tiles = int(n_tiles_ll);
    size_t tmp_bytes = 0;
    {
        unsigned int* nokey = nullptr; int* noval = nullptr;
        HIP_OK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, nokey, nokey, noval, noval, n_tiles, 0, 32, G().stream));
    }
    const size_t arr = (size_t(n_tiles) * 4 + 255) / 256 * 256, need = 5 * arr + tmp_bytes;
    if (need > G().order_bytes) {
        if (settle()) return 1;
        HIP_OK(hipStreamSynchronize(G().stream));
        if (G().order_buf) HIP_OK(hipFree(G().order_buf));
        G().order_buf = nullptr; G().order_bytes = 0;
        HIP_OK(hipMalloc(&G().order_buf, need));
        G().order_bytes = need;
    }
    if (settle()) return 1;
    const OrderBufs ob = order_bufs(n_tiles, tmp_bytes);
    HIP_OK(hipMemsetAsync(ob.cost, 0, size_t(n_tiles) * 4, G().stream));
    if (render_pass(who, w, h, y_begin, y_count, band_first, band_stride, tiles_y, spp_begin, 1, depth, seed, flags & ~RT_RENDER_NOWAIT, d_accum, d_rays, nullptr, 1)) return 1;
    hipLaunchKernelGGL(tile_iota_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, G().stream, n_tiles, ob.idx);
    HIP_OK(hipcub::DeviceRadixSort::SortPairsDescending(ob.tmp, tmp_bytes, ob.cost, ob.cost_sorted, ob.idx, ob.idx_sorted, n_tiles, 0, 32, G().stream));
    return render_pass(who, w, h, y_begin, y_count, band_first, band_stride, tiles_y, spp_begin + 1, spp_count - 1, depth, seed, flags | RT_RENDER_CONTINUE, d_accum,
                       d_rays, nullptr, 2);
}

int rt_render(int w, int h, int y_begin, int y_count, int spp_begin, int spp_count, int depth, uint64_t seed, uint32_t flags,
              float* d_accum, uint64_t* d_rays) {
    if (need_scene_cam()) return 1;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return fail("rt_render: bad size %dx%d", w, h);
    if (y_begin < 0 || y_count < 0 || y_begin + y_count > h) return fail("rt_render: rows [%d,+%d) outside 0..%d", y_begin, y_count, h);
    return render_impl("rt_render", w, h, y_begin, y_count, 0, 1, (y_count + kTile - 1) / kTile, spp_begin, spp_count, depth, seed, flags, d_accum,
                       d_rays);
}

// rt_render without any allocation: scratch (tile counter, HBM part of the path stack) is the file-scope
// `scratch` array, reached through the per-device address rt_init resolved. include/ray_amd.h has the contract.
int rt_render_noalloc(int w, int h, int y_begin, int y_count, int spp_begin, int spp_count, int depth, uint64_t seed, uint32_t flags,
                      float* d_accum, uint64_t* d_rays) {
    if (need_scene_cam()) return 1;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return fail("rt_render_noalloc: bad size %dx%d", w, h);
    if (y_begin < 0 || y_count < 0 || y_begin + y_count > h) return fail("rt_render_noalloc: rows [%d,+%d) outside 0..%d", y_begin, y_count, h);
    if (depth > kNoAllocMaxDepth)
        return fail("rt_render_noalloc: depth %d exceeds the limit of %d (the static scratch holds the path stack of %d bounces; use rt_render)",
                    depth, kNoAllocMaxDepth, kNoAllocMaxDepth);
    if (flags & RT_RENDER_VERIFY) return fail("rt_render_noalloc: RT_RENDER_VERIFY is an rt_render flag");
    void* sp = G().device >= 0 && G().device < kMaxDevices ? g_scratch_ptr[G().device] : nullptr;
    if (!sp) return fail("rt_render_noalloc: no scratch address for device %d (rt_init resolves it)", G().device);
    return render_impl("rt_render_noalloc", w, h, y_begin, y_count, 0, 1, (y_count + kTile - 1) / kTile, spp_begin, spp_count, depth, seed, flags,
                       d_accum, d_rays, sp);
}

int rt_render_bands(int w, int h, int band_first, int band_stride, int spp_begin, int spp_count, int depth, uint64_t seed, uint32_t flags,
                    float* d_accum, uint64_t* d_rays) {
    if (need_scene_cam()) return 1;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return fail("rt_render_bands: bad size %dx%d", w, h);
    if (band_first < 0 || band_stride <= 0) return fail("rt_render_bands: bad bands first=%d stride=%d", band_first, band_stride);
    const int n_bands = (h + kTile - 1) / kTile;
    const int mine = band_first >= n_bands ? 0 : (n_bands - band_first + band_stride - 1) / band_stride;
    return render_impl("rt_render_bands", w, h, 0, h, band_first, band_stride, mine, spp_begin, spp_count, depth, seed, flags, d_accum, d_rays);
}

// RayView::renderOneRay for a list of pixels (rayview.cpp:67-87): spp_count samples of each listed pixel,
// the pixel's accumulator overwritten (or added to with RT_RENDER_ACCUMULATE), other pixels untouched.
int rt_render_pixels(int w, int h, int64_t n, const int32_t* d_pixels, int spp_begin, int spp_count, int depth, uint64_t seed, uint32_t flags,
                     float* d_accum, uint64_t* d_rays) {
    if (need_scene_cam()) return 1;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return fail("rt_render_pixels: bad size %dx%d", w, h);
    if (n < 0 || n > kMaxWorkItems) return fail("rt_render_pixels: bad pixel count %lld", (long long)n);
    if (spp_begin < 0 || spp_count < 0) return fail("rt_render_pixels: bad sample range [%d,+%d)", spp_begin, spp_count);
    if (depth < 0 || depth > 128) return fail("rt_render_pixels: depth %d outside 0..128", depth);
    if (!d_accum || (n > 0 && !d_pixels)) return fail("rt_render_pixels: null pointer");
    if (flags & (RT_RENDER_VERIFY | RT_RENDER_CONTINUE)) return fail("rt_render_pixels: RT_RENDER_VERIFY and RT_RENDER_CONTINUE are rt_render flags");
    if (n == 0) { G().last_ms = 0; G().last_launches = 0; return 0; }
    if (spp_count == 0 || depth == 0) {   // no ray is traced: every sample is black (calc.cpp:96-97)
        if (begin_timed()) return 1;
        hipLaunchKernelGGL(render_black_pixels_kernel, dim3(int((n + 255) / 256)), dim3(256), 0, G().stream, int(n), w * h, d_pixels, spp_count, flags,
                           reinterpret_cast<float4*>(d_accum));
        return finish_timed(1);
    }
    if (settle()) return 1;
    const int resident = G().prop.multiProcessorCount * G().render_blocks_per_cu;
    const int want = int((n + 255) / 256);
    const int grid = want < resident ? want : resident;
    if (ensure_stack(depth, grid)) return 1;
    RenderArgs ra{};
    ra.W = w; ra.H = h; ra.y_begin = 0; ra.y_count = h; ra.spp_begin = spp_begin; ra.spp_count = spp_count; ra.depth = depth;
    ra.band_first = 0; ra.band_stride = 1; ra.tiles_y = 0;
    ra.seed = seed; ra.flags = flags;
    ra.inv_w = 1.0 / double(float(w)); ra.inv_h = 1.0 / double(float(h));
    ra.accum = reinterpret_cast<float4*>(d_accum); ra.stack = static_cast<float4*>(G().stack);
    ra.rays_out = reinterpret_cast<unsigned long long*>(d_rays);
    ra.work_counter = static_cast<unsigned int*>(G().counters);
    ra.mismatches = nullptr; ra.stamps = nullptr;
    ra.pixels = d_pixels; ra.n_pixels = int(n);
    ra.n_slices = 1; ra.n_tiles = 0; ra.slice_begin[1] = spp_count; ra.tile_state = nullptr;
    ra.n_tail_tiles = 0; ra.tail_chunk = 8; ra.tail_chunks = 1; ra.tail_stride = 1; ra.tail = nullptr;
    HIP_OK(hipMemsetAsync(G().counters, 0, 4, G().stream));
    // The list is rendered 64 entries per wave in list order. The reference draws its pixels at random (rayview.cpp:69-70), and a
    // wave whose 64 primary rays go 64 different ways filters the union of what they need: 43 % slower than the same pixels in
    // tile order on the 485-sphere frame (tools/pixels_order.py). The result of a pixel does not depend on where it stands in the
    // list (its samples are keyed by pixel and sample; listings of one pixel add up or overwrite with equal values), so the
    // list is sorted by 8x8 tile first: a stable radix sort of (tile, pixel) pairs on the same stream, in a grow-only workspace.
    int launches = 1;
    const long long n_tiles_ll = (long long)((w + kTile - 1) / kTile) * ((h + kTile - 1) / kTile);
    if (n >= 2048 && n_tiles_ll < 0x7fffffffLL) {
        const int n_tiles = int(n_tiles_ll), tiles_x = (w + kTile - 1) / kTile;
        int end_bit = 1;
        while ((1ll << end_bit) <= n_tiles_ll) ++end_bit;      // keys 0 .. n_tiles
        size_t tmp_bytes = 0;
        uint32_t* nokey = nullptr; int* noval = nullptr;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, nokey, nokey, noval, noval, int(n), 0, end_bit, G().stream));
        const size_t arr = (size_t(n) * 4 + 255) / 256 * 256, need = 3 * arr + tmp_bytes;
        if (need > G().pix_sort_bytes) {
            if (settle()) return 1;
            if (G().pix_sort) HIP_OK(hipFree(G().pix_sort));
            G().pix_sort = nullptr; G().pix_sort_bytes = 0;
            HIP_OK(hipMalloc(&G().pix_sort, need));
            G().pix_sort_bytes = need;
        }
        char* base = static_cast<char*>(G().pix_sort);
        uint32_t* keys_in = reinterpret_cast<uint32_t*>(base); uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + arr);
        int* sorted = reinterpret_cast<int*>(base + 2 * arr);
        if (begin_timed()) return 1;
        hipLaunchKernelGGL(pixel_tile_keys_kernel, dim3(int((n + 255) / 256)), dim3(256), 0, G().stream, int(n), w, h, tiles_x, n_tiles, d_pixels, keys_in);
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(base + 3 * arr, tmp_bytes, keys_in, keys_out, d_pixels, sorted, int(n), 0, end_bit, G().stream));
        ra.pixels = sorted;
        launches = 3;
    } else if (begin_timed()) return 1;
    if (G().scene.n_med > 0) hipLaunchKernelGGL((render_kernel<false, true, 0, 1, 0, false>), dim3(grid), dim3(kBlock), 0, G().stream, G().scene, G().cam, ra);
    else if (G().scene.tbvh != nullptr && G().scene.n_rect < kRectFlagMin) hipLaunchKernelGGL((render_kernel<false, true, 0, 0, 1, false>), dim3(grid), dim3(kBlock), 0, G().stream, G().scene, G().cam, ra);
    else if (G().scene.n_rect >= kRectFlagMin) hipLaunchKernelGGL((render_kernel<false, true, 1, 0, 0, false>), dim3(grid), dim3(kBlock), 0, G().stream, G().scene, G().cam, ra);
    else hipLaunchKernelGGL((render_kernel<false, true, 0, 0, 0, false>), dim3(grid), dim3(kBlock), 0, G().stream, G().scene, G().cam, ra);
    return finish_timed(launches);
}

#ifdef RT_STAMPS
// diagnostic builds only (tools/wave_times.py): start and end time of every wave of the last render
int rt_debug_wave_times(uint64_t* out, int n_waves) {
    if (need_ready()) return 1;
    HIP_OK(hipStreamSynchronize(G().stream));
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), size_t(n_waves) * 16, 0, hipMemcpyDeviceToHost));
    return 0;
}
// diagnostic builds only (tools/): section cycles and pass counts of the last render
int rt_debug_stamps(uint64_t out[12]) {
    if (need_ready()) return 1;
    HIP_OK(hipMemcpyAsync(out, static_cast<char*>(G().counters) + 16, 96, hipMemcpyDeviceToHost, G().stream));
    HIP_OK(hipStreamSynchronize(G().stream));
    return 0;
}
#endif

int rt_verify_mismatches(uint64_t* count) {
    if (need_ready()) return 1;
    if (settle()) return 1;
    if (!count) return fail("rt_verify_mismatches: null argument");
    HIP_OK(hipMemcpyAsync(count, static_cast<char*>(G().counters) + 8, 8, hipMemcpyDeviceToHost, G().stream));
    HIP_OK(hipStreamSynchronize(G().stream));
    return 0;
}

int rt_tonemap(int w, int h, const float* d_accum, int n_samples, uint32_t* d_rgb32) {
    if (need_ready()) return 1;
    if (settle()) return 1;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return fail("rt_tonemap: bad size %dx%d", w, h);
    if (!d_accum || !d_rgb32) return fail("rt_tonemap: null pointer");
    int n = w * h;
    hipLaunchKernelGGL(tonemap_kernel, dim3((n + 255) / 256), dim3(256), 0, G().stream, w, h, reinterpret_cast<const float4*>(d_accum),
                       n_samples, d_rgb32);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(G().stream));
    return 0;
}

int rt_last_kernel_ms(float* ms, int* launches) {
    if (G().ready && G().pending && settle()) return 1;
    if (ms) *ms = G().last_ms;
    if (launches) *launches = G().last_launches;
    return 0;
}

}  // extern "C"
