// libray_amd.so -- MI355X-native implementation of the joblobob/ray per-pixel path behind the C ABI
// of include/ray_amd.h: camera::get_ray (camera.h:21-30) -> hitFromList (hitPosition.cpp:6-22) over
// structure-of-arrays primitives -> calc::ray_color (calc.cpp:90-131) as an iterative bounce loop.
// gfx950 only; compiled with -ffp-contract=off (see rt_device.hpp for the arithmetic contract).
//
// No CPU fallback: every compute entry point needs rt_init() to have bound a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ray_amd.h"
#include "rt_device.hpp"

using namespace rt;

// ================================================================================================
// Device-side scene (HBM layout, DESIGN.md "data layout")
// ================================================================================================
struct DevScene {
    int n_sph, n_rect, n_tri, n_prims;
    const float4* __restrict__ sph;        // [n_sph]  (cx, cy, cz, r)
    const int* __restrict__ sph_prim;      // [n_sph]  flattened primitive index, ascending
    const double* __restrict__ rect;       // [n_rect][5] (a0, a1, b0, b1, k)
    const int* __restrict__ rect_axis;     // [n_rect] 0 = XY (z = k), 1 = XZ (y = k), 2 = YZ (x = k)
    const int* __restrict__ rect_prim;     // [n_rect]
    const float* __restrict__ tri;         // [n_tri][12] v0, e1, e2, unit normal
    const int* __restrict__ tri_prim;      // [n_tri]
    const int* __restrict__ prim_kind;     // [n_prims] 0 sphere, 1 rect, 2 triangle (device kinds)
    const int* __restrict__ prim_local;    // [n_prims] index inside its kind's arrays
    const int* __restrict__ prim_mat;      // [n_prims]
    const int* __restrict__ mat_kind;      // [n_mats]
    const float* __restrict__ mat_rgb;     // [n_mats][3]
    const float* __restrict__ mat_fuzz;    // [n_mats]
    const double* __restrict__ mat_ir;     // [n_mats]
    int light_kind;                        // RT_LIGHT_BASE, RT_PRIM_SPHERE, RT_PRIM_XY_RECT, RT_PRIM_XZ_RECT
    float light_c[3], light_r;             // sphere light
    double light_rect[5];                  // rect light
    float bg[3];
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
RT_DEV bool reject_t(double t, double t_min, double bt, int idx, int bi) {
    return t < t_min || bt < t || (t == bt && idx < bi);
}

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
RT_DEV bool rect_t(const Ray& r, int axis, const double* __restrict__ q, double t_min, double bt, int idx, int bi,
                   double* t_out) {
    float ok, dk, oa, da, ob, db;
    if (axis == 0) { ok = r.o.z; dk = r.d.z; oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
    else if (axis == 1) { ok = r.o.y; dk = r.d.y; oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
    else { ok = r.o.x; dk = r.d.x; oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
    double t = (q[4] - double(ok)) / double(dk);
    if (reject_t(t, t_min, bt, idx, bi)) return false;
    double a = double(oa) + t * double(da);
    double b = double(ob) + t * double(db);
    if (a < q[0] || a > q[1] || b < q[2] || b > q[3]) return false;
    *t_out = t;
    return true;
}

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
    if (!(t >= t_min && t <= bt) || (t == bt && idx < bi)) return false;
    *t_out = t;
    return true;
}

struct Hit {
    double t;
    int prim;   // flattened index, -1 = miss
};

// hitFromList (hitPosition.cpp:6-22) over the per-kind arrays. For totally ordered t the sequential
// scan equals "smallest t, ties to the later entry", which is what the grouped scan computes. If any
// accepted t is NaN the list order matters (a NaN closest_so_far accepts everything after it), so
// that lane redoes the scan in list order -- rare (a ray component exactly 0 on a plane it lies in).
__device__ __noinline__ Hit closest_hit_listorder(const DevScene& sc, const Ray& r, double t_min, double t_max) {
    Hit h; h.t = t_max; h.prim = -1;
    for (int i = 0; i < sc.n_prims; ++i) {
        int k = sc.prim_kind[i], j = sc.prim_local[i];
        double t; bool ok;
        if (k == 0) ok = sphere_t(r, sc.sph[j], t_min, h.t, i, h.prim, &t);
        else if (k == 1) ok = rect_t(r, sc.rect_axis[j], sc.rect + 5 * j, t_min, h.t, i, h.prim, &t);
        else ok = tri_t(r, sc.tri + 12 * j, t_min, h.t, i, h.prim, &t);
        if (ok) { h.t = t; h.prim = i; }
    }
    return h;
}

__device__ __forceinline__ Hit closest_hit(const DevScene& sc, const Ray& r, double t_min, double t_max) {
    Hit h; h.t = t_max; h.prim = -1;
    bool saw_nan = false;
    for (int j = 0; j < sc.n_sph; ++j) {
        double t;
        int idx = sc.sph_prim[j];
        if (sphere_t(r, sc.sph[j], t_min, h.t, idx, h.prim, &t)) {
            if (t != t) saw_nan = true; else { h.t = t; h.prim = idx; }
        }
    }
    for (int j = 0; j < sc.n_rect; ++j) {
        double t;
        int idx = sc.rect_prim[j];
        if (rect_t(r, sc.rect_axis[j], sc.rect + 5 * j, t_min, h.t, idx, h.prim, &t)) {
            if (t != t) saw_nan = true; else { h.t = t; h.prim = idx; }
        }
    }
    for (int j = 0; j < sc.n_tri; ++j) {
        double t;
        int idx = sc.tri_prim[j];
        if (tri_t(r, sc.tri + 12 * j, t_min, h.t, idx, h.prim, &t)) { h.t = t; h.prim = idx; }
    }
    if (saw_nan) h = closest_hit_listorder(sc, r, t_min, t_max);
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
        outward = (rec.p - v3(s.x, s.y, s.z)) / s.w;
    } else if (k == 1) {
        int axis = sc.rect_axis[j];
        outward = axis == 0 ? v3(0, 0, 1) : (axis == 1 ? v3(0, 1, 0) : v3(1, 0, 0));
    } else {
        const float* q = sc.tri + 12 * j;
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
        double t;
        float4 s = make_float4(sc.light_c[0], sc.light_c[1], sc.light_c[2], sc.light_r);
        if (!sphere_t(r, s, 0.001, kInf, 0, -1, &t)) return 0.0;
        V3 co = v3(s.x, s.y, s.z) - o;
        double cos_theta_max = __builtin_sqrt(double(1 - s.w * s.w / length_squared(co)));
        double solid_angle = 2 * kPi * (1 - cos_theta_max);
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
// bit-identical per-sample colours, not just "within tolerance".
// ------------------------------------------------------------------------------------------------
struct PathStack {
    float4* __restrict__ base;   // [depth][stride] entries (w.xyz, pdf)
    uint32_t stride, tid;
    RT_DEV void push(int level, V3 w, float pdf) { base[size_t(level) * stride + tid] = make_float4(w.x, w.y, w.z, pdf); }
    RT_DEV float4 get(int level) const { return base[size_t(level) * stride + tid]; }
};

__device__ __forceinline__ V3 trace_path(const DevScene& sc, Ray ray, Rng& g, int depth, bool normals, PathStack& st,
                                         uint32_t& n_rays) {
    int level = 0;
    uint64_t spec_lo = 0, spec_hi = 0;   // bit k set: level k is a specular bounce (depth <= 128)
    V3 L = v3(0.0f, 0.0f, 0.0f);
    for (int depth_left = depth;; --depth_left) {
        if (depth_left <= 0) { L = v3(0.0f, 0.0f, 0.0f); break; }           // calc.cpp:96-97
        ++n_rays;
        Hit h = closest_hit(sc, ray, double(0.001f), kInf);                  // calc.cpp:99
        if (h.prim < 0) { L = v3(sc.bg[0], sc.bg[1], sc.bg[2]); break; }      // calc.cpp:100-103
        if (normals) {                                                       // calc.cpp:105-109
            V3 N = normalized(ray_at(ray, h.t) - v3(0.0f, 0.0f, -1.0f));
            L = 0.5f * v3(N.x + 1.0f, N.y + 1.0f, N.z + 1.0f);
            break;
        }
        HitRec rec = hit_record(sc, ray, h);
        int m = sc.prim_mat[h.prim];
        int mk = sc.mat_kind[m];
        V3 rgb = v3(sc.mat_rgb[3 * m], sc.mat_rgb[3 * m + 1], sc.mat_rgb[3 * m + 2]);
        if (mk == RT_MAT_DIFFUSE_LIGHT) { L = rgb; break; }                   // hitPosition.h:301-310, calc.cpp:115-116
        bool specular;
        V3 w;
        float pdf = 1.0f;
        if (mk == RT_MAT_METAL) {                                            // hitPosition.h:230-239
            V3 reflected = reflect(normalized(ray.d), rec.normal);
            float fuzz = sc.mat_fuzz[m];
            ray = make_ray(rec.p, reflected + fuzz * random_in_unit_sphere(g));
            specular = true; w = rgb;
        } else if (mk == RT_MAT_DIELECTRIC) {                                // hitPosition.h:253-287
            double ir = sc.mat_ir[m];
            double refraction_ratio = rec.front_face ? (1.0 / ir) : ir;
            V3 unit_direction = normalized(ray.d);
            float cos_theta = float(__builtin_fmin(double(dot(-unit_direction, rec.normal)), 1.0));
            float sin_theta = float(__builtin_sqrt(double(1.0f - cos_theta * cos_theta)));
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
            specular = true; w = v3(1.0f, 1.0f, 1.0f);
        } else {                                                             // lambertian, hitPosition.h:202-216
            Onb uvw;
            uvw.build_from_w(rec.normal);                                    // cosine_pdf ctor, hitPosition.h:143
            V3 dir;
            if (g.u01() < 0.5f) dir = light_random(sc, g, rec.p);            // mixture_pdf::generate, hitPosition.cpp:67-72
            else dir = uvw.local(random_cosine_direction(g));
            Ray scattered = make_ray(rec.p, dir);
            V3 ndir = normalized(dir);
            float cosine_s = dot(rec.normal, ndir);                          // lambertian::scattering_pdf
            double spdf = cosine_s < 0 ? 0.0 : double(cosine_s) / kPi;
            float cosine_c = dot(ndir, uvw.w);                               // cosine_pdf::value
            double cval = (cosine_c <= 0) ? 0.0 : double(cosine_c) / kPi;
            double pval = 0.5 * light_pdf_value(sc, rec.p, dir) + 0.5 * cval;   // mixture_pdf::value
            ray = scattered;
            specular = false; w = rgb * float(spdf); pdf = float(pval);
        }
        st.push(level, w, pdf);
        if (specular) { if (level < 64) spec_lo |= 1ull << level; else spec_hi |= 1ull << (level - 64); }
        ++level;
    }
    while (level > 0) {                                                      // unwind, innermost first
        --level;
        float4 e = st.get(level);
        bool specular = level < 64 ? ((spec_lo >> level) & 1) : ((spec_hi >> (level - 64)) & 1);
        V3 w = v3(e.x, e.y, e.z);
        if (specular) L = w * L;                                              // calc.cpp:119-120
        else L = v3(0.0f, 0.0f, 0.0f) + w * L / e.w;                          // calc.cpp:129-130 (emitted == 0)
    }
    return L;
}

// ================================================================================================
// Kernels
// ================================================================================================
constexpr int kBlock = 256;   // 4 waves; each wave covers an 8x8 pixel tile of a 16x16 block tile
constexpr int kTile = 16;

__device__ __forceinline__ void tile_xy(int tile, int tiles_x, int y_begin, int& x, int& y) {
    int t = threadIdx.x;
    int wave = t >> 6, lane = t & 63;
    int lx = (lane & 7) + 8 * (wave & 1);
    int ly = (lane >> 3) + 8 * (wave >> 1);
    x = (tile % tiles_x) * kTile + lx;
    y = y_begin + (tile / tiles_x) * kTile + ly;
}

__global__ __launch_bounds__(kBlock) void render_kernel(DevScene sc, DevCamera cam, int W, int H, int y_begin, int y_count,
                                                        int spp_begin, int spp_count, int depth, uint64_t seed, uint32_t flags,
                                                        float4* __restrict__ accum, float4* __restrict__ stack,
                                                        unsigned long long* __restrict__ rays_out) {
    const int tiles_x = (W + kTile - 1) / kTile, tiles_y = (y_count + kTile - 1) / kTile;
    const int n_tiles = tiles_x * tiles_y;
    PathStack st; st.base = stack; st.stride = gridDim.x * kBlock; st.tid = blockIdx.x * kBlock + threadIdx.x;
    const bool normals = (flags & RT_RENDER_NORMALS) != 0;
    uint32_t n_rays = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int x, y;
        tile_xy(tile, tiles_x, y_begin, x, y);
        if (x >= W || y >= y_begin + y_count || y >= H) continue;
        const uint32_t pix = uint32_t(y) * uint32_t(W) + uint32_t(x);
        V3 sum = v3(0.0f, 0.0f, 0.0f);
        for (int s = spp_begin; s < spp_begin + spp_count; ++s) {
            Rng g; g.key(seed, pix, uint32_t(s));
            float u = (float(x) + g.u01()) / float(W);                       // rayview.cpp:80-81
            float v = (float(y) + g.u01()) / float(H);
            Ray ray = get_ray(cam, g, u, v);
            V3 c = trace_path(sc, ray, g, depth, normals, st, n_rays);        // rayview.cpp:83
            sum = sum + c;
        }
        float4 a = make_float4(sum.x, sum.y, sum.z, float(spp_count));
        if (flags & RT_RENDER_ACCUMULATE) {
            float4 o = accum[pix];
            a = make_float4(o.x + sum.x, o.y + sum.y, o.z + sum.z, o.w + float(spp_count));
        }
        accum[pix] = a;
    }
    if (rays_out) {
        // per-wave sum, one atomic per wave
        unsigned long long v = n_rays;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(rays_out, v);
    }
}

__global__ __launch_bounds__(kBlock) void primary_kernel(DevScene sc, DevCamera cam, int W, int H, uint64_t seed,
                                                         int* __restrict__ prim_out, double* __restrict__ t_out) {
    const int tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile;
    for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
        int x, y;
        tile_xy(tile, tiles_x, 0, x, y);
        if (x >= W || y >= H) continue;
        const uint32_t pix = uint32_t(y) * uint32_t(W) + uint32_t(x);
        Rng g; g.key(seed, pix, 0u);
        float u = (float(x) + 0.5f) / float(W), v = (float(y) + 0.5f) / float(H);
        Ray ray = get_ray(cam, g, u, v);
        Hit h = closest_hit(sc, ray, double(0.001f), kInf);
        if (prim_out) prim_out[pix] = h.prim;
        if (t_out) t_out[pix] = h.prim >= 0 ? h.t : 0.0;
    }
}

__global__ __launch_bounds__(kBlock) void hit_rays_kernel(DevScene sc, long long n, const float* __restrict__ ro,
                                                          const float* __restrict__ rd, double t_min, double t_max,
                                                          int* __restrict__ prim_out, double* __restrict__ t_out,
                                                          float* __restrict__ p_out, float* __restrict__ n_out,
                                                          uint8_t* __restrict__ ff_out) {
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        Ray ray = make_ray(v3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), v3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]));
        Hit h = closest_hit(sc, ray, t_min, t_max);
        prim_out[i] = h.prim;
        if (t_out) t_out[i] = h.prim >= 0 ? h.t : 0.0;
        HitRec rec; rec.p = v3(0, 0, 0); rec.normal = v3(0, 0, 0); rec.front_face = false;
        if (h.prim >= 0) rec = hit_record(sc, ray, h);
        if (p_out) { p_out[3 * i] = rec.p.x; p_out[3 * i + 1] = rec.p.y; p_out[3 * i + 2] = rec.p.z; }
        if (n_out) { n_out[3 * i] = rec.normal.x; n_out[3 * i + 1] = rec.normal.y; n_out[3 * i + 2] = rec.normal.z; }
        if (ff_out) ff_out[i] = rec.front_face ? 1 : 0;
    }
}

__global__ __launch_bounds__(kBlock) void trace_rays_kernel(DevScene sc, long long n, const float* __restrict__ ro,
                                                            const float* __restrict__ rd, int depth, uint64_t seed, uint32_t flags,
                                                            float* __restrict__ rgb, float4* __restrict__ stack) {
    PathStack st; st.base = stack; st.stride = gridDim.x * kBlock; st.tid = blockIdx.x * kBlock + threadIdx.x;
    uint32_t n_rays = 0;
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        Ray ray = make_ray(v3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), v3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]));
        Rng g; g.key(seed, uint32_t(i), 0u);
        V3 c = trace_path(sc, ray, g, depth, (flags & RT_RENDER_NORMALS) != 0, st, n_rays);
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
namespace {

thread_local std::string g_err;

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
} g;

int need_ready() { return g.ready ? 0 : fail("rt_init() has not bound a HIP device (no CPU fallback exists)"); }

template <class T, class P>
int upload(const std::vector<T>& v, P* out) {
    *out = nullptr;
    size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(T);
    void* d = nullptr;
    HIP_OK(hipMalloc(&d, bytes));
    g.scene_bufs.push_back(d);
    if (!v.empty()) HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(d);
    return 0;
}

void free_scene() {
    for (void* p : g.scene_bufs) (void)hipFree(p);
    g.scene_bufs.clear();
    g.have_scene = false;
}

int grid_for(int n_tiles) {
    int resident = g.prop.multiProcessorCount * 8;   // 8 blocks of 256 threads per CU = 32 waves/CU
    return n_tiles < resident ? (n_tiles > 0 ? n_tiles : 1) : resident;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

int rt_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail("no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail("device %d out of range (0..%d)", device, n - 1);
    if (g.ready && g.device == device) return 0;
    if (g.ready) rt_shutdown();
    HIP_OK(hipSetDevice(device));
    HIP_OK(hipGetDeviceProperties(&g.prop, device));
    HIP_OK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_OK(hipEventCreate(&g.ev0));
    HIP_OK(hipEventCreate(&g.ev1));
    g.device = device;
    g.ready = true;
    return 0;
}

void rt_shutdown(void) {
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    free_scene();
    if (g.stack) (void)hipFree(g.stack);
    g.stack = nullptr; g.stack_bytes = 0;
    (void)hipEventDestroy(g.ev0); (void)hipEventDestroy(g.ev1);
    (void)hipStreamDestroy(g.stream);
    g.ready = false; g.have_cam = false; g.device = -1;
}

int rt_device_info(char* buf, int buflen) {
    if (need_ready()) return 1;
    snprintf(buf, buflen, "%s|%s|%d|%d", g.prop.name, g.prop.gcnArchName, g.prop.multiProcessorCount, g.prop.clockRate / 1000);
    return 0;
}

int rt_malloc(void** d_ptr, uint64_t bytes) { if (need_ready()) return 1; HIP_OK(hipMalloc(d_ptr, bytes ? bytes : 1)); return 0; }
int rt_free(void* d_ptr) { if (need_ready()) return 1; HIP_OK(hipFree(d_ptr)); return 0; }
int rt_memcpy_d2h(void* dst, const void* d_src, uint64_t bytes) {
    if (need_ready()) return 1;
    HIP_OK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, g.stream));
    HIP_OK(hipStreamSynchronize(g.stream));
    return 0;
}
int rt_memcpy_h2d(void* d_dst, const void* src, uint64_t bytes) {
    if (need_ready()) return 1;
    HIP_OK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, g.stream));
    HIP_OK(hipStreamSynchronize(g.stream));
    return 0;
}
int rt_memset(void* d_ptr, int value, uint64_t bytes) {
    if (need_ready()) return 1;
    HIP_OK(hipMemsetAsync(d_ptr, value, bytes, g.stream));
    HIP_OK(hipStreamSynchronize(g.stream));
    return 0;
}
int rt_sync(void) { if (need_ready()) return 1; HIP_OK(hipStreamSynchronize(g.stream)); return 0; }

int rt_upload_scene(const RtScene* s) {
    if (need_ready()) return 1;
    if (!s) return fail("rt_upload_scene: null scene");
    if (s->n_prims < 0 || s->n_mats <= 0) return fail("rt_upload_scene: n_prims=%d n_mats=%d", s->n_prims, s->n_mats);
    if (s->n_prims > 0 && (!s->prim_kind || !s->prim_mat || !s->prim_data)) return fail("rt_upload_scene: null primitive arrays");
    if (!s->mat_kind || !s->mat_rgb || !s->mat_fuzz || !s->mat_ir) return fail("rt_upload_scene: null material arrays");
    std::vector<float4> sph; std::vector<int> sph_prim;
    std::vector<double> rect; std::vector<int> rect_axis, rect_prim;
    std::vector<float> tri; std::vector<int> tri_prim;
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
            const float q[12] = {v0.x, v0.y, v0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, nn.x, nn.y, nn.z};
            tri.insert(tri.end(), q, q + 12); tri_prim.push_back(i);
            break; }
        default: return fail("rt_upload_scene: primitive %d has unknown kind %d", i, s->prim_kind[i]);
        }
    }
    for (int i = 0; i < s->n_mats; ++i)
        if (s->mat_kind[i] < RT_MAT_LAMBERTIAN || s->mat_kind[i] > RT_MAT_DIFFUSE_LIGHT)
            return fail("rt_upload_scene: material %d has unknown kind %d", i, s->mat_kind[i]);
    if (s->light_kind != RT_LIGHT_BASE && s->light_kind != RT_PRIM_SPHERE && s->light_kind != RT_PRIM_XY_RECT &&
        s->light_kind != RT_PRIM_XZ_RECT)
        return fail("rt_upload_scene: light kind %d has no pdf_value/random in the reference; pass RT_LIGHT_BASE", s->light_kind);

    HIP_OK(hipStreamSynchronize(g.stream));
    free_scene();
    DevScene ds{};
    ds.n_sph = int(sph.size()); ds.n_rect = int(rect_axis.size()); ds.n_tri = int(tri_prim.size()); ds.n_prims = s->n_prims;
    if (upload(sph, &ds.sph) || upload(sph_prim, &ds.sph_prim) || upload(rect, &ds.rect) || upload(rect_axis, &ds.rect_axis) ||
        upload(rect_prim, &ds.rect_prim) || upload(tri, &ds.tri) || upload(tri_prim, &ds.tri_prim) ||
        upload(prim_kind, &ds.prim_kind) || upload(prim_local, &ds.prim_local) || upload(prim_mat, &ds.prim_mat))
        return 1;
    std::vector<int> mk(s->mat_kind, s->mat_kind + s->n_mats);
    std::vector<float> mrgb(s->mat_rgb, s->mat_rgb + 3 * size_t(s->n_mats)), mf(s->mat_fuzz, s->mat_fuzz + s->n_mats);
    std::vector<double> mir(s->mat_ir, s->mat_ir + s->n_mats);
    if (upload(mk, &ds.mat_kind) || upload(mrgb, &ds.mat_rgb) || upload(mf, &ds.mat_fuzz) || upload(mir, &ds.mat_ir)) return 1;
    ds.light_kind = s->light_kind;
    for (int k = 0; k < 3; ++k) ds.light_c[k] = float(s->light_data[k]);
    ds.light_r = float(s->light_data[3]);
    for (int k = 0; k < 5; ++k) ds.light_rect[k] = s->light_data[k];
    for (int k = 0; k < 3; ++k) ds.bg[k] = s->background[k];
    g.scene = ds;
    g.have_scene = true;
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
    g.cam.origin = get(c->origin); g.cam.llc = get(c->lower_left_corner); g.cam.horizontal = get(c->horizontal);
    g.cam.vertical = get(c->vertical); g.cam.u = get(c->u); g.cam.v = get(c->v); g.cam.w = get(c->w);
    g.cam.lens_radius = c->lens_radius;
    g.have_cam = true;
    return 0;
}

static int need_scene_cam() {
    if (need_ready()) return 1;
    if (!g.have_scene) return fail("no scene uploaded (rt_upload_scene)");
    if (!g.have_cam) return fail("no camera set (rt_set_camera)");
    return 0;
}

static int finish_timed(int launches) {
    HIP_OK(hipEventRecord(g.ev1, g.stream));
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    HIP_OK(hipEventElapsedTime(&g.last_ms, g.ev0, g.ev1));
    g.last_launches = launches;
    return 0;
}

int rt_primary_hits(int w, int h, uint64_t seed, int32_t* d_prim, double* d_t) {
    if (need_scene_cam()) return 1;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return fail("rt_primary_hits: bad size %dx%d", w, h);
    int n_tiles = ((w + kTile - 1) / kTile) * ((h + kTile - 1) / kTile);
    HIP_OK(hipEventRecord(g.ev0, g.stream));
    hipLaunchKernelGGL(primary_kernel, dim3(grid_for(n_tiles)), dim3(kBlock), 0, g.stream, g.scene, g.cam, w, h, seed, d_prim, d_t);
    return finish_timed(1);
}

int rt_hit_rays(int64_t n, const float* d_origin, const float* d_dir, double t_min, double t_max, int32_t* d_prim, double* d_t,
                float* d_p, float* d_normal, uint8_t* d_front) {
    if (need_ready()) return 1;
    if (!g.have_scene) return fail("no scene uploaded (rt_upload_scene)");
    if (n < 0 || (n > 0 && (!d_origin || !d_dir || !d_prim))) return fail("rt_hit_rays: bad arguments");
    if (n == 0) { g.last_ms = 0; g.last_launches = 0; return 0; }
    int blocks = int((n + kBlock - 1) / kBlock);
    HIP_OK(hipEventRecord(g.ev0, g.stream));
    hipLaunchKernelGGL(hit_rays_kernel, dim3(grid_for(blocks)), dim3(kBlock), 0, g.stream, g.scene, (long long)n, d_origin, d_dir,
                       t_min, t_max, d_prim, d_t, d_p, d_normal, d_front);
    return finish_timed(1);
}

static int ensure_stack(int depth, int grid) {
    size_t need = size_t(depth > 0 ? depth : 1) * size_t(grid) * kBlock * sizeof(float4);
    if (need > g.stack_bytes) {
        HIP_OK(hipStreamSynchronize(g.stream));
        if (g.stack) HIP_OK(hipFree(g.stack));
        g.stack = nullptr; g.stack_bytes = 0;
        HIP_OK(hipMalloc(&g.stack, need));
        g.stack_bytes = need;
    }
    return 0;
}

int rt_trace_rays(int64_t n, const float* d_origin, const float* d_dir, int depth, uint64_t seed, uint32_t flags, float* d_rgb) {
    if (need_ready()) return 1;
    if (!g.have_scene) return fail("no scene uploaded (rt_upload_scene)");
    if (n < 0 || (n > 0 && (!d_origin || !d_dir || !d_rgb))) return fail("rt_trace_rays: bad arguments");
    if (depth < 0 || depth > 128) return fail("rt_trace_rays: depth %d outside 0..128", depth);
    if (n == 0) { g.last_ms = 0; g.last_launches = 0; return 0; }
    int grid = grid_for(int((n + kBlock - 1) / kBlock));
    if (ensure_stack(depth, grid)) return 1;
    HIP_OK(hipEventRecord(g.ev0, g.stream));
    hipLaunchKernelGGL(trace_rays_kernel, dim3(grid), dim3(kBlock), 0, g.stream, g.scene, (long long)n, d_origin, d_dir, depth, seed,
                       flags, d_rgb, static_cast<float4*>(g.stack));
    return finish_timed(1);
}

int rt_render(int w, int h, int y_begin, int y_count, int spp_begin, int spp_count, int depth, uint64_t seed, uint32_t flags,
              float* d_accum, uint64_t* d_rays) {
    if (need_scene_cam()) return 1;
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return fail("rt_render: bad size %dx%d", w, h);
    if (y_begin < 0 || y_count < 0 || y_begin + y_count > h) return fail("rt_render: rows [%d,+%d) outside 0..%d", y_begin, y_count, h);
    if (spp_begin < 0 || spp_count < 0) return fail("rt_render: bad sample range [%d,+%d)", spp_begin, spp_count);
    if (depth < 0 || depth > 128) return fail("rt_render: depth %d outside 0..128", depth);
    if (!d_accum) return fail("rt_render: null accumulator");
    if (y_count == 0) { g.last_ms = 0; g.last_launches = 0; return 0; }
    int n_tiles = ((w + kTile - 1) / kTile) * ((y_count + kTile - 1) / kTile);
    int grid = grid_for(n_tiles);
    if (ensure_stack(depth, grid)) return 1;
    HIP_OK(hipEventRecord(g.ev0, g.stream));
    hipLaunchKernelGGL(render_kernel, dim3(grid), dim3(kBlock), 0, g.stream, g.scene, g.cam, w, h, y_begin, y_count, spp_begin,
                       spp_count, depth, seed, flags, reinterpret_cast<float4*>(d_accum), static_cast<float4*>(g.stack),
                       reinterpret_cast<unsigned long long*>(d_rays));
    return finish_timed(1);
}

int rt_tonemap(int w, int h, const float* d_accum, int n_samples, uint32_t* d_rgb32) {
    if (need_ready()) return 1;
    if (w <= 0 || h <= 0 || !d_accum || !d_rgb32) return fail("rt_tonemap: bad arguments");
    int n = w * h;
    hipLaunchKernelGGL(tonemap_kernel, dim3((n + 255) / 256), dim3(256), 0, g.stream, w, h, reinterpret_cast<const float4*>(d_accum),
                       n_samples, d_rgb32);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    return 0;
}

int rt_last_kernel_ms(float* ms, int* launches) {
    if (ms) *ms = g.last_ms;
    if (launches) *launches = g.last_launches;
    return 0;
}

}  // extern "C"
