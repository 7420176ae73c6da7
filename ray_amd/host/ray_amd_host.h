// The device-backed half of the C++ host mirror: everything that needs the HIP library (include/ray_amd.h).
//   * rt_host::Session            one context of the library; flattens std::vector<std::shared_ptr<shape>> -> RtScene
//   * calc::ray_color(...)        the reference's six-argument signature (calc.h:89-94)      -> rt_trace_rays
//   * hitFromList(list, ray, ..)  the reference's four-argument signature (hitPosition.h:187) -> rt_hit_rays
//   * shape::hit                  one shape as a one-entry list
//   * renderAll(...)              the function rayview.h:32 declares and the reference never defines -> rt_render
//   * ProgressiveView             RayView::go for the GPU
// The five headers with the reference's names (Ray.h, calc.h, hitPosition.h, shapes.h, camera.h) each end by
// including this file, so a translation unit written against the reference's headers compiles unchanged and finds
// the definitions. No CPU fallback: without a HIP device every one of these throws rt_host::error.
#ifndef RAY_AMD_HOST_H
#define RAY_AMD_HOST_H

#ifndef RAY_AMD_MIRROR_OUTER
#define RAY_AMD_MIRROR_OUTER
#define RAY_AMD_MIRROR_OUTER_IS_HOST
#endif
#include "Ray.h"
#include "calc.h"
#include "hitPosition.h"
#include "shapes.h"
#include "camera.h"
#ifdef RAY_AMD_MIRROR_OUTER_IS_HOST
#undef RAY_AMD_MIRROR_OUTER_IS_HOST
#undef RAY_AMD_MIRROR_OUTER
#endif

#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace rt_host {

// Flattens worldObjects (+ lights, background) into the arrays of RtScene and uploads them; keeps the flattened
// list so hit records can point back at the caller's material objects. upload() returns without touching the device
// when the flattened content is what the device already holds (content hash), so per-ray callers pay one
// flattening pass per call, not one upload.
class Session {
public:
    explicit Session(int device = 0) : device_(device) { check(rt_init(device)); }
    void bind() const { check(rt_set_device(device_)); }
    bool upload(const std::vector<std::shared_ptr<shape>>& world, const std::shared_ptr<shape>& lights, const QVector3D& background) {
        std::vector<flat_prim> fp;
        std::vector<int32_t> kind, mat, outer, side;
        std::vector<double> data;
        std::vector<std::shared_ptr<material>> ms;
        for (size_t i = 0; i < world.size(); ++i) {
            const size_t before = fp.size();
            world[i]->rt_flatten(fp);
            for (size_t j = before; j < fp.size(); ++j) {
                const flat_prim& p = fp[j];
                if (!p.mat) throw error("world shape without a material");
                int m = -1;
                for (size_t k = 0; k < ms.size(); ++k) if (ms[k] == p.mat) m = int(k);
                if (m < 0) { ms.push_back(p.mat); m = int(ms.size()) - 1; }
                kind.push_back(p.kind); mat.push_back(m); outer.push_back(int(i)); side.push_back(p.side);
                data.insert(data.end(), p.d, p.d + RT_PRIM_STRIDE);
            }
        }
        if (ms.empty()) ms.push_back(std::make_shared<lambertian>(QVector3D(0, 0, 0)));
        std::vector<int32_t> mk; std::vector<float> rgb, fz; std::vector<double> ir;
        for (auto& m : ms) {
            mk.push_back(m->rt_kind()); QVector3D c = m->rt_rgb();
            rgb.insert(rgb.end(), {c.x(), c.y(), c.z()}); fz.push_back(m->rt_fuzz()); ir.push_back(m->rt_ir());
        }
        RtScene s{};
        s.n_prims = int(kind.size());
        s.prim_kind = kind.data(); s.prim_mat = mat.data(); s.prim_outer = outer.data(); s.prim_side = side.data();
        s.prim_data = data.data();
        s.n_mats = int(mk.size()); s.mat_kind = mk.data(); s.mat_rgb = rgb.data(); s.mat_fuzz = fz.data(); s.mat_ir = ir.data();
        s.light_kind = RT_LIGHT_BASE;
        if (lights) {
            std::vector<flat_prim> l; lights->rt_flatten(l);
            if (l.size() == 1 && (l[0].kind == RT_PRIM_SPHERE || l[0].kind == RT_PRIM_XY_RECT || l[0].kind == RT_PRIM_XZ_RECT)) {
                s.light_kind = l[0].kind;
                for (int k = 0; k < RT_PRIM_STRIDE; ++k) s.light_data[k] = l[0].d[k];
            }   // box / yz_rect / medium lights use the base class pdf (shapes.h:56-62)
        }
        s.background[0] = background.x(); s.background[1] = background.y(); s.background[2] = background.z();
        uint64_t h = 1469598103934665603ull;   // FNV-1a over everything the device would receive
        auto mix = [&h](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
        mix(kind.data(), kind.size() * 4); mix(mat.data(), mat.size() * 4); mix(data.data(), data.size() * 8);
        mix(mk.data(), mk.size() * 4); mix(rgb.data(), rgb.size() * 4); mix(fz.data(), fz.size() * 4); mix(ir.data(), ir.size() * 8);
        mix(&s.light_kind, 4); mix(s.light_data, sizeof s.light_data); mix(s.background, sizeof s.background);
        bind();
        const bool fresh = !have_ || h != hash_ || kind.size() != prims.size();
        if (fresh) { check(rt_upload_scene(&s)); hash_ = h; have_ = true; ++uploads; }
        prims.swap(fp); mats.swap(ms); outer_of.swap(outer);
        return fresh;
    }
    std::vector<flat_prim> prims;
    std::vector<std::shared_ptr<material>> mats;
    std::vector<int32_t> outer_of;
    int uploads = 0;              // how often the device copy was actually replaced
private:
    int device_;
    uint64_t hash_ = 0;
    bool have_ = false;
};

// The session behind the reference-signature entry points (device: $RAY_AMD_DEVICE, default 0).
inline Session& default_session() {
    static Session s(std::getenv("RAY_AMD_DEVICE") ? std::atoi(std::getenv("RAY_AMD_DEVICE")) : 0);
    return s;
}
// calc::ray_color's random numbers: call k of the process draws from the counter-RNG stream (seed + k, 0, 0)
// (the reference advances three global engines instead; DESIGN.md "RNG contract"). Settable for reproducible runs.
inline uint64_t& ray_color_stream() { static uint64_t k = 0; return k; }

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(uint64_t bytes) { check(rt_malloc(&p, bytes)); }
    ~DevBuf() { if (p) rt_free(p); }
    DevBuf(const DevBuf&) = delete;
};

inline std::optional<hit_record> hit_uploaded(Session& s, const Ray& ray, double t_min, double t_max) {
    const float o[3] = {ray.pointOfOrigin.x(), ray.pointOfOrigin.y(), ray.pointOfOrigin.z()};
    const float d[3] = {ray.direction.x(), ray.direction.y(), ray.direction.z()};
    DevBuf bo(12), bd(12), bprim(4), bt(8), bp(12), bn(12), bf(1);
    check(rt_memcpy_h2d(bo.p, o, 12)); check(rt_memcpy_h2d(bd.p, d, 12));
    check(rt_hit_rays(1, (const float*)bo.p, (const float*)bd.p, t_min, t_max, (int32_t*)bprim.p, (double*)bt.p, (float*)bp.p, (float*)bn.p,
                      (uint8_t*)bf.p));
    int32_t prim; double t; float p[3], n[3]; uint8_t ff;
    check(rt_memcpy_d2h(&prim, bprim.p, 4)); check(rt_memcpy_d2h(&t, bt.p, 8));
    check(rt_memcpy_d2h(p, bp.p, 12)); check(rt_memcpy_d2h(n, bn.p, 12)); check(rt_memcpy_d2h(&ff, bf.p, 1));
    if (prim < 0) return std::nullopt;
    hit_record rec;
    rec.p = QVector3D(p[0], p[1], p[2]); rec.normal = QVector3D(n[0], n[1], n[2]); rec.t = t; rec.front_face = ff != 0;
    rec.rt_prim = prim; rec.mat_ptr = s.prims[size_t(prim)].mat;
    return rec;
}

inline QVector3D trace_uploaded(const Ray& r, int depth, bool drawOnlyColors, uint64_t seed) {
    const float o[3] = {r.pointOfOrigin.x(), r.pointOfOrigin.y(), r.pointOfOrigin.z()};
    const float d[3] = {r.direction.x(), r.direction.y(), r.direction.z()};
    DevBuf bo(12), bd(12), bc(12);
    check(rt_memcpy_h2d(bo.p, o, 12)); check(rt_memcpy_h2d(bd.p, d, 12));
    check(rt_trace_rays(1, (const float*)bo.p, (const float*)bd.p, depth, seed, drawOnlyColors ? RT_RENDER_NORMALS : 0u, (float*)bc.p));
    float c[3];
    check(rt_memcpy_d2h(c, bc.p, 12));
    return QVector3D(c[0], c[1], c[2]);
}

}  // namespace rt_host

// ---- the reference's signatures -----------------------------------------------------------------------------------
// hitFromList (hitPosition.h:187-191): one ray against the list, on the GPU. The list is uploaded when its content
// differs from what the device holds (lights / background of the previous upload are replaced by none / default).
inline std::optional<hit_record> hitFromList(const std::vector<std::shared_ptr<shape>>& sphereList, const Ray& ray, double t_min, double t_max) {
    rt_host::Session& s = rt_host::default_session();
    s.upload(sphereList, nullptr, img::gradientBgVec);
    return rt_host::hit_uploaded(s, ray, t_min, t_max);
}
// the same against an explicitly managed session whose scene is already uploaded
inline std::optional<hit_record> hitFromList(rt_host::Session& s, const Ray& ray, double t_min, double t_max) {
    s.bind();
    return rt_host::hit_uploaded(s, ray, t_min, t_max);
}

// shape::hit (shapes.h:53): this shape as a one-entry list
inline std::optional<hit_record> shape::hit(const Ray& r, double t_min, double t_max) const {
    std::vector<std::shared_ptr<shape>> one;
    one.emplace_back(std::shared_ptr<shape>(), const_cast<shape*>(this));   // aliasing constructor: refers to *this, owns nothing
    return hitFromList(one, r, t_min, t_max);
}

namespace calc {
// calc::ray_color (calc.h:89-94, calc.cpp:90-131), the reference's six arguments, on the GPU.
inline QVector3D ray_color(const Ray& r, const QVector3D& background, const std::vector<std::shared_ptr<shape>>& worldObjects,
                           std::shared_ptr<shape>& lights, int depth, bool drawOnlyColors) {
    rt_host::Session& s = rt_host::default_session();
    s.upload(worldObjects, lights, background);
    return rt_host::trace_uploaded(r, depth, drawOnlyColors, rt_host::ray_color_stream()++);
}
// against an explicitly managed session whose scene is already uploaded; `seed` selects the counter-RNG stream
inline QVector3D ray_color(rt_host::Session& s, const Ray& r, int depth, bool drawOnlyColors = false, uint64_t seed = 0) {
    s.bind();
    return rt_host::trace_uploaded(r, depth, drawOnlyColors, seed);
}
}  // namespace calc

// RayView::renderAll (declared rayview.h:32, never defined in the reference) with the per-pixel contract of
// renderOneRay + writeToImg (rayview.cpp:67-87): returns the Format_RGB32 image (w*h uint32, mirrored on both axes
// as writeToImg does) and optionally the linear accumulators.
inline std::vector<uint32_t> renderAll(rt_host::Session& s, int width, int height, int samples, const camera& cam,
                                       const std::vector<std::shared_ptr<shape>>& worldObjects, int max_depth,
                                       const std::shared_ptr<shape>& lights = nullptr,
                                       const QVector3D& background = img::gradientBgVec, uint64_t seed = 0,
                                       std::vector<float>* accum_out = nullptr, uint64_t* rays_out = nullptr) {
    s.upload(worldObjects, lights, background);
    rt_host::check(rt_set_camera(&cam.rt()));
    const uint64_t n = uint64_t(width) * uint64_t(height);
    rt_host::DevBuf acc(n * 16), img32(n * 4), rays(8);
    rt_host::check(rt_memset(rays.p, 0, 8));
    rt_host::check(rt_render(width, height, 0, height, 0, samples, max_depth, seed, 0, (float*)acc.p, (uint64_t*)rays.p));
    rt_host::check(rt_tonemap(width, height, (const float*)acc.p, samples, (uint32_t*)img32.p));
    std::vector<uint32_t> out(n);
    rt_host::check(rt_memcpy_d2h(out.data(), img32.p, n * 4));
    if (accum_out) { accum_out->resize(n * 4); rt_host::check(rt_memcpy_d2h(accum_out->data(), acc.p, n * 16)); }
    if (rays_out) rt_host::check(rt_memcpy_d2h(rays_out, rays.p, 8));
    return out;
}
// the signature exactly as rayview.h:32 declares it (RayView's other state -- lights, background -- as trailing defaults)
inline std::vector<uint32_t> renderAll(int width, int height, int samples, const camera& cam,
                                       const std::vector<std::shared_ptr<shape>>& worldObjects, int max_depth,
                                       const std::shared_ptr<shape>& lights = nullptr, const QVector3D& background = img::gradientBgVec,
                                       uint64_t seed = 0) {
    return renderAll(rt_host::default_session(), width, height, samples, cam, worldObjects, max_depth, lights, background, seed);
}

// RayView::go (rayview.cpp:98-117) for the GPU. The reference spends each 16 ms slice on random
// pixels, every one rendered with all its samples and overwritten on the next visit; here a slice is
// one or more FULL-FRAME passes of a few samples each, continued in sample order (RT_RENDER_CONTINUE), so the
// picture refines everywhere at once. Sample indices continue across slices (pass k renders samples
// [done, done + n) of every pixel), so N slices of the same scene give exactly the samples of one
// N-slice-long render. A scene or camera change (the sphere-Y slider, WASD: rayview.cpp:130-184) is
// reset(): re-upload and start again from sample 0.
// renderPixels() and go() are two schedules over ONE accumulator: call reset() when switching from one to the other
// (after renderPixels the per-pixel sample sets differ, and go() would add samples [done, ..) on top of them).
class ProgressiveView {
public:
    ProgressiveView(rt_host::Session& s, int width, int height, int max_depth, uint64_t seed = 0)
        : s_(s), w_(width), h_(height), depth_(max_depth), seed_(seed), acc_(uint64_t(width) * height * 16), img_(uint64_t(width) * height * 4) {
        rt_host::check(rt_memset(acc_.p, 0, uint64_t(w_) * h_ * 16));   // pixels never listed by renderPixels() read as "no samples"
    }
    void reset(const std::vector<std::shared_ptr<shape>>& world, const std::shared_ptr<shape>& lights, const camera& cam,
               const QVector3D& background = img::gradientBgVec) {
        s_.upload(world, lights, background);
        rt_host::check(rt_set_camera(&cam.rt()));
        rt_host::check(rt_memset(acc_.p, 0, uint64_t(w_) * h_ * 16));
        done_ = 0;
    }
    // RayView::go's own schedule (rayview.cpp:98-107): `pixels` holds the (randX, randY) draws of one tick as
    // randY * width + randX; each listed pixel gets `samples` samples and its accumulator is overwritten, as
    // renderOneRay + writeToImg do (rayview.cpp:67-87). `pass` selects the RNG streams of this tick.
    void renderPixels(const std::vector<int32_t>& pixels, int samples, uint64_t pass = 0) {
        if (pixels.empty()) return;
        s_.bind();
        rt_host::DevBuf list(pixels.size() * 4);
        rt_host::check(rt_memcpy_h2d(list.p, pixels.data(), pixels.size() * 4));
        rt_host::check(rt_render_pixels(w_, h_, int64_t(pixels.size()), (const int32_t*)list.p, 0, samples, depth_, seed_ + pass, 0u,
                                        (float*)acc_.p, nullptr));
        if (done_ == 0) done_ = samples;   // image(): every pixel is normalised by its own sample count (accum.w)
    }
    // One time slice: passes until budget_ms of kernel time are used (at least one pass). Returns the
    // number of samples per pixel accumulated so far.
    int go(double budget_ms = 16.0) {
        s_.bind();
        double used = 0.0;
        do {
            rt_host::check(rt_render(w_, h_, 0, h_, done_, per_pass_, depth_, seed_, done_ == 0 ? 0u : RT_RENDER_CONTINUE,
                                     (float*)acc_.p, nullptr));
            float ms = 0.0f; int launches = 0;
            rt_last_kernel_ms(&ms, &launches);
            done_ += per_pass_;
            used += ms;
            const double per_sample = ms / per_pass_;   // size the next pass to about half a slice
            int next = per_sample > 0.0 ? int(0.5 * budget_ms / per_sample) : per_pass_ * 2;
            per_pass_ = next < 1 ? 1 : (next > 64 ? 64 : next);
        } while (used + last_pass_estimate() <= budget_ms);
        return done_;
    }
    int samples() const { return done_; }
    // calc::rgbPerSamples + writeToImg over the accumulators (Format_RGB32, mirrored)
    std::vector<uint32_t> image() {
        std::vector<uint32_t> out(size_t(w_) * h_);
        if (done_ == 0) return out;
        s_.bind();
        rt_host::check(rt_tonemap(w_, h_, (const float*)acc_.p, 0, (uint32_t*)img_.p));
        rt_host::check(rt_memcpy_d2h(out.data(), img_.p, out.size() * 4));
        return out;
    }
    std::vector<float> accumulators() {
        std::vector<float> out(size_t(w_) * h_ * 4);
        s_.bind();
        rt_host::check(rt_memcpy_d2h(out.data(), acc_.p, out.size() * 4));
        return out;
    }
private:
    double last_pass_estimate() const { float ms = 0.0f; int l = 0; rt_last_kernel_ms(&ms, &l); return double(ms); }
    rt_host::Session& s_;
    int w_, h_, depth_;
    uint64_t seed_;
    rt_host::DevBuf acc_, img_;
    int done_ = 0, per_pass_ = 1;
};

#endif  // RAY_AMD_HOST_H
