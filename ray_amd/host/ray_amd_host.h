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
#include <unordered_map>

namespace rt_host {

// Flattens worldObjects (+ lights, background) into the arrays of RtScene and uploads them; keeps the flattened
// list so hit records can point back at the caller's material objects. upload() returns without touching the device
// when the flattened content is, byte for byte, what the device already holds, so per-ray callers pay one
// flattening pass per call (into buffers that keep their capacity), not one upload.
class Session {
public:
    explicit Session(int device = 0) : device_(device) { check(rt_init(device)); }
    ~Session() { rt_set_device(device_); if (stage_) rt_free(stage_); if (pinned_) rt_free_host(pinned_); }
    Session(const Session&) = delete;
    void bind() const { check(rt_set_device(device_)); }
    // One device block that the per-ray entry points reuse for their rays and results (no rt_malloc / rt_free per call).
    void* stage(uint64_t bytes) {
        if (bytes > stage_bytes_) {
            bind();
            if (stage_) check(rt_free(stage_));
            stage_ = nullptr; stage_bytes_ = 0;
            const uint64_t want = bytes < 4096 ? 4096 : bytes;
            check(rt_malloc(&stage_, want));
            stage_bytes_ = want;
        }
        return stage_;
    }
    // The same in page-locked host memory that kernels read and write in place (rt_malloc_host): a call with a handful of rays
    // spends its time in the copy calls otherwise.
    void* pinned(uint64_t bytes) {
        if (bytes > pinned_bytes_) {
            bind();
            if (pinned_) check(rt_free_host(pinned_));
            pinned_ = nullptr; pinned_bytes_ = 0;
            const uint64_t want = bytes < 4096 ? 4096 : bytes;
            check(rt_malloc_host(&pinned_, want));
            pinned_bytes_ = want;
        }
        return pinned_;
    }
    bool upload(const std::vector<std::shared_ptr<shape>>& world, const std::shared_ptr<shape>& lights, const QVector3D& background) {
        // Per-ray callers (calc::ray_color in a loop) come here once per ray with the same list. Every call flattens the list again --
        // shapes and materials are the caller's objects and any member may have been written since (RayView's slider moves a centre,
        // rayview.cpp:130-134; a freed shape's address may be reused) -- into buffers that keep their capacity, and compares the bytes
        // with what the device holds: no allocation and no device call when nothing changed (about 10 ns per shape).
        Flat& f = next_;
        f.clear();
        fp_.clear(); ms_.clear(); mat_index_.clear();
        for (size_t i = 0; i < world.size(); ++i) {
            const size_t before = fp_.size();
            world[i]->rt_flatten(fp_);
            for (size_t j = before; j < fp_.size(); ++j) {
                const flat_prim& p = fp_[j];
                if (!p.mat) throw error("world shape without a material");
                int m = -1;
                if (ms_.size() <= 8) { for (size_t k = 0; k < ms_.size(); ++k) if (ms_[k].get() == p.mat.get()) { m = int(k); break; } }
                else { auto found = mat_index_.find(p.mat.get()); if (found != mat_index_.end()) m = found->second; }
                if (m < 0) {
                    m = int(ms_.size());
                    ms_.push_back(p.mat);
                    if (ms_.size() == 9) for (size_t k = 0; k < 8; ++k) mat_index_.emplace(ms_[k].get(), int(k));
                    if (ms_.size() > 8) mat_index_.emplace(p.mat.get(), m);
                }
                f.kind.push_back(p.kind); f.mat.push_back(m); f.outer.push_back(int(i)); f.side.push_back(p.side);
                f.data.insert(f.data.end(), p.d, p.d + RT_PRIM_STRIDE);
            }
        }
        if (ms_.empty()) ms_.push_back(std::make_shared<lambertian>(QVector3D(0, 0, 0)));
        for (auto& m : ms_) {
            f.mk.push_back(m->rt_kind()); QVector3D c = m->rt_rgb();
            f.rgb.insert(f.rgb.end(), {c.x(), c.y(), c.z()}); f.fz.push_back(m->rt_fuzz()); f.ir.push_back(m->rt_ir());
        }
        f.light_kind = RT_LIGHT_BASE;
        for (double& v : f.light_data) v = 0.0;
        if (lights) {
            lt_.clear(); lights->rt_flatten(lt_);
            if (lt_.size() == 1 && (lt_[0].kind == RT_PRIM_SPHERE || lt_[0].kind == RT_PRIM_XY_RECT || lt_[0].kind == RT_PRIM_XZ_RECT)) {
                f.light_kind = lt_[0].kind;
                for (int k = 0; k < RT_PRIM_STRIDE; ++k) f.light_data[k] = lt_[0].d[k];
            }   // box / yz_rect / medium lights use the base class pdf (shapes.h:56-62)
        }
        f.background[0] = background.x(); f.background[1] = background.y(); f.background[2] = background.z();
        bind();
        const bool fresh = !have_ || !(f == held_);
        if (fresh) {
            RtScene s{};
            s.n_prims = int(f.kind.size());
            s.prim_kind = f.kind.data(); s.prim_mat = f.mat.data(); s.prim_outer = f.outer.data(); s.prim_side = f.side.data();
            s.prim_data = f.data.data();
            s.n_mats = int(f.mk.size()); s.mat_kind = f.mk.data(); s.mat_rgb = f.rgb.data(); s.mat_fuzz = f.fz.data(); s.mat_ir = f.ir.data();
            s.light_kind = f.light_kind;
            for (int k = 0; k < RT_PRIM_STRIDE; ++k) s.light_data[k] = f.light_data[k];
            for (int k = 0; k < 3; ++k) s.background[k] = f.background[k];
            have_ = false;
            check(rt_upload_scene(&s));
            have_ = true; ++uploads;
            std::swap(held_, next_);
        }
        prims.swap(fp_); mats.swap(ms_);
        return fresh;
    }
    std::vector<flat_prim> prims;                     // the flattened list of the last upload() call: record -> the caller's material objects
    std::vector<std::shared_ptr<material>> mats;
    const std::vector<int32_t>& outer_of() const { return held_.outer; }   // flattened primitive -> index in the caller's list
    int uploads = 0;              // how often the device copy was actually replaced
private:
    struct Flat {                 // everything rt_upload_scene receives
        std::vector<int32_t> kind, mat, outer, side, mk;
        std::vector<double> data, ir;
        std::vector<float> rgb, fz;
        int32_t light_kind = 0;
        double light_data[RT_PRIM_STRIDE] = {};
        float background[3] = {};
        void clear() { kind.clear(); mat.clear(); outer.clear(); side.clear(); mk.clear(); data.clear(); ir.clear(); rgb.clear(); fz.clear(); }
        template <class T> static bool same(const std::vector<T>& a, const std::vector<T>& b) {   // bytes, not values: -0.0 is not 0.0 here, and NaN is NaN
            return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
        }
        bool operator==(const Flat& o) const {
            return same(kind, o.kind) && same(mat, o.mat) && same(outer, o.outer) && same(side, o.side) && same(mk, o.mk) && same(data, o.data) && same(ir, o.ir) &&
                   same(rgb, o.rgb) && same(fz, o.fz) && light_kind == o.light_kind && std::memcmp(light_data, o.light_data, sizeof light_data) == 0 &&
                   std::memcmp(background, o.background, sizeof background) == 0;
        }
    };
    int device_;
    bool have_ = false;
    Flat held_, next_;            // what the device holds; this call's list
    std::vector<flat_prim> fp_, lt_;
    std::vector<std::shared_ptr<material>> ms_;
    std::unordered_map<const material*, int> mat_index_;
    void* stage_ = nullptr;
    void* pinned_ = nullptr;
    uint64_t stage_bytes_ = 0, pinned_bytes_ = 0;
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

// hit_record::u, v (shapes.h:105,144-158 for a sphere, :183-184 / :249-250 / :315-316 for the rects) from the record the library
// returned, with the reference's own expressions on the host (same libm as the reference: exact). The outward normal of a sphere
// is the record's normal, flipped back when the hit was from inside (setFaceNormal only negates). Other kinds keep 0.
inline void fill_uv(const flat_prim& fp, const Ray& ray, hit_record& rec) {
    if (fp.kind == RT_PRIM_SPHERE) {
        sphere::get_sphere_uv(rec.front_face ? rec.normal : -rec.normal, rec.u, rec.v);
    } else if (fp.kind == RT_PRIM_XY_RECT || fp.kind == RT_PRIM_XZ_RECT || fp.kind == RT_PRIM_YZ_RECT) {
        const QVector3D &o = ray.pointOfOrigin, &d = ray.direction;
        const float oa = fp.kind == RT_PRIM_YZ_RECT ? o.y() : o.x(), da = fp.kind == RT_PRIM_YZ_RECT ? d.y() : d.x();
        const float ob = fp.kind == RT_PRIM_XY_RECT ? o.y() : o.z(), db = fp.kind == RT_PRIM_XY_RECT ? d.y() : d.z();
        rec.u = ((oa + rec.t * da) - fp.d[0]) / (fp.d[1] - fp.d[0]);
        rec.v = ((ob + rec.t * db) - fp.d[2]) / (fp.d[3] - fp.d[2]);
    }
}

// One ray against the uploaded list: ray in, record out through the session's page-locked block (one launch, no copy calls).
inline std::optional<hit_record> hit_uploaded(Session& s, const Ray& ray, double t_min, double t_max) {
    s.bind();
    struct In { float o[3], d[3]; float pad[2]; };
    struct Res { double t; float p[3], n[3]; int32_t prim; uint8_t ff; uint8_t pad[3]; };
    char* blk = static_cast<char*>(s.pinned(sizeof(In) + sizeof(Res)));
    In* in = reinterpret_cast<In*>(blk);
    Res* out = reinterpret_cast<Res*>(blk + sizeof(In));
    *in = In{{ray.pointOfOrigin.x(), ray.pointOfOrigin.y(), ray.pointOfOrigin.z()}, {ray.direction.x(), ray.direction.y(), ray.direction.z()}, {0, 0}};
    check(rt_hit_rays(1, in->o, in->d, t_min, t_max, &out->prim, &out->t, out->p, out->n, &out->ff));
    const Res r = *out;
    if (r.prim < 0) return std::nullopt;
    hit_record rec;
    rec.p = QVector3D(r.p[0], r.p[1], r.p[2]); rec.normal = QVector3D(r.n[0], r.n[1], r.n[2]); rec.t = r.t; rec.front_face = r.ff != 0;
    rec.rt_prim = r.prim; rec.mat_ptr = s.prims[size_t(r.prim)].mat;
    fill_uv(s.prims[size_t(r.prim)], ray, rec);
    return rec;
}

// calc::ray_color for n rays in ONE launch: ray i draws from the counter-RNG stream (seed, i, 0) (include/ray_amd.h, rt_trace_rays)
inline void trace_uploaded(Session& s, const Ray* rays, size_t n, int depth, bool drawOnlyColors, uint64_t seed, QVector3D* colours) {
    if (n == 0) return;
    s.bind();
    const unsigned flags = drawOnlyColors ? RT_RENDER_NORMALS : 0u;
    if (n <= 4096) {   // few rays: the kernel reads them from, and writes the colours to, page-locked host memory
        float* blk = static_cast<float*>(s.pinned(uint64_t(9 * n) * 4));
        for (size_t i = 0; i < n; ++i) {
            const QVector3D &o = rays[i].pointOfOrigin, &d = rays[i].direction;
            blk[3 * i] = o.x(); blk[3 * i + 1] = o.y(); blk[3 * i + 2] = o.z();
            blk[3 * n + 3 * i] = d.x(); blk[3 * n + 3 * i + 1] = d.y(); blk[3 * n + 3 * i + 2] = d.z();
        }
        check(rt_trace_rays(int64_t(n), blk, blk + 3 * n, depth, seed, flags, blk + 6 * n));
        for (size_t i = 0; i < n; ++i) colours[i] = QVector3D(blk[6 * n + 3 * i], blk[6 * n + 3 * i + 1], blk[6 * n + 3 * i + 2]);
        return;
    }
    std::vector<float> in(6 * n), out(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const QVector3D &o = rays[i].pointOfOrigin, &d = rays[i].direction;
        in[3 * i] = o.x(); in[3 * i + 1] = o.y(); in[3 * i + 2] = o.z();
        in[3 * n + 3 * i] = d.x(); in[3 * n + 3 * i + 1] = d.y(); in[3 * n + 3 * i + 2] = d.z();
    }
    float* blk = static_cast<float*>(s.stage(uint64_t(9 * n) * 4));
    check(rt_memcpy_h2d(blk, in.data(), uint64_t(6 * n) * 4));
    check(rt_trace_rays(int64_t(n), blk, blk + 3 * n, depth, seed, flags, blk + 6 * n));
    check(rt_memcpy_d2h(out.data(), blk + 6 * n, uint64_t(3 * n) * 4));
    for (size_t i = 0; i < n; ++i) colours[i] = QVector3D(out[3 * i], out[3 * i + 1], out[3 * i + 2]);
}
inline QVector3D trace_uploaded(Session& s, const Ray& r, int depth, bool drawOnlyColors, uint64_t seed) {
    QVector3D c;
    trace_uploaded(s, &r, 1, depth, drawOnlyColors, seed, &c);
    return c;
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
    return rt_host::trace_uploaded(s, r, depth, drawOnlyColors, rt_host::ray_color_stream()++);
}
// The same for a batch of rays in one launch (not in the reference, whose caller loops over single rays, rayview.cpp:79-84): what a
// caller that has its samples' rays at hand should use -- a launch costs the GPU tens of microseconds whether it traces one ray or
// a hundred thousand. Ray i draws from the counter-RNG stream (k, i, 0), k = this call's number as above.
inline std::vector<QVector3D> ray_color(const std::vector<Ray>& rays, const QVector3D& background, const std::vector<std::shared_ptr<shape>>& worldObjects,
                                        std::shared_ptr<shape>& lights, int depth, bool drawOnlyColors = false) {
    rt_host::Session& s = rt_host::default_session();
    s.upload(worldObjects, lights, background);
    std::vector<QVector3D> out(rays.size());
    rt_host::trace_uploaded(s, rays.data(), rays.size(), depth, drawOnlyColors, rt_host::ray_color_stream()++, out.data());
    return out;
}
// against an explicitly managed session whose scene is already uploaded; `seed` selects the counter-RNG stream
inline QVector3D ray_color(rt_host::Session& s, const Ray& r, int depth, bool drawOnlyColors = false, uint64_t seed = 0) {
    return rt_host::trace_uploaded(s, r, depth, drawOnlyColors, seed);
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
