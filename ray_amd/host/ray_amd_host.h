// C++ host mirror of the reference's ray-path interface (Ray.h, calc.h, hitPosition.h, shapes.h,
// camera.h of joblobob/ray) on top of the C ABI of include/ray_amd.h. Same class names and
// constructor signatures as the reference, so scene-building code written against it (e.g.
// RayView::normal_scene, rayview.cpp:233-260) compiles unchanged; the classes only DESCRIBE the
// scene -- every intersection, bounce and tonemap runs in the HIP library. No CPU fallback.
//
// QVector3D: with -DRAY_AMD_USE_QT the real <QVector3D> is used; otherwise a minimal stand-in
// with the accessors scene-building code needs (no arithmetic on the ray path happens here).
#ifndef RAY_AMD_HOST_H
#define RAY_AMD_HOST_H

#include <cstdint>
#include <limits>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "ray_amd.h"

#ifdef RAY_AMD_USE_QT
#include <QVector3D>
#else
class QVector3D {
public:
    constexpr QVector3D() : xp(0), yp(0), zp(0) {}
    constexpr QVector3D(float x, float y, float z) : xp(x), yp(y), zp(z) {}
    constexpr float x() const { return xp; }
    constexpr float y() const { return yp; }
    constexpr float z() const { return zp; }
    void setX(float v) { xp = v; }
    void setY(float v) { yp = v; }
    void setZ(float v) { zp = v; }
    float operator[](int i) const { return i == 0 ? xp : (i == 1 ? yp : zp); }
    QVector3D& operator+=(const QVector3D& o) { xp += o.xp; yp += o.yp; zp += o.zp; return *this; }
    QVector3D& operator-=(const QVector3D& o) { xp -= o.xp; yp -= o.yp; zp -= o.zp; return *this; }
    QVector3D& operator*=(float f) { xp *= f; yp *= f; zp *= f; return *this; }
    friend QVector3D operator+(const QVector3D& a, const QVector3D& b) { return QVector3D(a.xp + b.xp, a.yp + b.yp, a.zp + b.zp); }
    friend QVector3D operator-(const QVector3D& a, const QVector3D& b) { return QVector3D(a.xp - b.xp, a.yp - b.yp, a.zp - b.zp); }
    friend QVector3D operator*(const QVector3D& a, float f) { return QVector3D(a.xp * f, a.yp * f, a.zp * f); }
    friend QVector3D operator*(float f, const QVector3D& a) { return QVector3D(a.xp * f, a.yp * f, a.zp * f); }
    friend QVector3D operator*(const QVector3D& a, const QVector3D& b) { return QVector3D(a.xp * b.xp, a.yp * b.yp, a.zp * b.zp); }
private:
    float xp, yp, zp;
};
#endif

namespace rt_host {
struct error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc) { if (rc != 0) throw error(rt_last_error()); }
}  // namespace rt_host

// ---- calc.h:13-32 ------------------------------------------------------------------------------
namespace img {
constexpr float aspect_ratio = 16.0f / 9.0f;
constexpr int width = 800;
constexpr int height = static_cast<int>(width / aspect_ratio);
constexpr QVector3D defaultVec{0.0f, 0.0f, 0.0f};
constexpr QVector3D gradientBgVec{0.5f, 0.7f, 1.0f};
}  // namespace img

// ---- Ray.h ---------------------------------------------------------------------------------------
struct Ray {
    Ray() {}
    Ray(const QVector3D& origin, const QVector3D& dir) : pointOfOrigin(origin), direction(dir) {}
    QVector3D pointOfOrigin;
    QVector3D direction;
};

// ---- hitPosition.h: textures and materials (descriptions only) -----------------------------------
class texture { public: virtual ~texture() {} virtual QVector3D value(double, double, const QVector3D&) const = 0; };
class solid_color : public texture {
public:
    solid_color() {}
    solid_color(QVector3D c) : color_value(c) {}
    solid_color(double r, double g, double b) : color_value(float(r), float(g), float(b)) {}
    QVector3D value(double, double, const QVector3D&) const override { return color_value; }
private:
    QVector3D color_value;
};
class material {
public:
    virtual ~material() {}
    virtual int rt_kind() const = 0;
    virtual QVector3D rt_rgb() const { return QVector3D(); }
    virtual float rt_fuzz() const { return 0.0f; }
    virtual double rt_ir() const { return 1.0; }
};
namespace rt_host {
inline QVector3D solid(const std::shared_ptr<texture>& t) {
    auto* s = dynamic_cast<solid_color*>(t.get());
    if (!s) throw error("only solid_color textures exist in the reference (hitPosition.h:28-48)");
    return s->value(0, 0, QVector3D());
}
}  // namespace rt_host
class lambertian : public material {
public:
    lambertian(const QVector3D& a) : albedo(std::make_shared<solid_color>(a)) {}
    lambertian(std::shared_ptr<texture> a) : albedo(a) {}
    int rt_kind() const override { return RT_MAT_LAMBERTIAN; }
    QVector3D rt_rgb() const override { return rt_host::solid(albedo); }
    std::shared_ptr<texture> albedo;
};
class metal : public material {
public:
    metal(const QVector3D& a, double f) : albedo(a), fuzz(f < 1.0f ? f : 1.0f) {}   // hitPosition.h:224-227
    int rt_kind() const override { return RT_MAT_METAL; }
    QVector3D rt_rgb() const override { return albedo; }
    float rt_fuzz() const override { return fuzz; }
    QVector3D albedo;
    float fuzz;
};
class dielectric : public material {
public:
    dielectric(double index_of_refraction) : ir(index_of_refraction) {}
    int rt_kind() const override { return RT_MAT_DIELECTRIC; }
    double rt_ir() const override { return ir; }
    double ir;
};
class diffuse_light : public material {
public:
    diffuse_light(std::shared_ptr<texture> a) : emiter(a) {}
    diffuse_light(QVector3D c) : emiter(std::make_shared<solid_color>(c)) {}
    int rt_kind() const override { return RT_MAT_DIFFUSE_LIGHT; }
    QVector3D rt_rgb() const override { return rt_host::solid(emiter); }
    std::shared_ptr<texture> emiter;
};

// hitPosition.h:95-109
struct hit_record {
    QVector3D p;
    QVector3D normal;
    std::shared_ptr<material> mat_ptr;
    double t = 0;
    bool front_face = false;
    int rt_prim = -1;   // flattened primitive index (include/ray_amd.h)
};

// ---- shapes.h (descriptions only) ------------------------------------------------------------------
namespace rt_host {
struct flat_prim { int kind; double d[RT_PRIM_STRIDE]; std::shared_ptr<material> mat; int side; };
}
struct shape {
    shape() {}
    shape(QVector3D c) : center(c) {}
    virtual ~shape() {}
    virtual void rt_flatten(std::vector<rt_host::flat_prim>& out) const = 0;
    QVector3D center;   // mutable, as RayView's slider does (rayview.cpp:130-134)
};
struct sphere : public shape {
    sphere(QVector3D c, float r, std::shared_ptr<material> m) : shape(c), radius(r), mat(m) {}
    void rt_flatten(std::vector<rt_host::flat_prim>& out) const override {
        out.push_back({RT_PRIM_SPHERE, {center.x(), center.y(), center.z(), radius}, mat, -1});
    }
    const float radius;
    const std::shared_ptr<material> mat;
};
namespace rt_host {
template <int KIND>
struct rect_base : public shape {
    rect_base(double a0, double a1, double b0, double b1, double k, std::shared_ptr<material> m) : q{a0, a1, b0, b1, k}, mp(m) {}
    void rt_flatten(std::vector<flat_prim>& out) const override { out.push_back({KIND, {q[0], q[1], q[2], q[3], q[4]}, mp, -1}); }
    const double q[5];
    const std::shared_ptr<material> mp;
};
}  // namespace rt_host
struct rectangle : public rt_host::rect_base<RT_PRIM_XY_RECT> { using rect_base::rect_base; };
class xz_rect : public rt_host::rect_base<RT_PRIM_XZ_RECT> { public: using rect_base::rect_base; };
class yz_rect : public rt_host::rect_base<RT_PRIM_YZ_RECT> { public: using rect_base::rect_base; };
class box : public shape {
public:
    box(const QVector3D& p0, const QVector3D& p1, std::shared_ptr<material> ptr) : box_min(p0), box_max(p1), mat(ptr) {}
    void rt_flatten(std::vector<rt_host::flat_prim>& out) const override {   // shapes.h:347-354
        const QVector3D &a = box_min, &b = box_max;
        out.push_back({RT_PRIM_XY_RECT, {a.x(), b.x(), a.y(), b.y(), b.z()}, mat, 0});
        out.push_back({RT_PRIM_XY_RECT, {a.x(), b.x(), a.y(), b.y(), a.z()}, mat, 1});
        out.push_back({RT_PRIM_XZ_RECT, {a.x(), b.x(), a.z(), b.z(), b.y()}, mat, 2});
        out.push_back({RT_PRIM_XZ_RECT, {a.x(), b.x(), a.z(), b.z(), a.y()}, mat, 3});
        out.push_back({RT_PRIM_YZ_RECT, {a.y(), b.y(), a.z(), b.z(), b.x()}, mat, 4});
        out.push_back({RT_PRIM_YZ_RECT, {a.y(), b.y(), a.z(), b.z(), a.x()}, mat, 5});
    }
    QVector3D box_min, box_max;
    std::shared_ptr<material> mat;
};
// not in the reference (BASELINE.json configs[3])
struct triangle : public shape {
    triangle(QVector3D a, QVector3D b, QVector3D c, std::shared_ptr<material> m) : v0(a), v1(b), v2(c), mat(m) {}
    void rt_flatten(std::vector<rt_host::flat_prim>& out) const override {
        out.push_back({RT_PRIM_TRIANGLE, {v0.x(), v0.y(), v0.z(), v1.x(), v1.y(), v1.z(), v2.x(), v2.y(), v2.z()}, mat, -1});
    }
    QVector3D v0, v1, v2;
    std::shared_ptr<material> mat;
};

// ---- camera.h ----------------------------------------------------------------------------------------
class camera {
public:
    camera(QVector3D lookfrom, QVector3D lookat, QVector3D vup, double focus_dist, double vfov = 40.0,
           double aspect_ratio = img::aspect_ratio, double aperture = 0.1) {
        resetCam(lookfrom, lookat, vup, focus_dist, vfov, aspect_ratio, aperture);
    }
    void resetCam(QVector3D lookfrom, QVector3D lookat, QVector3D vup, double focus_dist, double vfov = 40.0,
                  double aspect_ratio = img::aspect_ratio, double aperture = 0.1) {
        const float f[3] = {lookfrom.x(), lookfrom.y(), lookfrom.z()}, a[3] = {lookat.x(), lookat.y(), lookat.z()},
                    u[3] = {vup.x(), vup.y(), vup.z()};
        rt_host::check(rt_camera_from_lookat(f, a, u, focus_dist, vfov, aspect_ratio, aperture, &c));
    }
    const RtCamera& rt() const { return c; }
private:
    RtCamera c;
};

// ---- the device-backed entry points --------------------------------------------------------------------
namespace rt_host {

// Flattens worldObjects (+ lights, background) and uploads it; keeps the flattened material list so
// hit records can point back at the caller's material objects.
class Session {
public:
    explicit Session(int device = 0) { check(rt_init(device)); }
    void upload(const std::vector<std::shared_ptr<shape>>& world, const std::shared_ptr<shape>& lights,
                const QVector3D& background) {
        prims.clear();
        std::vector<int32_t> kind, mat, outer, side;
        std::vector<double> data;
        mats.clear();
        for (size_t i = 0; i < world.size(); ++i) {
            size_t before = prims.size();
            world[i]->rt_flatten(prims);
            for (size_t j = before; j < prims.size(); ++j) {
                const flat_prim& p = prims[j];
                if (!p.mat) throw error("world shape without a material");
                int m = -1;
                for (size_t k = 0; k < mats.size(); ++k) if (mats[k] == p.mat) m = int(k);
                if (m < 0) { mats.push_back(p.mat); m = int(mats.size()) - 1; }
                kind.push_back(p.kind); mat.push_back(m); outer.push_back(int(i)); side.push_back(p.side);
                data.insert(data.end(), p.d, p.d + RT_PRIM_STRIDE);
            }
        }
        if (mats.empty()) mats.push_back(std::make_shared<lambertian>(QVector3D(0, 0, 0)));
        std::vector<int32_t> mk; std::vector<float> rgb, fz; std::vector<double> ir;
        for (auto& m : mats) {
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
            }   // box / yz_rect lights use the base class pdf (shapes.h:56-62)
        }
        s.background[0] = background.x(); s.background[1] = background.y(); s.background[2] = background.z();
        check(rt_upload_scene(&s));
        outer_of = outer;
    }
    std::vector<flat_prim> prims;
    std::vector<std::shared_ptr<material>> mats;
    std::vector<int32_t> outer_of;
};

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(uint64_t bytes) { check(rt_malloc(&p, bytes)); }
    ~DevBuf() { if (p) rt_free(p); }
    DevBuf(const DevBuf&) = delete;
};

}  // namespace rt_host

// hitFromList (hitPosition.h:187-191): one ray against the list, on the GPU.
inline std::optional<hit_record> hitFromList(rt_host::Session& s, const Ray& ray, double t_min, double t_max) {
    const float o[3] = {ray.pointOfOrigin.x(), ray.pointOfOrigin.y(), ray.pointOfOrigin.z()};
    const float d[3] = {ray.direction.x(), ray.direction.y(), ray.direction.z()};
    rt_host::DevBuf bo(12), bd(12), bprim(4), bt(8), bp(12), bn(12), bf(1);
    rt_host::check(rt_memcpy_h2d(bo.p, o, 12)); rt_host::check(rt_memcpy_h2d(bd.p, d, 12));
    rt_host::check(rt_hit_rays(1, (const float*)bo.p, (const float*)bd.p, t_min, t_max, (int32_t*)bprim.p, (double*)bt.p,
                               (float*)bp.p, (float*)bn.p, (uint8_t*)bf.p));
    int32_t prim; double t; float p[3], n[3]; uint8_t ff;
    rt_host::check(rt_memcpy_d2h(&prim, bprim.p, 4)); rt_host::check(rt_memcpy_d2h(&t, bt.p, 8));
    rt_host::check(rt_memcpy_d2h(p, bp.p, 12)); rt_host::check(rt_memcpy_d2h(n, bn.p, 12)); rt_host::check(rt_memcpy_d2h(&ff, bf.p, 1));
    if (prim < 0) return std::nullopt;
    hit_record rec;
    rec.p = QVector3D(p[0], p[1], p[2]); rec.normal = QVector3D(n[0], n[1], n[2]); rec.t = t; rec.front_face = ff != 0;
    rec.rt_prim = prim; rec.mat_ptr = s.prims[size_t(prim)].mat;
    return rec;
}

namespace calc {
constexpr double infinity = std::numeric_limits<double>::infinity();
constexpr float smallestVal = 0.001f;

// calc::ray_color (calc.h:89-94) for one ray, on the GPU; `stream` selects the counter-RNG stream.
inline QVector3D ray_color(rt_host::Session& s, const Ray& r, int depth, bool drawOnlyColors = false, uint64_t seed = 0) {
    (void)s;
    const float o[3] = {r.pointOfOrigin.x(), r.pointOfOrigin.y(), r.pointOfOrigin.z()};
    const float d[3] = {r.direction.x(), r.direction.y(), r.direction.z()};
    rt_host::DevBuf bo(12), bd(12), bc(12);
    rt_host::check(rt_memcpy_h2d(bo.p, o, 12)); rt_host::check(rt_memcpy_h2d(bd.p, d, 12));
    rt_host::check(rt_trace_rays(1, (const float*)bo.p, (const float*)bd.p, depth, seed, drawOnlyColors ? RT_RENDER_NORMALS : 0u,
                                 (float*)bc.p));
    float c[3];
    rt_host::check(rt_memcpy_d2h(c, bc.p, 12));
    return QVector3D(c[0], c[1], c[2]);
}
}  // namespace calc

// RayView::renderAll (declared rayview.h:32, never defined in the reference) with the per-pixel
// contract of renderOneRay + writeToImg (rayview.cpp:67-87): returns the Format_RGB32 image
// (w*h uint32, mirrored on both axes as writeToImg does) and optionally the linear accumulators.
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

// RayView::go (rayview.cpp:98-117) for the GPU. The reference spends each 16 ms slice on random
// pixels, every one rendered with all its samples and overwritten on the next visit; here a slice is
// one or more FULL-FRAME passes of a few samples each, accumulated (RT_RENDER_ACCUMULATE), so the
// picture refines everywhere at once. Sample indices continue across slices (pass k renders samples
// [done, done + n) of every pixel), so N slices of the same scene give exactly the samples of one
// N-slice-long render. A scene or camera change (the sphere-Y slider, WASD: rayview.cpp:130-184) is
// reset(): re-upload and start again from sample 0.
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
        rt_host::DevBuf list(pixels.size() * 4);
        rt_host::check(rt_memcpy_h2d(list.p, pixels.data(), pixels.size() * 4));
        rt_host::check(rt_render_pixels(w_, h_, int64_t(pixels.size()), (const int32_t*)list.p, 0, samples, depth_, seed_ + pass, 0u,
                                        (float*)acc_.p, nullptr));
        if (done_ == 0) done_ = samples;   // image(): every pixel is normalised by its own sample count (accum.w)
    }
    // One time slice: passes until budget_ms of kernel time are used (at least one pass). Returns the
    // number of samples per pixel accumulated so far.
    int go(double budget_ms = 16.0) {
        double used = 0.0;
        do {
            rt_host::check(rt_render(w_, h_, 0, h_, done_, per_pass_, depth_, seed_, done_ == 0 ? 0u : RT_RENDER_ACCUMULATE,
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
        rt_host::check(rt_tonemap(w_, h_, (const float*)acc_.p, 0, (uint32_t*)img_.p));
        rt_host::check(rt_memcpy_d2h(out.data(), img_.p, out.size() * 4));
        return out;
    }
    std::vector<float> accumulators() {
        std::vector<float> out(size_t(w_) * h_ * 4);
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
