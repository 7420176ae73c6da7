// Mirror of the reference's shapes.h: aabb, shape, sphere, rectangle, xz_rect, yz_rect, box, constant_medium with
// the reference's constructors and public members. A shape DESCRIBES geometry to the HIP library
// (rt_flatten -> RtScene, include/ray_amd.h); shape::hit forwards to the library like hitFromList does.
// bounding_box / aabb are host arithmetic as in the reference (shapes.h:14-44 and the *_output_box members), and so
// are shape::pdf_value / random (shapes.h:56-62,117-135,202-219,268-284), the light-sampling half of hittable_pdf:
// they ask one shape about one ray (hit_host: sphere::hit / the rect hits restated for the host, shapes.h:77-109,
// 172-192,238-258,304-323), which is no work for a GPU. Pinned by tests/golden/{shapehits,materials}.npz (mirror_kat).
#ifndef SHAPES_H
#define SHAPES_H
#ifndef RAY_AMD_MIRROR_OUTER   // the header the translation unit included itself pulls in ray_amd_host.h at its end
#define RAY_AMD_MIRROR_OUTER
#define RAY_AMD_MIRROR_OUTER_IS_SHAPES_H
#endif

#include <optional>

#include "hitPosition.h"

namespace RectangleNormal {                                           // shapes.h:8-12
constexpr QVector3D xy{0, 0, 1};
constexpr QVector3D xz{0, 1, 0};
constexpr QVector3D yz{1, 0, 0};
}  // namespace RectangleNormal

class aabb {                                                          // shapes.h:14-44
public:
    aabb() {}
    aabb(const QVector3D& a, const QVector3D& b) : minimum(a), maximum(b) {}
    QVector3D min() const { return minimum; }
    QVector3D max() const { return maximum; }
    bool hit(const Ray& r, double t_min, double t_max) const {        // slab test, FP32 per axis as in the reference
        for (int a = 0; a < 3; a++) {
            float invD = 1.0f / r.direction[a];
            float t0 = (min()[a] - r.pointOfOrigin[a]) * invD;
            float t1 = (max()[a] - r.pointOfOrigin[a]) * invD;
            if (invD < 0.0f) std::swap(t0, t1);
            t_min = t0 > t_min ? t0 : t_min;
            t_max = t1 < t_max ? t1 : t_max;
            if (t_max <= t_min) return false;
        }
        return true;
    }
    QVector3D minimum;
    QVector3D maximum;
};

namespace rt_host {
struct flat_prim { int kind; double d[RT_PRIM_STRIDE]; std::shared_ptr<material> mat; int side; };
}

struct shape {                                                        // shapes.h:46-65
    shape() {}
    shape(QVector3D center) : center(center) {}
    virtual ~shape() {}
    // one ray against this shape alone, on the GPU (definition: ray_amd_host.h)
    virtual std::optional<hit_record> hit(const Ray& r, double t_min, double t_max) const;
    virtual bool bounding_box(aabb& output_box) const = 0;
    virtual double pdf_value(const QVector3D&, const QVector3D&) const { return 0.0; }              // shapes.h:56-58
    virtual QVector3D random(const QVector3D&) const { return QVector3D(1, 0, 0); }                  // shapes.h:60-62
    // the reference's own hit arithmetic for this ONE shape, on the host (what pdf_value needs); shapes without one: no hit
    virtual std::optional<hit_record> hit_host(const Ray&, double, double) const { return std::nullopt; }
    virtual void rt_flatten(std::vector<rt_host::flat_prim>& out) const = 0;
    QVector3D center;   // mutable, as RayView's slider does (rayview.cpp:130-134)
};

struct sphere : public shape {                                        // shapes.h:67-159
    sphere(QVector3D center, float radius, std::shared_ptr<material> material)
        : shape{center}, radius{radius}, radded{radius, radius, radius}, radiusSquared{radius * radius}, mat{material} {}
    bool bounding_box(aabb& output_box) const override { output_box = aabb(center - radded, center + radded); return true; }
    void rt_flatten(std::vector<rt_host::flat_prim>& out) const override {
        out.push_back({RT_PRIM_SPHERE, {center.x(), center.y(), center.z(), radius}, mat, -1});
    }
    std::optional<hit_record> hit_host(const Ray& ray, double t_min, double t_max) const override {   // shapes.h:77-109
        const QVector3D oc = ray.pointOfOrigin - center;
        const float a = ray.DirlengthSquared;
        const float b = QVector3D::dotProduct(oc, ray.direction);
        const float c = oc.lengthSquared() - radiusSquared;
        const float d = b * b - a * c;
        if (d < 0.0) return {};
        const double sqrtd = std::sqrt(double(d));
        double root = (-b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-b + sqrtd) / a;
            if (root < t_min || t_max < root) return {};
        }
        hit_record rec;
        rec.t = root;
        rec.p = ray.at(root);
        rec.mat_ptr = mat;
        const QVector3D outward = (rec.p - center) / radius;
        get_sphere_uv(outward, rec.u, rec.v);
        rec.setFaceNormal(ray, outward);
        return rec;
    }
    double pdf_value(const QVector3D& o, const QVector3D& v) const override {                          // shapes.h:117-127
        if (!hit_host(Ray(o, v), 0.001, calc::infinity).has_value()) return 0.0;
        const double cos_theta_max = std::sqrt(double(1 - radius * radius / (center - o).lengthSquared()));
        const double solid_angle = 2 * calc::pi * (1 - cos_theta_max);
        return 1 / solid_angle;
    }
    QVector3D random(const QVector3D& o) const override {                                              // shapes.h:129-135
        const QVector3D direction = center - o;
        const float distance_squared = direction.lengthSquared();
        onb uvw;
        uvw.build_from_w(direction);
        return uvw.local(calc::random_to_sphere(radius, distance_squared));
    }
    // shapes.h:144-158: p a point of the unit sphere; u the angle around Y from X = -1, v the angle from Y = -1, both in [0, 1]
    static void get_sphere_uv(const QVector3D& p, double& u, double& v) {
        const double theta = std::acos(double(-p.y()));
        const double phi = std::atan2(double(-p.z()), double(p.x())) + calc::pi;
        u = phi / (2.0f * calc::pi);
        v = theta / calc::pi;
    }
    const float radius;
    const QVector3D radded;
    const float radiusSquared;
    const std::shared_ptr<material> mat;
};

namespace rt_host {
template <int KIND>
struct rect_base : public shape {
    rect_base(double a0, double a1, double b0, double b1, double k, std::shared_ptr<material> m) : mp(m), q{a0, a1, b0, b1, k} {}
    void rt_flatten(std::vector<flat_prim>& out) const override { out.push_back({KIND, {q[0], q[1], q[2], q[3], q[4]}, mp, -1}); }
    bool bounding_box(aabb& output_box) const override {             // shapes.h:170,236,302: the plane padded by 0.0001
        const float a0 = float(q[0]), a1 = float(q[1]), b0 = float(q[2]), b1 = float(q[3]);
        if (KIND == RT_PRIM_XY_RECT) output_box = aabb(QVector3D(a0, b0, float(q[4] - 0.0001)), QVector3D(a1, b1, float(q[4] + 0.0001)));
        else if (KIND == RT_PRIM_XZ_RECT) output_box = aabb(QVector3D(a0, float(q[4] - 0.0001f), b0), QVector3D(a1, float(q[4] + 0.0001f), b1));
        else output_box = aabb(QVector3D(float(q[4] - 0.0001), a0, b0), QVector3D(float(q[4] + 0.0001), a1, b1));
        return true;
    }
    // shapes.h:172-192, 238-258, 304-323: the plane distance and the in-plane point in double (k and the bounds are doubles)
    std::optional<hit_record> hit_host(const Ray& r, double t_min, double t_max) const override {
        const QVector3D &o = r.pointOfOrigin, &d = r.direction;
        const float ok = KIND == RT_PRIM_XY_RECT ? o.z() : (KIND == RT_PRIM_XZ_RECT ? o.y() : o.x());
        const float dk = KIND == RT_PRIM_XY_RECT ? d.z() : (KIND == RT_PRIM_XZ_RECT ? d.y() : d.x());
        const float oa = KIND == RT_PRIM_YZ_RECT ? o.y() : o.x(), da = KIND == RT_PRIM_YZ_RECT ? d.y() : d.x();
        const float ob = KIND == RT_PRIM_XY_RECT ? o.y() : o.z(), db = KIND == RT_PRIM_XY_RECT ? d.y() : d.z();
        const double t = (q[4] - ok) / dk;
        if (t < t_min || t > t_max) return {};
        const double a = oa + t * da, b = ob + t * db;
        if (a < q[0] || a > q[1] || b < q[2] || b > q[3]) return {};
        hit_record rec;
        rec.u = (a - q[0]) / (q[1] - q[0]);
        rec.v = (b - q[2]) / (q[3] - q[2]);
        rec.t = t;
        rec.setFaceNormal(r, KIND == RT_PRIM_XY_RECT ? RectangleNormal::xy : (KIND == RT_PRIM_XZ_RECT ? RectangleNormal::xz : RectangleNormal::yz));
        rec.mat_ptr = mp;
        rec.p = r.at(t);
        return rec;
    }
    // shapes.h:202-219, 268-284 (rectangle and xz_rect; yz_rect has neither and keeps the base class's 0 and (1, 0, 0))
    double pdf_value(const QVector3D& origin, const QVector3D& v) const override {
        if (KIND == RT_PRIM_YZ_RECT) return 0.0;
        const auto h = hit_host(Ray(origin, v), 0.001, calc::infinity);
        if (!h.has_value()) return 0.0;
        const double area = (q[1] - q[0]) * (q[3] - q[2]);
        const double distance_squared = h->t * h->t * v.lengthSquared();
        const double cosine = std::fabs(double(QVector3D::dotProduct(v, h->normal) / v.length()));
        return distance_squared / (cosine * area);
    }
    QVector3D random(const QVector3D& origin) const override {
        if (KIND == RT_PRIM_YZ_RECT) return QVector3D(1, 0, 0);
        // QVector3D(random_double(a0, a1), k, random_double(b0, b1)) - origin for BOTH kinds: the reference's XY rectangle also puts k
        // in the y slot; its binary (g++) evaluates the constructor's arguments right to left, so the (b0, b1)
        // draw comes first (pinned by tests/golden/materials.npz)
        const float second = calc::random_double(float(q[2]), float(q[3]));
        const float first = calc::random_double(float(q[0]), float(q[1]));
        return QVector3D(first, float(q[4]), second) - origin;
    }
    std::shared_ptr<material> mp;
    const double q[5];   // (x0, x1, y0, y1, k) / (x0, x1, z0, z1, k) / (y0, y1, z0, z1, k)
};
}  // namespace rt_host
// the reference's public members by their names (shapes.h:222, 288, 335)
struct rectangle : public rt_host::rect_base<RT_PRIM_XY_RECT> {                                                // shapes.h:161-224
    using rect_base::rect_base;
    const double &x0 = q[0], &x1 = q[1], &y0 = q[2], &y1 = q[3], &k = q[4];
};
class xz_rect : public rt_host::rect_base<RT_PRIM_XZ_RECT> {                                                   // shapes.h:226-290
public:
    using rect_base::rect_base;
    const double &x0 = q[0], &x1 = q[1], &z0 = q[2], &z1 = q[3], &k = q[4];
};
class yz_rect : public rt_host::rect_base<RT_PRIM_YZ_RECT> {                                                   // shapes.h:292-337
public:
    using rect_base::rect_base;
    const double &y0 = q[0], &y1 = q[1], &z0 = q[2], &z1 = q[3], &k = q[4];
};

class box : public shape {                                            // shapes.h:339-372
public:
    box() {}
    box(const QVector3D& p0, const QVector3D& p1, std::shared_ptr<material> ptr) : box_min(p0), box_max(p1) {
        sides.push_back(std::make_shared<rectangle>(p0.x(), p1.x(), p0.y(), p1.y(), p1.z(), ptr));
        sides.push_back(std::make_shared<rectangle>(p0.x(), p1.x(), p0.y(), p1.y(), p0.z(), ptr));
        sides.push_back(std::make_shared<xz_rect>(p0.x(), p1.x(), p0.z(), p1.z(), p1.y(), ptr));
        sides.push_back(std::make_shared<xz_rect>(p0.x(), p1.x(), p0.z(), p1.z(), p0.y(), ptr));
        sides.push_back(std::make_shared<yz_rect>(p0.y(), p1.y(), p0.z(), p1.z(), p1.x(), ptr));
        sides.push_back(std::make_shared<yz_rect>(p0.y(), p1.y(), p0.z(), p1.z(), p0.x(), ptr));
    }
    void rt_flatten(std::vector<rt_host::flat_prim>& out) const override {   // the six sides in constructor order
        for (size_t i = 0; i < sides.size(); ++i) {
            const size_t before = out.size();
            sides[i]->rt_flatten(out);
            for (size_t j = before; j < out.size(); ++j) out[j].side = int(i);
        }
    }
    bool bounding_box(aabb& output_box) const override { output_box = aabb(box_min, box_max); return boundingBoxFromList(sides, output_box); }
    QVector3D box_min;
    QVector3D box_max;
    std::vector<std::shared_ptr<shape>> sides;
};

// shapes.h:374-455. boundary: a sphere or a box (the two shapes a smoke volume is built from); density d.
class constant_medium : public shape {
public:
    constant_medium(std::shared_ptr<shape> b, double d, std::shared_ptr<texture> a)
        : boundary(b), phase_function(std::make_shared<isotropic>(a)), neg_inv_density(-1 / d) {}
    constant_medium(std::shared_ptr<shape> b, double d, QVector3D c)
        : boundary(b), phase_function(std::make_shared<isotropic>(c)), neg_inv_density(-1 / d) {}
    void rt_flatten(std::vector<rt_host::flat_prim>& out) const override {
        if (auto* s = dynamic_cast<sphere*>(boundary.get())) {
            out.push_back({RT_PRIM_MEDIUM_SPHERE, {s->center.x(), s->center.y(), s->center.z(), s->radius, neg_inv_density}, phase_function, -1});
        } else if (auto* b = dynamic_cast<box*>(boundary.get())) {
            out.push_back({RT_PRIM_MEDIUM_BOX, {b->box_min.x(), b->box_min.y(), b->box_min.z(), b->box_max.x(), b->box_max.y(), b->box_max.z(),
                                                neg_inv_density}, phase_function, -1});
        } else {
            throw rt_host::error("constant_medium: the boundary must be a sphere or a box");
        }
    }
    bool bounding_box(aabb& output_box) const override { return boundary->bounding_box(output_box); }
    std::shared_ptr<shape> boundary;
    std::shared_ptr<material> phase_function;
    double neg_inv_density;
};

// not in the reference (BASELINE.json configs[3]; include/ray_amd_tri.h)
struct triangle : public shape {
    triangle(QVector3D a, QVector3D b, QVector3D c, std::shared_ptr<material> m) : v0(a), v1(b), v2(c), mat(m) {}
    void rt_flatten(std::vector<rt_host::flat_prim>& out) const override {
        out.push_back({RT_PRIM_TRIANGLE, {v0.x(), v0.y(), v0.z(), v1.x(), v1.y(), v1.z(), v2.x(), v2.y(), v2.z()}, mat, -1});
    }
    bool bounding_box(aabb& output_box) const override {
        output_box = aabb(QVector3D(std::fmin(std::fmin(v0.x(), v1.x()), v2.x()), std::fmin(std::fmin(v0.y(), v1.y()), v2.y()), std::fmin(std::fmin(v0.z(), v1.z()), v2.z())),
                          QVector3D(std::fmax(std::fmax(v0.x(), v1.x()), v2.x()), std::fmax(std::fmax(v0.y(), v1.y()), v2.y()), std::fmax(std::fmax(v0.z(), v1.z()), v2.z())));
        return true;
    }
    QVector3D v0, v1, v2;
    std::shared_ptr<material> mat;
};

// hitPosition.cpp:55-61
inline double hittable_pdf::value(const QVector3D& direction) const { return ptr->pdf_value(o, direction); }
inline QVector3D hittable_pdf::generate() const { return ptr->random(o); }

// hitPosition.cpp:24-53
inline aabb surrounding_box(aabb box0, aabb box1) {
    QVector3D small(float(std::fmin(box0.min().x(), box1.min().x())), float(std::fmin(box0.min().y(), box1.min().y())),
                    float(std::fmin(box0.min().z(), box1.min().z())));
    QVector3D big(float(std::fmax(box0.max().x(), box1.max().x())), float(std::fmax(box0.max().y(), box1.max().y())),
                  float(std::fmax(box0.max().z(), box1.max().z())));
    return aabb(small, big);
}
inline bool boundingBoxFromList(const std::vector<std::shared_ptr<shape>>& list, aabb& output_box) {
    if (list.empty()) return false;
    aabb temp_box;
    bool first_box = true;
    for (const auto& object : list) {
        if (!object->bounding_box(temp_box)) return false;
        output_box = first_box ? temp_box : surrounding_box(output_box, temp_box);
        first_box = false;
    }
    return true;
}

#ifdef RAY_AMD_MIRROR_OUTER_IS_SHAPES_H
#undef RAY_AMD_MIRROR_OUTER_IS_SHAPES_H
#undef RAY_AMD_MIRROR_OUTER
#include "ray_amd_host.h"
#endif
#endif  // SHAPES_H
