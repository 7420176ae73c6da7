// Mirror of the reference's hitPosition.h: textures, materials, pdfs, hit_record and the hitFromList entry point, with
// the reference's class names, constructors and signatures. For the HIP library a material class only DESCRIBES a
// material (rt_kind / rt_rgb / ...: calc::ray_color runs scatter, emitted and the pdfs on the GPU, ray_amd/csrc/). A
// caller that uses the classes one at a time -- material::scatter / scattering_pdf / emitted, onb, cosine_pdf,
// hittable_pdf, mixture_pdf (hitPosition.h:72-93,112-185, hitPosition.cpp:55-72) -- gets the reference's HOST arithmetic
// on the calc:: engines, pinned bit for bit against tests/golden/materials.npz by mirror_kat (no GPU involved).
#ifndef HITPOSITION_H
#define HITPOSITION_H
#ifndef RAY_AMD_MIRROR_OUTER   // the header the translation unit included itself pulls in ray_amd_host.h at its end
#define RAY_AMD_MIRROR_OUTER
#define RAY_AMD_MIRROR_OUTER_IS_HITPOSITION_H
#endif

#include <memory>
#include <optional>
#include <stdexcept>
#include <vector>

#include "Ray.h"
#include "calc.h"
#include "ray_amd.h"

struct shape;
class aabb;
struct hit_record;

namespace rt_host {
struct error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc) { if (rc != 0) throw error(rt_last_error()); }
}  // namespace rt_host

class texture {                                                       // hitPosition.h:14-17
public:
    virtual ~texture() {}
    virtual QVector3D value(double u, double v, const QVector3D& p) const = 0;
};
class solid_color : public texture {                                  // hitPosition.h:28-48
public:
    solid_color() {}
    solid_color(QVector3D c) : color_value(c) {}
    solid_color(double red, double green, double blue) : solid_color(QVector3D(float(red), float(green), float(blue))) {}
    QVector3D value(double, double, const QVector3D&) const override { return color_value; }
private:
    QVector3D color_value;
};

class material;
class pdf {                                                           // hitPosition.h:19-26
public:
    virtual ~pdf() {}
    virtual double value(const QVector3D& direction) const = 0;
    virtual QVector3D generate() const = 0;
};
struct hit_record {                                                   // hitPosition.h:95-109
    QVector3D p;
    QVector3D normal;
    std::shared_ptr<material> mat_ptr;
    double t = 0;
    double u = 0;       // texture coordinates (shapes.h:144-158,183-184): host arithmetic on the record the library returns,
    double v = 0;       // exactly the reference's expressions (rt_host::fill_uv)
    bool front_face = false;
    int rt_prim = -1;   // flattened primitive index (include/ray_amd.h); not in the reference
    inline void setFaceNormal(const Ray& r, const QVector3D& outward_normal) {
        front_face = QVector3D::dotProduct(r.direction, outward_normal) < 0.0f;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

struct scatter_record {                                               // hitPosition.h:72-77 (members initialised here: the reference
    Ray specular_ray;                                                 // leaves them indeterminate and isotropic::scatter never sets
    bool is_specular = false;                                         // is_specular)
    QVector3D attenuation;
    std::shared_ptr<pdf> pdf_ptr;
};

// hitPosition.h:79-93. rt_* : how the material is described to the library (include/ray_amd.h RT_MAT_*).
class material {
public:
    virtual ~material() {}
    virtual bool scatter(const Ray&, const hit_record&, scatter_record&) const { return false; }
    virtual double scattering_pdf(const Ray&, const hit_record&, const Ray&) const { return 0; }
    virtual QVector3D emitted(const Ray&, const hit_record&, double, double, const QVector3D&) const { return img::defaultVec; }
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
class onb {                                                           // hitPosition.h:112-139
public:
    onb() {}
    QVector3D operator[](int i) const { return axis[i]; }
    QVector3D u() const { return axis[0]; }
    QVector3D v() const { return axis[1]; }
    QVector3D w() const { return axis[2]; }
    QVector3D local(double a, double b, double c) const { return float(a) * u() + float(b) * v() + float(c) * w(); }   // (scalar * vector takes a float)
    QVector3D local(const QVector3D& a) const { return a.x() * u() + a.y() * v() + a.z() * w(); }
    void build_from_w(const QVector3D& n) {
        axis[2] = n.normalized();
        const QVector3D a = std::fabs(double(w().x())) > 0.9 ? QVector3D(0, 1, 0) : QVector3D(1, 0, 0);
        axis[1] = QVector3D::crossProduct(w(), a).normalized();
        axis[0] = QVector3D::crossProduct(w(), v());
    }
    QVector3D axis[3];
};
class cosine_pdf : public pdf {                                       // hitPosition.h:141-156
public:
    cosine_pdf(const QVector3D& w) { uvw.build_from_w(w); }
    double value(const QVector3D& direction) const override {
        const float cosine = QVector3D::dotProduct(direction.normalized(), uvw.w());
        return cosine <= 0 ? 0 : cosine / calc::pi;
    }
    QVector3D generate() const override { return uvw.local(calc::random_cosine_direction()); }
    onb uvw;
};
class hittable_pdf : public pdf {                                     // hitPosition.h:158-170, hitPosition.cpp:55-61 (bodies: shapes.h, after shape)
public:
    hittable_pdf() {}
    hittable_pdf(std::shared_ptr<shape> p, const QVector3D& origin) : o(origin), ptr(p) {}
    double value(const QVector3D& direction) const override;
    QVector3D generate() const override;
    QVector3D o;
    std::shared_ptr<shape> ptr;
};
class mixture_pdf : public pdf {                                      // hitPosition.h:172-185, hitPosition.cpp:63-72
public:
    mixture_pdf(pdf* p0, pdf* p1) { p[0] = p0; p[1] = p1; }
    double value(const QVector3D& direction) const override { return 0.5 * p[0]->value(direction) + 0.5 * p[1]->value(direction); }
    QVector3D generate() const override { return calc::random_double01() < 0.5 ? p[0]->generate() : p[1]->generate(); }
    pdf* p[2];
};

class lambertian : public material {                                  // hitPosition.h:197-220
public:
    lambertian(const QVector3D& a) : albedo(std::make_shared<solid_color>(a)) {}
    lambertian(std::shared_ptr<texture> a) : albedo(a) {}
    bool scatter(const Ray&, const hit_record& rec, scatter_record& srec) const override {
        srec.is_specular = false;
        srec.attenuation = albedo->value(rec.u, rec.v, rec.p);
        srec.pdf_ptr = std::make_shared<cosine_pdf>(rec.normal);
        return true;
    }
    double scattering_pdf(const Ray&, const hit_record& rec, const Ray& scattered) const override {
        const float cosine = QVector3D::dotProduct(rec.normal, scattered.direction.normalized());
        return cosine < 0 ? 0 : cosine / calc::pi;
    }
    int rt_kind() const override { return RT_MAT_LAMBERTIAN; }
    QVector3D rt_rgb() const override { return rt_host::solid(albedo); }
    std::shared_ptr<texture> albedo;
};
class metal : public material {                                       // hitPosition.h:222-244
public:
    metal(const QVector3D& a, double f) : albedo(a), fuzz(f < 1.0f ? f : 1.0f) {}
    bool scatter(const Ray& r_in, const hit_record& rec, scatter_record& srec) const override {   // (always draws, also for fuzz 0)
        const QVector3D reflected = calc::reflect(r_in.direction.normalized(), rec.normal);
        srec.specular_ray = Ray(rec.p, reflected + fuzz * calc::random_in_unit_sphere());
        srec.attenuation = albedo;
        srec.is_specular = true;
        srec.pdf_ptr = nullptr;
        return true;
    }
    int rt_kind() const override { return RT_MAT_METAL; }
    QVector3D rt_rgb() const override { return albedo; }
    float rt_fuzz() const override { return fuzz; }
    QVector3D albedo;
    float fuzz;
};
class dielectric : public material {                                  // hitPosition.h:246-288
public:
    dielectric(double index_of_refraction) : ir(index_of_refraction) {}
    bool scatter(const Ray& r_in, const hit_record& rec, scatter_record& srec) const override {
        srec.is_specular = true;
        srec.pdf_ptr = nullptr;
        srec.attenuation = QVector3D(1.0f, 1.0f, 1.0f);
        const double refraction_ratio = rec.front_face ? (1.0f / ir) : ir;
        const QVector3D unit_direction = r_in.direction.normalized();
        const float cos_theta = float(std::fmin(double(QVector3D::dotProduct(-unit_direction, rec.normal)), 1.0));
        const float sin_theta = float(std::sqrt(double(1.0f - cos_theta * cos_theta)));
        const bool cannot_refract = refraction_ratio * sin_theta > 1.0f;
        // (the random number is drawn only when the ray can refract: || short-circuits)
        const bool reflects = cannot_refract || reflectance(cos_theta, float(refraction_ratio)) > calc::random_double01();
        srec.specular_ray = Ray(rec.p, reflects ? calc::reflect(unit_direction, rec.normal) : calc::refract(unit_direction, rec.normal, refraction_ratio));
        return true;
    }
    static float reflectance(float cosine, float ref_idx) {          // Schlick, hitPosition.h:281-287: pow in double, narrowed
        float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
        r0 = r0 * r0;
        return r0 + (1.0f - r0) * float(std::pow(double(1.0f - cosine), 5));
    }
    int rt_kind() const override { return RT_MAT_DIELECTRIC; }
    double rt_ir() const override { return ir; }
    double ir;
};
class diffuse_light : public material {                               // hitPosition.h:290-314
public:
    diffuse_light(std::shared_ptr<texture> a) : emiter(a) {}
    diffuse_light(QVector3D c) : emiter(std::make_shared<solid_color>(c)) {}
    bool scatter(const Ray&, const hit_record&, scatter_record&) const override { return false; }
    QVector3D emitted(const Ray&, const hit_record&, double u, double v, const QVector3D& p) const override { return emiter->value(u, v, p); }
    int rt_kind() const override { return RT_MAT_DIFFUSE_LIGHT; }
    QVector3D rt_rgb() const override { return rt_host::solid(emiter); }
    std::shared_ptr<texture> emiter;
};
class isotropic : public material {                                   // hitPosition.h:316-337
public:
    isotropic(QVector3D c) : albedo(std::make_shared<solid_color>(c)) {}
    isotropic(std::shared_ptr<texture> a) : albedo(a) {}
    bool scatter(const Ray&, const hit_record& rec, scatter_record& scattered) const override {   // hitPosition.h:327-333
        scattered.specular_ray = Ray(rec.p, calc::random_in_unit_sphere());
        scattered.attenuation = albedo->value(rec.u, rec.v, rec.p);
        scattered.is_specular = true;    // the reference never sets it and calc.cpp:118 reads it: the reading the library adopts
        return true;
    }
    int rt_kind() const override { return RT_MAT_ISOTROPIC; }
    QVector3D rt_rgb() const override { return rt_host::solid(albedo); }
    std::shared_ptr<texture> albedo;
};

// hitPosition.h:187-191, hitPosition.cpp:6-22 -- ON THE GPU (definition: ray_amd_host.h)
std::optional<hit_record> hitFromList(const std::vector<std::shared_ptr<shape>>& sphereList, const Ray& ray, double t_min, double t_max);
// hitPosition.h:193-195, hitPosition.cpp:24-53: host arithmetic (definitions: shapes.h, after aabb)
bool boundingBoxFromList(const std::vector<std::shared_ptr<shape>>& list, aabb& output_box);
aabb surrounding_box(aabb box0, aabb box1);

#ifdef RAY_AMD_MIRROR_OUTER_IS_HITPOSITION_H
#undef RAY_AMD_MIRROR_OUTER_IS_HITPOSITION_H
#undef RAY_AMD_MIRROR_OUTER
#include "ray_amd_host.h"
#endif
#endif  // HITPOSITION_H
