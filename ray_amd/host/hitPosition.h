// Mirror of the reference's hitPosition.h: textures, materials, hit_record and the hitFromList entry point, with
// the reference's class names, constructors and signatures. The material classes only DESCRIBE a material to the
// HIP library (their scatter / emitted arithmetic runs inside calc::ray_color on the GPU, ray_amd/csrc/); the
// pdf classes of hitPosition.h:141-185 have no host-side counterpart for the same reason.
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

// hitPosition.h:79-93. rt_* : how the material is described to the library (include/ray_amd.h RT_MAT_*).
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
class lambertian : public material {                                  // hitPosition.h:197-220
public:
    lambertian(const QVector3D& a) : albedo(std::make_shared<solid_color>(a)) {}
    lambertian(std::shared_ptr<texture> a) : albedo(a) {}
    int rt_kind() const override { return RT_MAT_LAMBERTIAN; }
    QVector3D rt_rgb() const override { return rt_host::solid(albedo); }
    std::shared_ptr<texture> albedo;
};
class metal : public material {                                       // hitPosition.h:222-244
public:
    metal(const QVector3D& a, double f) : albedo(a), fuzz(f < 1.0f ? f : 1.0f) {}
    int rt_kind() const override { return RT_MAT_METAL; }
    QVector3D rt_rgb() const override { return albedo; }
    float rt_fuzz() const override { return fuzz; }
    QVector3D albedo;
    float fuzz;
};
class dielectric : public material {                                  // hitPosition.h:246-288
public:
    dielectric(double index_of_refraction) : ir(index_of_refraction) {}
    int rt_kind() const override { return RT_MAT_DIELECTRIC; }
    double rt_ir() const override { return ir; }
    double ir;
};
class diffuse_light : public material {                               // hitPosition.h:290-314
public:
    diffuse_light(std::shared_ptr<texture> a) : emiter(a) {}
    diffuse_light(QVector3D c) : emiter(std::make_shared<solid_color>(c)) {}
    int rt_kind() const override { return RT_MAT_DIFFUSE_LIGHT; }
    QVector3D rt_rgb() const override { return rt_host::solid(emiter); }
    std::shared_ptr<texture> emiter;
};
class isotropic : public material {                                   // hitPosition.h:316-337
public:
    isotropic(QVector3D c) : albedo(std::make_shared<solid_color>(c)) {}
    isotropic(std::shared_ptr<texture> a) : albedo(a) {}
    int rt_kind() const override { return RT_MAT_ISOTROPIC; }
    QVector3D rt_rgb() const override { return rt_host::solid(albedo); }
    std::shared_ptr<texture> albedo;
};

struct hit_record {                                                   // hitPosition.h:95-109
    QVector3D p;
    QVector3D normal;
    std::shared_ptr<material> mat_ptr;
    double t = 0;
    double u = 0;       // texture coordinates: solid colours ignore them, the library does not compute them
    double v = 0;
    bool front_face = false;
    int rt_prim = -1;   // flattened primitive index (include/ray_amd.h); not in the reference
    inline void setFaceNormal(const Ray& r, const QVector3D& outward_normal) {
        front_face = QVector3D::dotProduct(r.direction, outward_normal) < 0.0f;
        normal = front_face ? outward_normal : -outward_normal;
    }
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
