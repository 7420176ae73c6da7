// Mirror of the reference's calc.h / calc.cpp (constants, the three global minstd_rand0 engines and the samplers
// built on them, ray_color, rgbPerSamples, reflect / refract) with the reference's names and signatures.
// What runs where: calc::ray_color forwards to the HIP library (rt_trace_rays) -- its definition is in
// ray_amd_host.h; everything else in this header is the small host arithmetic a caller like RayView::renderOneRay
// (rayview.cpp:67-87) does BETWEEN its calls into the path (pixel draws, jitter, tonemap of one pixel), written
// with the same libstdc++ <random> objects and the same operation order as the reference, so the numbers agree.
#ifndef CALC_H
#define CALC_H
#ifndef RAY_AMD_MIRROR_OUTER   // the header the translation unit included itself pulls in ray_amd_host.h at its end
#define RAY_AMD_MIRROR_OUTER
#define RAY_AMD_MIRROR_OUTER_IS_CALC_H
#endif

#include <algorithm>
#include <cmath>
#include <limits>
#include <memory>
#include <random>
#include <vector>

#include "Ray.h"

struct shape;
namespace img {                                       // calc.h:13-24
constexpr float aspect_ratio = 16.0f / 9.0f;
constexpr int min0 = 0;
constexpr int width = 800;
constexpr int height = static_cast<int>(width / aspect_ratio);
constexpr int totalPixels = width * height;
constexpr QVector3D defaultVec{0.0f, 0.0f, 0.0f};
constexpr QVector3D infiniteZ{0.0f, 0.0f, -1.0f};
constexpr QVector3D gradientBgVec{0.5f, 0.7f, 1.0f};
constexpr QVector3D bgColor{1.0f, 1.0f, 1.0f};
}  // namespace img

namespace calc {
constexpr double infinity = std::numeric_limits<double>::infinity();   // calc.h:30-32
constexpr double pi = 3.1415926535897932385;
constexpr float smallestVal = 0.001f;
inline constexpr double degrees_to_radians(double degrees) { return degrees * pi / 180.0; }

// calc.h:103-113: three engines (seeded from std::random_device; a caller may re-seed them, they are plain globals)
inline std::random_device rd;
inline std::minstd_rand0 gen(rd());
inline std::random_device rd01;
inline std::minstd_rand0 gen01(rd01());
inline std::random_device rd11;
inline std::minstd_rand0 gen11(rd11());
inline std::uniform_real_distribution<float> simple_distribution(0.0f, 1.0f);
inline std::uniform_int_distribution<int> simple_distributionW(img::min0, img::width);
inline std::uniform_int_distribution<int> simple_distributionH(img::min0, img::height);
inline std::uniform_real_distribution<float> simple_distributionMinusOneToOne(-1.0f, 1.0f);

// calc.cpp:12-36
inline float random_double(float min, float max) { std::uniform_real_distribution<float> d(min, max); return d(gen); }
inline int random_doubleW() { return simple_distributionW(gen); }
inline int random_doubleH() { return simple_distributionH(gen); }
inline float random_double01() { return simple_distribution(gen01); }
inline float random_double11() { return simple_distributionMinusOneToOne(gen11); }
inline int random_int(int min, int max) { return static_cast<int>(random_double(float(min), float(max + 1))); }
// calc.cpp:38-52: brace-initialised, so the three draws happen in x, y, z order
inline QVector3D randomVec() { float x = random_double01(), y = random_double01(), z = random_double01(); return QVector3D(x, y, z); }
inline QVector3D randomVec11() { float x = random_double11(), y = random_double11(), z = random_double11(); return QVector3D(x, y, z); }
inline QVector3D randomVec(float min, float max) { float x = random_double(min, max), y = random_double(min, max), z = random_double(min, max); return QVector3D(x, y, z); }
// calc.cpp:54-62
inline QVector3D random_in_unit_sphere() {
    QVector3D point(-1.0f, -1.0f, -1.0f);
    while (point.lengthSquared() >= 1.0f) point = randomVec11();
    return point;
}
// calc.cpp:64-72: a parenthesised constructor call; the reference binary (g++) evaluates its arguments right to
// left, so the FIRST draw becomes y (pinned by tests/golden/camera.npz)
inline QVector3D random_in_unit_disk() {
    for (;;) {
        const float y = random_double11();
        const float x = random_double11();
        const QVector3D p(x, y, 0.0f);
        if (!(p.lengthSquared() >= 1.0f)) return p;
    }
}
inline QVector3D random_unit_vector() { return random_in_unit_sphere().normalized(); }           // calc.cpp:74-78
inline QVector3D random_in_hemisphere(const QVector3D& normal) {                                   // calc.cpp:80-87
    const QVector3D s = random_in_unit_sphere();
    return QVector3D::dotProduct(s, normal) > 0.0f ? s : -s;
}
inline QVector3D random_cosine_direction() {                                                       // calc.h:62-73
    const float r1 = random_double01(), r2 = random_double01();
    const double z = std::sqrt(double(1 - r2));
    const double phi = 2 * pi * r1;
    const double x = std::cos(phi) * std::sqrt(double(r2)), y = std::sin(phi) * std::sqrt(double(r2));
    return QVector3D(float(x), float(y), float(z));
}
inline QVector3D random_to_sphere(double radius, double distance_squared) {                        // calc.h:75-86
    const float r1 = random_double01(), r2 = random_double01();
    const double z = 1 + r2 * (std::sqrt(1 - radius * radius / distance_squared) - 1);
    const double phi = 2 * pi * r1;
    const double x = std::cos(phi) * std::sqrt(1 - z * z), y = std::sin(phi) * std::sqrt(1 - z * z);
    return QVector3D(float(x), float(y), float(z));
}

// calc.h:89-94, calc.cpp:90-131 -- ON THE GPU (definition: ray_amd_host.h). Same six arguments as the reference.
QVector3D ray_color(const Ray& r, const QVector3D& background, const std::vector<std::shared_ptr<shape>>& worldObjects,
                    std::shared_ptr<shape>& lights, int depth, bool drawOnlyColors = false);

inline QVector3D reflect(const QVector3D& v, const QVector3D& n) { return v - 2 * QVector3D::dotProduct(v, n) * n; }   // calc.cpp:163-166
inline QVector3D refract(const QVector3D& uv, const QVector3D& n, double etai_over_etat) {                             // calc.cpp:168-174
    const double cos_theta = std::fmin(double(QVector3D::dotProduct(-uv, n)), 1.0);
    const QVector3D r_out_perp = float(etai_over_etat) * (uv + float(cos_theta) * n);
    const QVector3D r_out_parallel = float(-std::sqrt(std::fabs(1.0 - r_out_perp.lengthSquared()))) * n;
    return r_out_perp + r_out_parallel;
}
inline bool near_zero(const QVector3D& vec) {                                                                           // calc.cpp:156-161
    const double s = 1e-8;
    return (std::fabs(vec[0]) < s) && (std::fabs(vec[1]) < s) && (std::fabs(vec[2]) < s);
}
// calc.cpp:133-154 for ONE pixel on the host (whole frames: rt_tonemap). The reference's NaN scrub is dead code
// (the values are recomputed from `pixel` after it), so NaN channels stay NaN here too.
inline QVector3D rgbPerSamples(const QVector3D& pixel, int samples) {
    const float scale = 1.0f / samples;
    const float r = float(std::sqrt(double(scale * pixel.x()))), g = float(std::sqrt(double(scale * pixel.y()))),
                b = float(std::sqrt(double(scale * pixel.z())));
    return QVector3D(255.0f * std::clamp(r, 0.0f, 0.999f), 255.0f * std::clamp(g, 0.0f, 0.999f), 255.0f * std::clamp(b, 0.0f, 0.999f));
}
}  // namespace calc

#ifdef RAY_AMD_MIRROR_OUTER_IS_CALC_H
#undef RAY_AMD_MIRROR_OUTER_IS_CALC_H
#undef RAY_AMD_MIRROR_OUTER
#include "ray_amd_host.h"
#endif
#endif  // CALC_H
