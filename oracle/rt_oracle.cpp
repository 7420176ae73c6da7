// TEST INFRASTRUCTURE ONLY -- the CPU oracle for the ray path. Never linked, imported or executed
// by the product path (ray_amd/csrc, ray_amd/*.py); only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg load it (oracle/librt_oracle.so, built by oracle/Makefile).
//
// What it is: a Qt-free restatement of the reference's per-pixel path
//     camera::get_ray -> hitFromList -> shape::hit -> calc::ray_color (+ materials, pdfs, RNG)
// following the reference's own structure (recursive ray_color, class hierarchy, call order), with
// the arithmetic the reference has when compiled by g++ on x86-64:
// FP32 QVector3D operators, FP64 for every unqualified <cmath> call, no FMA contraction.
// Each function cites the reference file:line it follows.
//
// Parity status: PINNED. oracle/_ref/ref_harness (the unmodified reference sources compiled here
// against the image's Qt 5.9.7) generated tests/golden/*.npz; tests/test_oracle_golden.py checks
// this file against them bit for bit (RNG streams, QVector3D ops, camera, sphere/rect hits,
// materials/pdfs, primary-hit id maps, seeded radiance sums). Triangles are not in the reference:
// "parity unpinned" for that primitive (analytic known answers only).
//
// Two switches select what the oracle is compared with:
//   rng_mode  0 = RefRng: the reference's three std::minstd_rand0 engines (calc.h:103-108), for
//                 sequence-exact comparison with the reference;
//             1 = CounterRng: one PCG-RXS-M-XS-32 stream per (seed, pixel, sample), the generator
//                 the HIP kernels use (DESIGN.md "RNG contract"), for comparison with the GPU.
//   math_mode 0 = libm cos/sin/pow (what the reference calls);
//             1 = portable cos/sin/pow5 written with IEEE +,-,* only, so the same source gives the
//                 same bits on x86-64 and gfx950 (DESIGN.md "portable transcendentals").
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <thread>
#include <vector>

#include "../include/ray_amd.h"
#include "../include/ray_amd_tri.h"

#if defined(__FMA__)
#error "build the oracle without -mfma / -march=native: the reference arithmetic has no FMA contraction"
#endif

namespace orc {

constexpr double kInf = std::numeric_limits<double>::infinity();   // calc.h:30
constexpr double kPi = 3.1415926535897932385;                        // calc.h:31
constexpr float kSmallest = 0.001f;                                  // calc.h:32

// ------------------------------------------------------------------------------------------------
// QVector3D (Qt5Gui 5.9.7 / 5.15.2): inline operators from qvector3d.h:184-284 are plain FP32 per
// component; length/normalized/dot/cross (qvector3d.cpp) verified against libQt5Gui via vec3.npz.
// ------------------------------------------------------------------------------------------------
struct V3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    V3() {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float lengthSquared() const { return x * x + y * y + z * z; }
    float length() const {
        double len = double(x) * double(x) + double(y) * double(y) + double(z) * double(z);
        return float(std::sqrt(len));
    }
    V3 normalized() const {
        double len = double(x) * double(x) + double(y) * double(y) + double(z) * double(z);
        if (std::fabs(len - 1.0) <= 0.000000000001) return *this;          // qFuzzyIsNull(len - 1.0f)
        if (!(std::fabs(len) <= 0.000000000001)) {
            double s = std::sqrt(len);
            return V3(float(double(x) / s), float(double(y) / s), float(double(z) / s));
        }
        return V3();
    }
};
inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(float f, const V3& v) { return V3(v.x * f, v.y * f, v.z * f); }
inline V3 operator*(const V3& v, float f) { return V3(v.x * f, v.y * f, v.z * f); }
inline V3 operator*(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator-(const V3& v) { return V3(-v.x, -v.y, -v.z); }
inline V3 operator/(const V3& v, float d) { return V3(v.x / d, v.y / d, v.z / d); }
inline float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3& a, const V3& b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// ------------------------------------------------------------------------------------------------
// Portable transcendentals (math_mode 1). Published fdlibm/FreeBSD msun algorithms (k_sin.c,
// k_cos.c, medium-size branch of e_rem_pio2.c) restricted to 0 <= x <= 2*pi + a bit, written with
// separately rounded IEEE double +,-,* only. The HIP kernels carry the same sequence of operations.
// ------------------------------------------------------------------------------------------------
namespace pm {
constexpr double invpio2 = 6.36619772367581382433e-01;
constexpr double pio2_1 = 1.57079632673412561417e+00;    // first 33 bits of pi/2
constexpr double pio2_1t = 6.07710050650619224932e-11;   // pi/2 - pio2_1
constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
inline double ksin(double x, double y) {
    double z = x * x, w = z * z;
    double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
inline double kcos(double x, double y) {
    double z = x * x, w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
inline void sincos(double x, double* s, double* c) {
    int k = int(x * invpio2 + 0.5);
    double fn = double(k);
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double sn = ksin(y0, y1), cs = kcos(y0, y1);
    switch (k & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}
inline double pow5(double x) { double x2 = x * x; double x4 = x2 * x2; return x4 * x; }
// fdlibm e_log.c (__ieee754_log) for positive normal finite x, written with IEEE double +,-,*,/ and integer
// operations on the high word only: log(x) = k*ln2 + log(1+f), 1+f in [sqrt(2)/2, sqrt(2)), log(1+f) from
// s = f/(2+f) and a degree-14 polynomial in s^2. x = 0 gives -inf (what libm returns for log(0)).
constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
inline double log_pos(double x) {
    if (x == 0.0) return -std::numeric_limits<double>::infinity();
    uint64_t bits; std::memcpy(&bits, &x, 8);
    int32_t hx = int32_t(bits >> 32);
    int32_t k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i0 = (hx + 0x95f64) & 0x100000;
    bits = (uint64_t(uint32_t(hx | (i0 ^ 0x3ff00000))) << 32) | (bits & 0xffffffffull);   // normalise x or x/2
    std::memcpy(&x, &bits, 8);
    k += (i0 >> 20);
    const double f = x - 1.0;
    const double dk = double(k);
    if ((0x000fffff & (2 + hx)) < 3) {                  // |f| < 2^-20
        if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    int32_t i = hx - 0x6147a;
    const double w = z * z;
    const int32_t j = 0x6b851 - hx;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    const double R = t2 + t1;
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}
}  // namespace pm

// ------------------------------------------------------------------------------------------------
// RNG. RefRng: calc.h:103-113 + calc.cpp:12-36 with libstdc++ 11 semantics: minstd_rand0 is
// x <- 16807 x mod (2^31 - 1) (seed 0 -> 1); generate_canonical<float,24> draws ONE engine value,
// float(x - 1) / float(2147483646) (bits/random.tcc:3348-3380), nextafter(1,0) if that is >= 1;
// uniform_real_distribution<float>(a,b) = canon * (b - a) + a.
// CounterRng: all three reference streams map to one PCG-RXS-M-XS-32 stream per sample.
// ------------------------------------------------------------------------------------------------
struct Minstd {
    uint32_t x = 1;
    void seed(uint32_t s) { x = s % 2147483647u; if (x == 0) x = 1; }
    uint32_t next() { x = uint32_t((uint64_t(x) * 16807ull) % 2147483647ull); return x; }
    float canon() {
        float r = float(next() - 1u) / 2147483648.0f;   // float(2147483646) rounds to 2^31
        if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
        return r;
    }
};

inline uint32_t pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}

struct Rng {
    int mode = 0;             // 0 RefRng, 1 CounterRng
    Minstd gen, gen01, gen11; // calc.h:104,106,108
    uint32_t state = 0;       // CounterRng
    uint64_t seed = 0;
    void seed_ref(uint32_t a, uint32_t b, uint32_t c) { mode = 0; gen.seed(a); gen01.seed(b); gen11.seed(c); }
    void key(uint64_t sd, uint32_t pixel, uint32_t sample) {
        mode = 1; seed = sd;
        uint32_t lo = uint32_t(sd), hi = uint32_t(sd >> 32);
        state = pcg_hash(pixel + pcg_hash(sample + pcg_hash(lo ^ pcg_hash(hi))));
    }
    float counter_canon() {
        state = state * 747796405u + 2891336453u;
        uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
        uint32_t bits = (word >> 22) ^ word;
        return float(bits >> 8) * 5.9604644775390625e-08f;   // 2^-24
    }
    float canon(Minstd& g) { return mode == 0 ? g.canon() : counter_canon(); }
    float random_double(float mn, float mx) { return canon(gen) * (mx - mn) + mn; }    // calc.cpp:12-16
    float random_double01() { return canon(gen01) * (1.0f - 0.0f) + 0.0f; }            // calc.cpp:28-31
    float random_double11() { return canon(gen11) * (1.0f - -1.0f) + -1.0f; }          // calc.cpp:33-36
};

struct Ctx {
    Rng rng;
    int math_mode = 0;
    uint64_t rays = 0;   // closest-hit scans executed by ray_color (primary + bounces)
    void sincos(double x, double* s, double* c) const {
        if (math_mode == 0) { *s = std::sin(x); *c = std::cos(x); } else pm::sincos(x, s, c);
    }
    double pow5(double x) const { return math_mode == 0 ? std::pow(x, 5) : pm::pow5(x); }
    double log(double x) const { return math_mode == 0 ? std::log(x) : pm::log_pos(x); }
};

// calc.cpp:43-46: brace-init list, evaluated left to right.
inline V3 randomVec11(Ctx& c) {
    float x = c.rng.random_double11(); float y = c.rng.random_double11(); float z = c.rng.random_double11();
    return V3(x, y, z);
}
// calc.cpp:54-62
inline V3 random_in_unit_sphere(Ctx& c) {
    V3 p(-1.0f, -1.0f, -1.0f);
    while (p.lengthSquared() >= 1.0f) p = randomVec11(c);
    return p;
}
// calc.cpp:64-72. g++ evaluates the arguments of the parenthesised constructor call right to
// left (pinned by camera.npz / rng.npz): the first draw becomes y.
inline V3 random_in_unit_disk(Ctx& c) {
    for (;;) {
        float y = c.rng.random_double11();
        float x = c.rng.random_double11();
        V3 p(x, y, 0.0f);
        if (!(p.lengthSquared() >= 1.0f)) return p;
    }
}
// calc.h:62-73
inline V3 random_cosine_direction(Ctx& c) {
    float r1 = c.rng.random_double01();
    float r2 = c.rng.random_double01();
    double z = std::sqrt(double(1 - r2));
    double phi = 2 * kPi * r1;
    double sn, cs;
    c.sincos(phi, &sn, &cs);
    double x = cs * std::sqrt(double(r2));
    double y = sn * std::sqrt(double(r2));
    return V3(float(x), float(y), float(z));
}
// calc.h:75-86
inline V3 random_to_sphere(Ctx& c, double radius, double distance_squared) {
    float r1 = c.rng.random_double01();
    float r2 = c.rng.random_double01();
    double z = 1 + r2 * (std::sqrt(1 - radius * radius / distance_squared) - 1);
    double phi = 2 * kPi * r1;
    double sn, cs;
    c.sincos(phi, &sn, &cs);
    double x = cs * std::sqrt(1 - z * z);
    double y = sn * std::sqrt(1 - z * z);
    return V3(float(x), float(y), float(z));
}
// calc.cpp:163-166
inline V3 reflect(const V3& v, const V3& n) { return v - 2 * dot(v, n) * n; }
// calc.cpp:168-174
inline V3 refract(const V3& uv, const V3& n, double etai_over_etat) {
    double cos_theta = std::fmin(double(dot(-uv, n)), 1.0);
    V3 r_out_perp = float(etai_over_etat) * (uv + float(cos_theta) * n);
    V3 r_out_parallel = float(-std::sqrt(std::fabs(1.0 - r_out_perp.lengthSquared()))) * n;
    return r_out_perp + r_out_parallel;
}
// calc.cpp:133-154 (the NaN scrub there is dead code: values are recomputed from `pixel`)
inline V3 rgbPerSamples(const V3& pixel, int samples) {
    float scale = 1.0f / samples;
    float r = float(std::sqrt(double(scale * pixel.x)));
    float g = float(std::sqrt(double(scale * pixel.y)));
    float b = float(std::sqrt(double(scale * pixel.z)));
    auto clamp = [](float v) { return (v < 0.0f) ? 0.0f : (0.999f < v) ? 0.999f : v; };
    return V3(255.0f * clamp(r), 255.0f * clamp(g), 255.0f * clamp(b));
}

// Ray.h:6-16
struct Ray {
    V3 o, d;
    float dl2 = 0.0f;
    Ray() {}
    Ray(const V3& origin, const V3& dir) : o(origin), d(dir), dl2(dir.lengthSquared()) {}
    V3 at(double t) const { return o + float(t) * d; }
};

struct Material;
// hitPosition.h:95-109
struct HitRecord {
    V3 p, normal;
    const Material* mat = nullptr;
    double t = 0, u = 0, v = 0;
    bool front_face = false;
    int prim = -1;   // oracle bookkeeping only: flattened primitive index
    void setFaceNormal(const Ray& r, const V3& outward) {
        front_face = dot(r.d, outward) < 0.0f;
        normal = front_face ? outward : -outward;
    }
};

// hitPosition.h:112-139
struct Onb {
    V3 axis[3];
    void build_from_w(const V3& n) {
        axis[2] = n.normalized();
        V3 a = (std::fabs(double(axis[2].x)) > 0.9) ? V3(0, 1, 0) : V3(1, 0, 0);
        axis[1] = cross(axis[2], a).normalized();
        axis[0] = cross(axis[2], axis[1]);
    }
    V3 local(const V3& a) const { return a.x * axis[0] + a.y * axis[1] + a.z * axis[2]; }
};

// ------------------------------------------------------------------------------------------------
// shapes (shapes.h)
// ------------------------------------------------------------------------------------------------
struct Shape {
    int prim = -1;
    const Material* mat = nullptr;
    virtual ~Shape() {}
    // c: the RNG / math context; only constant_medium::hit draws from it (shapes.h:425)
    virtual std::optional<HitRecord> hit(Ctx& c, const Ray& r, double t_min, double t_max) const = 0;
    virtual double pdf_value(Ctx&, const V3&, const V3&) const { return 0.0; }   // shapes.h:56-58
    virtual V3 random(Ctx&, const V3&) const { return V3(1, 0, 0); }       // shapes.h:60-62
};

// shapes.h:67-159 (get_sphere_uv is dead for solid colours but kept: the fixtures carry u, v)
struct Sphere : Shape {
    V3 center; float radius, radiusSquared;
    Sphere(V3 c, float r) : center(c), radius(r), radiusSquared(r * r) {}
    std::optional<HitRecord> hit(Ctx&, const Ray& ray, double t_min, double t_max) const override {
        V3 oc = ray.o - center;
        float a = ray.dl2;
        float b = dot(oc, ray.d);
        float c = oc.lengthSquared() - radiusSquared;
        float d = b * b - a * c;
        if (d < 0.0) return {};
        double sqrtd = std::sqrt(double(d));
        double root = (-b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-b + sqrtd) / a;
            if (root < t_min || t_max < root) return {};
        }
        V3 at = ray.at(root);
        V3 outward = (at - center) / radius;
        HitRecord rec;
        rec.t = root; rec.p = at; rec.mat = mat; rec.prim = prim;
        double theta = std::acos(double(-outward.y));
        double phi = std::atan2(double(-outward.z), double(outward.x)) + kPi;
        rec.u = phi / (2.0f * kPi);
        rec.v = theta / kPi;
        rec.setFaceNormal(ray, outward);
        return rec;
    }
    double pdf_value(Ctx& c, const V3& o, const V3& v) const override {   // shapes.h:117-127
        auto h = this->hit(c, Ray(o, v), 0.001, kInf);
        if (!h.has_value()) return 0.0;
        double cos_theta_max = std::sqrt(double(1 - radius * radius / (center - o).lengthSquared()));
        double solid_angle = 2 * kPi * (1 - cos_theta_max);
        return 1 / solid_angle;
    }
    V3 random(Ctx& c, const V3& o) const override {                // shapes.h:129-135
        V3 direction = center - o;
        float distance_squared = direction.lengthSquared();
        Onb uvw;
        uvw.build_from_w(direction);
        return uvw.local(random_to_sphere(c, radius, distance_squared));
    }
};

// shapes.h:161-337: axis 0 = rectangle (XY, plane z=k), 1 = xz_rect (y=k), 2 = yz_rect (x=k).
// k and the bounds are doubles, so t and the in-plane coordinates are computed in double.
struct Rect : Shape {
    int axis; double a0, a1, b0, b1, k;
    Rect(int ax, double A0, double A1, double B0, double B1, double K) : axis(ax), a0(A0), a1(A1), b0(B0), b1(B1), k(K) {}
    std::optional<HitRecord> hit(Ctx&, const Ray& r, double t_min, double t_max) const override {
        float ok, dk, oa, da, ob, db;
        if (axis == 0) { ok = r.o.z; dk = r.d.z; oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
        else if (axis == 1) { ok = r.o.y; dk = r.d.y; oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
        else { ok = r.o.x; dk = r.d.x; oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
        double t = (k - ok) / dk;
        if (t < t_min || t > t_max) return {};
        double a = oa + t * da;
        double b = ob + t * db;
        if (a < a0 || a > a1 || b < b0 || b > b1) return {};
        HitRecord rec;
        rec.u = (a - a0) / (a1 - a0);
        rec.v = (b - b0) / (b1 - b0);
        rec.t = t;
        V3 outward = axis == 0 ? V3(0, 0, 1) : (axis == 1 ? V3(0, 1, 0) : V3(1, 0, 0));   // shapes.h:8-12
        rec.setFaceNormal(r, outward);
        rec.mat = mat; rec.prim = prim;
        rec.p = r.at(t);
        return rec;
    }
    // shapes.h:202-214, 268-279; yz_rect has neither (falls to the base class, quirk Q3)
    double pdf_value(Ctx& c, const V3& origin, const V3& v) const override {
        if (axis == 2) return 0.0;
        auto h = this->hit(c, Ray(origin, v), 0.001, kInf);
        if (!h.has_value()) return 0.0;
        double area = (a1 - a0) * (b1 - b0);
        double distance_squared = h->t * h->t * v.lengthSquared();
        double cosine = std::fabs(double(dot(v, h->normal) / v.length()));
        return distance_squared / (cosine * area);
    }
    // shapes.h:216-219, 281-284: QVector3D(random_double(a0,a1), k, random_double(b0,b1)) for BOTH
    // the XY and the XZ rect (quirk Q2: the XY rect puts k in the y slot); g++ evaluates the
    // constructor arguments right to left, so the (b0,b1) draw comes first.
    V3 random(Ctx& c, const V3& origin) const override {
        if (axis == 2) return V3(1, 0, 0);
        float zc = c.rng.random_double(float(b0), float(b1));
        float xc = c.rng.random_double(float(a0), float(a1));
        V3 random_point(xc, float(k), zc);
        return random_point - origin;
    }
};

// Not in the reference (BASELINE.json configs[3]): Moeller-Trumbore in FP32 with separately
// rounded operations plus containment of the hit point in the triangle's bounding sphere
// (include/ray_amd_tri.h), build-defined; "parity unpinned". e1, e2, the unit normal and the
// bounding sphere are derived once per triangle. Comparisons are written so that NaN means miss.
struct Triangle : Shape {
    V3 v0, e1, e2, n, cb;
    float rb2;
    Triangle(V3 a, V3 b, V3 c) : v0(a), e1(b - a), e2(c - a) {
        n = cross(e1, e2).normalized();
        const float fa[3] = {a.x, a.y, a.z}, fb[3] = {b.x, b.y, b.z}, fc[3] = {c.x, c.y, c.z};
        float ctr[3];
        rt_triangle_bound(fa, fb, fc, ctr, &rb2);
        cb = V3(ctr[0], ctr[1], ctr[2]);
    }
    std::optional<HitRecord> hit(Ctx&, const Ray& r, double t_min, double t_max) const override {
        V3 pvec = cross(r.d, e2);
        float det = dot(e1, pvec);
        if (!(det != 0.0f)) return {};
        float inv = 1.0f / det;
        V3 tvec = r.o - v0;
        float u = dot(tvec, pvec) * inv;
        if (!(u >= 0.0f && u <= 1.0f)) return {};
        V3 qvec = cross(tvec, e1);
        float v = dot(r.d, qvec) * inv;
        if (!(v >= 0.0f && u + v <= 1.0f)) return {};
        double t = double(dot(e2, qvec) * inv);
        V3 dp = r.at(t) - cb;
        if (!(dp.lengthSquared() <= rb2)) return {};
        if (!(t >= t_min && t <= t_max)) return {};
        HitRecord rec;
        rec.t = t; rec.u = u; rec.v = v; rec.mat = mat; rec.prim = prim;
        rec.setFaceNormal(r, n);
        rec.p = r.at(t);
        return rec;
    }
};

// hitPosition.cpp:6-22
inline std::optional<HitRecord> hitFromList(Ctx& c, const std::vector<std::unique_ptr<Shape>>& list, const Ray& ray,
                                            double t_min, double t_max) {
    double closest_so_far = t_max;
    std::optional<HitRecord> ret;
    for (const auto& object : list) {
        auto h = object->hit(c, ray, t_min, closest_so_far);
        if (h.has_value()) { closest_so_far = h->t; ret = h; }
    }
    return ret;
}

// shapes.h:339-372: six rects, hit = hitFromList over them (only ever a constant_medium's boundary here: boxes in
// the world are flattened into their sides by the scene description, with the same result)
struct Box : Shape {
    std::vector<std::unique_ptr<Shape>> sides;
    Box(V3 p0, V3 p1) {
        sides.push_back(std::make_unique<Rect>(0, p0.x, p1.x, p0.y, p1.y, p1.z));
        sides.push_back(std::make_unique<Rect>(0, p0.x, p1.x, p0.y, p1.y, p0.z));
        sides.push_back(std::make_unique<Rect>(1, p0.x, p1.x, p0.z, p1.z, p1.y));
        sides.push_back(std::make_unique<Rect>(1, p0.x, p1.x, p0.z, p1.z, p0.y));
        sides.push_back(std::make_unique<Rect>(2, p0.y, p1.y, p0.z, p1.z, p1.x));
        sides.push_back(std::make_unique<Rect>(2, p0.y, p1.y, p0.z, p1.z, p0.x));
    }
    std::optional<HitRecord> hit(Ctx& c, const Ray& r, double t_min, double t_max) const override {
        return hitFromList(c, sides, r, t_min, t_max);
    }
};

// shapes.h:374-455. Every ray whose clamped boundary interval is non-empty draws ONE number from gen01, inside
// hitFromList's loop; the hit record has normal (1,0,0) and front_face true ("arbitrary", shapes.h:439-440).
struct ConstantMedium : Shape {
    std::unique_ptr<Shape> boundary;
    double neg_inv_density;
    ConstantMedium(std::unique_ptr<Shape> b, double niv) : boundary(std::move(b)), neg_inv_density(niv) {}
    std::optional<HitRecord> hit(Ctx& c, const Ray& r, double t_min, double t_max) const override {
        auto h1 = boundary->hit(c, r, -kInf, kInf);
        if (!h1.has_value()) return {};
        auto h2 = boundary->hit(c, r, h1->t + 0.0001, kInf);
        if (!h2.has_value()) return {};
        double t1 = h1->t, t2 = h2->t;
        if (t1 < t_min) t1 = t_min;
        if (t2 > t_max) t2 = t_max;
        if (t1 >= t2) return {};
        if (t1 < 0) t1 = 0;
        const float ray_length = r.d.length();
        const double distance_inside_boundary = (t2 - t1) * ray_length;
        const double hit_distance = neg_inv_density * c.log(double(c.rng.random_double01()));
        if (hit_distance > distance_inside_boundary) return {};
        HitRecord rec;
        rec.t = t1 + hit_distance / ray_length;
        rec.p = r.at(rec.t);
        rec.normal = V3(1, 0, 0);
        rec.front_face = true;
        rec.mat = mat; rec.prim = prim;
        return rec;
    }
};

// ------------------------------------------------------------------------------------------------
// pdfs (hitPosition.h:141-185, hitPosition.cpp:55-72)
// ------------------------------------------------------------------------------------------------
struct Pdf {
    virtual ~Pdf() {}
    virtual double value(Ctx& c, const V3& direction) const = 0;
    virtual V3 generate(Ctx& c) const = 0;
};
struct CosinePdf : Pdf {
    Onb uvw;
    explicit CosinePdf(const V3& w) { uvw.build_from_w(w); }
    double value(Ctx&, const V3& direction) const override {
        float cosine = dot(direction.normalized(), uvw.axis[2]);
        return (cosine <= 0) ? 0 : cosine / kPi;
    }
    V3 generate(Ctx& c) const override { return uvw.local(random_cosine_direction(c)); }
};
struct HittablePdf : Pdf {
    const Shape* ptr = nullptr; V3 o;
    HittablePdf() {}
    HittablePdf(const Shape* p, const V3& origin) : ptr(p), o(origin) {}
    double value(Ctx& c, const V3& direction) const override { return ptr->pdf_value(c, o, direction); }
    V3 generate(Ctx& c) const override { return ptr->random(c, o); }
};
struct MixturePdf : Pdf {
    const Pdf* p[2];
    MixturePdf(const Pdf* p0, const Pdf* p1) { p[0] = p0; p[1] = p1; }
    double value(Ctx& c, const V3& direction) const override { return 0.5 * p[0]->value(c, direction) + 0.5 * p[1]->value(c, direction); }
    V3 generate(Ctx& c) const override {
        if (c.rng.random_double01() < 0.5) return p[0]->generate(c);
        return p[1]->generate(c);
    }
};

// ------------------------------------------------------------------------------------------------
// materials (hitPosition.h:72-93, 197-314)
// ------------------------------------------------------------------------------------------------
struct ScatterRecord {
    Ray specular_ray; bool is_specular = false; V3 attenuation; std::unique_ptr<Pdf> pdf_ptr;
};
struct Material {
    virtual ~Material() {}
    virtual bool scatter(Ctx&, const Ray&, const HitRecord&, ScatterRecord&) const { return false; }
    virtual double scattering_pdf(const Ray&, const HitRecord&, const Ray&) const { return 0; }
    virtual V3 emitted() const { return V3(0, 0, 0); }
};
struct Lambertian : Material {
    V3 albedo;
    explicit Lambertian(V3 a) : albedo(a) {}
    bool scatter(Ctx&, const Ray&, const HitRecord& rec, ScatterRecord& s) const override {
        s.is_specular = false;
        s.attenuation = albedo;
        s.pdf_ptr = std::make_unique<CosinePdf>(rec.normal);
        return true;
    }
    double scattering_pdf(const Ray&, const HitRecord& rec, const Ray& scattered) const override {
        float cosine = dot(rec.normal, scattered.d.normalized());
        return cosine < 0 ? 0 : cosine / kPi;
    }
};
struct Metal : Material {
    V3 albedo; float fuzz;
    Metal(V3 a, float f) : albedo(a), fuzz(f) {}
    bool scatter(Ctx& c, const Ray& r_in, const HitRecord& rec, ScatterRecord& s) const override {
        V3 reflected = reflect(r_in.d.normalized(), rec.normal);
        s.specular_ray = Ray(rec.p, reflected + fuzz * random_in_unit_sphere(c));
        s.attenuation = albedo;
        s.is_specular = true;
        s.pdf_ptr.reset();
        return true;
    }
};
struct Dielectric : Material {
    double ir;
    explicit Dielectric(double i) : ir(i) {}
    static float reflectance(const Ctx& c, float cosine, float ref_idx) {      // hitPosition.h:281-287
        float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
        r0 = r0 * r0;
        return r0 + (1.0f - r0) * float(c.pow5(double(1.0f - cosine)));
    }
    bool scatter(Ctx& c, const Ray& r_in, const HitRecord& rec, ScatterRecord& s) const override {
        s.is_specular = true;
        s.pdf_ptr.reset();
        s.attenuation = V3(1.0f, 1.0f, 1.0f);
        double refraction_ratio = rec.front_face ? (1.0f / ir) : ir;
        V3 unit_direction = r_in.d.normalized();
        float cos_theta = float(std::fmin(double(dot(-unit_direction, rec.normal)), double(1.0f)));
        float sin_theta = float(std::sqrt(double(1.0f - cos_theta * cos_theta)));
        bool cannot_refract = refraction_ratio * sin_theta > 1.0f;
        V3 direction;
        if (cannot_refract || reflectance(c, cos_theta, float(refraction_ratio)) > c.rng.random_double01())
            direction = reflect(unit_direction, rec.normal);
        else
            direction = refract(unit_direction, rec.normal, refraction_ratio);
        s.specular_ray = Ray(rec.p, direction);
        return true;
    }
};
// hitPosition.h:316-337. The reference's scatter leaves scatter_record::is_specular (and pdf_ptr) unset, so
// calc.cpp:118 reads an indeterminate bool: undefined behaviour. The one reading under which ray_color is defined
// (a null pdf_ptr cannot be used by the mixture branch) is "specular", which is also what the book this code derives
// from does for its isotropic phase function; that reading is adopted here and documented as a deviation.
struct Isotropic : Material {
    V3 albedo;
    explicit Isotropic(V3 a) : albedo(a) {}
    bool scatter(Ctx& c, const Ray&, const HitRecord& rec, ScatterRecord& s) const override {
        s.specular_ray = Ray(rec.p, random_in_unit_sphere(c));
        s.attenuation = albedo;
        s.is_specular = true;
        s.pdf_ptr.reset();
        return true;
    }
};
struct DiffuseLight : Material {
    V3 emit;
    explicit DiffuseLight(V3 c) : emit(c) {}
    V3 emitted() const override { return emit; }
};

// camera.h:7-64
struct Camera {
    V3 origin, lower_left_corner, horizontal, vertical, w, u, v;
    double lens_radius = 0;
    Camera() {}
    Camera(V3 lookfrom, V3 lookat, V3 vup, double focus_dist, double vfov, double aspect_ratio, double aperture) {
        double theta = vfov * kPi / 180.0;     // calc.h:36-39
        double h = std::tan(theta / 2.0);
        double viewport_height = 2.0 * h;
        double viewport_width = aspect_ratio * viewport_height;
        w = (lookfrom - lookat).normalized();
        u = cross(vup, w).normalized();
        v = cross(w, u);
        origin = lookfrom;
        horizontal = float(focus_dist * viewport_width) * u;
        vertical = float(focus_dist * viewport_height) * v;
        lower_left_corner = origin - horizontal / 2.0f - vertical / 2.0f - float(focus_dist) * w;
        lens_radius = aperture / 2.0f;
    }
    explicit Camera(const RtCamera& c) {
        auto v3 = [](const float* p) { return V3(p[0], p[1], p[2]); };
        origin = v3(c.origin); lower_left_corner = v3(c.lower_left_corner); horizontal = v3(c.horizontal);
        vertical = v3(c.vertical); u = v3(c.u); v = v3(c.v); w = v3(c.w); lens_radius = c.lens_radius;
    }
    Ray get_ray(Ctx& c, double s, double t) const {
        V3 rd = float(lens_radius) * random_in_unit_disk(c);
        V3 offset = u * rd.x + v * rd.y;
        return Ray(origin + offset, lower_left_corner + float(s) * horizontal + float(t) * vertical - origin - offset);
    }
};

// ------------------------------------------------------------------------------------------------
// scene assembled from the flattened RtScene arrays (include/ray_amd.h)
// ------------------------------------------------------------------------------------------------
struct World {
    std::vector<std::unique_ptr<Material>> mats;
    std::vector<std::unique_ptr<Shape>> list;
    std::unique_ptr<Shape> light;
    V3 background;
};

inline std::unique_ptr<Shape> make_shape(int kind, const double* d) {
    switch (kind) {
    case RT_PRIM_SPHERE: return std::make_unique<Sphere>(V3(float(d[0]), float(d[1]), float(d[2])), float(d[3]));
    case RT_PRIM_XY_RECT: return std::make_unique<Rect>(0, d[0], d[1], d[2], d[3], d[4]);
    case RT_PRIM_XZ_RECT: return std::make_unique<Rect>(1, d[0], d[1], d[2], d[3], d[4]);
    case RT_PRIM_YZ_RECT: return std::make_unique<Rect>(2, d[0], d[1], d[2], d[3], d[4]);
    case RT_PRIM_TRIANGLE:
        return std::make_unique<Triangle>(V3(float(d[0]), float(d[1]), float(d[2])), V3(float(d[3]), float(d[4]), float(d[5])),
                                          V3(float(d[6]), float(d[7]), float(d[8])));
    case RT_PRIM_MEDIUM_SPHERE:
        return std::make_unique<ConstantMedium>(std::make_unique<Sphere>(V3(float(d[0]), float(d[1]), float(d[2])), float(d[3])), d[4]);
    case RT_PRIM_MEDIUM_BOX:
        return std::make_unique<ConstantMedium>(std::make_unique<Box>(V3(float(d[0]), float(d[1]), float(d[2])), V3(float(d[3]), float(d[4]), float(d[5]))), d[6]);
    }
    return nullptr;
}

inline bool build_world(const RtScene* sc, World& w) {
    for (int i = 0; i < sc->n_mats; ++i) {
        V3 c(sc->mat_rgb[3 * i], sc->mat_rgb[3 * i + 1], sc->mat_rgb[3 * i + 2]);
        switch (sc->mat_kind[i]) {
        case RT_MAT_LAMBERTIAN: w.mats.push_back(std::make_unique<Lambertian>(c)); break;
        case RT_MAT_METAL: w.mats.push_back(std::make_unique<Metal>(c, sc->mat_fuzz[i])); break;
        case RT_MAT_DIELECTRIC: w.mats.push_back(std::make_unique<Dielectric>(sc->mat_ir[i])); break;
        case RT_MAT_DIFFUSE_LIGHT: w.mats.push_back(std::make_unique<DiffuseLight>(c)); break;
        case RT_MAT_ISOTROPIC: w.mats.push_back(std::make_unique<Isotropic>(c)); break;
        default: return false;
        }
    }
    for (int i = 0; i < sc->n_prims; ++i) {
        auto s = make_shape(sc->prim_kind[i], sc->prim_data + size_t(i) * RT_PRIM_STRIDE);
        if (!s || sc->prim_mat[i] < 0 || sc->prim_mat[i] >= sc->n_mats) return false;
        s->prim = i;
        s->mat = w.mats[sc->prim_mat[i]].get();
        w.list.push_back(std::move(s));
    }
    if (sc->light_kind == RT_LIGHT_BASE) w.light = std::make_unique<Rect>(2, 0, 1, 0, 1, 0);   // base-class behaviour
    else if (sc->light_kind == RT_PRIM_SPHERE || sc->light_kind == RT_PRIM_XY_RECT || sc->light_kind == RT_PRIM_XZ_RECT) w.light = make_shape(sc->light_kind, sc->light_data);
    if (!w.light) return false;
    w.background = V3(sc->background[0], sc->background[1], sc->background[2]);
    return true;
}

// calc.cpp:90-131. m_light_ptr (calc.cpp:9) is a file-scope object there; here it lives on the
// stack of each call: g++ evaluates p.value() before the recursive call, so
// the clobbering of the global by deeper bounces is never observed.
inline V3 ray_color(Ctx& c, const Ray& r, const World& w, int depth, bool drawOnlyColors = false) {
    if (depth <= 0.0f) return V3(0, 0, 0);
    ++c.rays;
    auto hr = hitFromList(c, w.list, r, kSmallest, kInf);
    if (!hr.has_value()) return w.background;
    if (drawOnlyColors) {
        V3 N = (r.at(hr->t) - V3(0.0f, 0.0f, -1.0f)).normalized();
        return 0.5f * V3(N.x + 1.0f, N.y + 1.0f, N.z + 1.0f);
    }
    const HitRecord& rec = *hr;
    ScatterRecord srec;
    V3 emitted = rec.mat->emitted();
    if (!rec.mat->scatter(c, r, rec, srec)) return emitted;
    if (srec.is_specular) return srec.attenuation * ray_color(c, srec.specular_ray, w, depth - 1);
    HittablePdf light_pdf(w.light.get(), rec.p);
    MixturePdf p(&light_pdf, srec.pdf_ptr.get());
    Ray scattered(rec.p, p.generate(c));
    double spdf = rec.mat->scattering_pdf(r, rec, scattered);
    double pval = p.value(c, scattered.d);
    V3 li = ray_color(c, scattered, w, depth - 1);
    return emitted + srec.attenuation * float(spdf) * li / float(pval);
}

}  // namespace orc

// =================================================================================================
// C entry points for tests/ (ctypes). Array arguments are caller-owned host memory.
// =================================================================================================
using namespace orc;

extern "C" {

// rng.npz: which = 0 random_double01 (gen01), 1 random_double11 (gen11), 2 random_double(a,b) (gen)
void orc_refrng_fill(uint32_t seed, int which, float a, float b, int skip, int n, float* out) {
    Ctx c; c.rng.seed_ref(seed, seed, seed);
    for (int i = 0; i < skip; ++i) (which == 0 ? c.rng.gen01 : which == 1 ? c.rng.gen11 : c.rng.gen).next();
    for (int i = 0; i < n; ++i)
        out[i] = which == 0 ? c.rng.random_double01() : which == 1 ? c.rng.random_double11() : c.rng.random_double(a, b);
}
// CounterRng raw canonical floats for the stream (seed, pixel, sample)
void orc_counter_fill(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float* out) {
    Ctx c; c.rng.key(seed, pixel, sample);
    for (int i = 0; i < n; ++i) out[i] = c.rng.counter_canon();
}
// rng.npz derived samplers: which = 0 in_unit_sphere, 1 in_unit_disk, 2 cosine_direction,
// 3 to_sphere(radius, d2_base + d2_step * i); all four are run back to back on one seeding, as the
// harness does, so pass n for each and the outputs in order.
void orc_samplers(uint32_t sg, uint32_t s01, uint32_t s11, int math_mode, int n, double radius, double d2_base,
                  double d2_step, float* sph, float* disk, float* cosd, float* tosph) {
    Ctx c; c.rng.seed_ref(sg, s01, s11); c.math_mode = math_mode;
    auto put = [](float* o, int i, const V3& v) { o[3 * i] = v.x; o[3 * i + 1] = v.y; o[3 * i + 2] = v.z; };
    for (int i = 0; i < n; ++i) put(sph, i, random_in_unit_sphere(c));
    for (int i = 0; i < n; ++i) put(disk, i, random_in_unit_disk(c));
    for (int i = 0; i < n; ++i) put(cosd, i, random_cosine_direction(c));
    for (int i = 0; i < n; ++i) put(tosph, i, random_to_sphere(c, radius, d2_base + d2_step * i));
}
void orc_sincos(int math_mode, int n, const double* x, double* s, double* cs) {
    Ctx c; c.math_mode = math_mode;
    for (int i = 0; i < n; ++i) c.sincos(x[i], &s[i], &cs[i]);
}
void orc_pow5(int math_mode, int n, const double* x, double* y) {
    Ctx c; c.math_mode = math_mode;
    for (int i = 0; i < n; ++i) y[i] = c.pow5(x[i]);
}

static void put_rec(const std::optional<HitRecord>& h, int i, uint8_t* has, double* t, float* p, float* n, uint8_t* ff, double* uv);
void orc_log(int math_mode, int n, const double* x, double* y) {
    Ctx c; c.math_mode = math_mode;
    for (int i = 0; i < n; ++i) y[i] = c.log(x[i]);
}
// medium.npz: the call sequence of ref_harness.cpp cmd_medium, RefRng seeded (31,32,33) once; kind 0 = sphere
// boundary (bound = cx cy cz r), 1 = box boundary (bound = p0 p1)
void orc_medium_kat(int n, int math_mode, const int32_t* kind, const float* bound, const double* density, const float* ro, const float* rd,
                    const double* t_min, const double* t_max, uint8_t* has, double* t, float* p, float* nrm, uint8_t* ff) {
    Ctx c; c.rng.seed_ref(31, 32, 33); c.math_mode = math_mode;
    for (int i = 0; i < n; ++i) {
        const float* b = bound + 6 * i;
        std::unique_ptr<Shape> bs;
        if (kind[i] == 0) bs = std::make_unique<Sphere>(V3(b[0], b[1], b[2]), b[3]);
        else bs = std::make_unique<Box>(V3(b[0], b[1], b[2]), V3(b[3], b[4], b[5]));
        ConstantMedium m(std::move(bs), -1 / density[i]);
        Ray r(V3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), V3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]));
        put_rec(m.hit(c, r, t_min[i], t_max[i]), i, has, t, p, nrm, ff, nullptr);
    }
}

void orc_vec3(int n, const float* a, const float* b, float* lsq, float* len, float* nrm, float* dt, float* crs) {
    for (int i = 0; i < n; ++i) {
        V3 va(a[3 * i], a[3 * i + 1], a[3 * i + 2]), vb(b[3 * i], b[3 * i + 1], b[3 * i + 2]);
        lsq[i] = va.lengthSquared(); len[i] = va.length();
        V3 nn = va.normalized(); nrm[3 * i] = nn.x; nrm[3 * i + 1] = nn.y; nrm[3 * i + 2] = nn.z;
        dt[i] = dot(va, vb);
        V3 cr = cross(va, vb); crs[3 * i] = cr.x; crs[3 * i + 1] = cr.y; crs[3 * i + 2] = cr.z;
    }
}

static void cam_to_c(const Camera& cam, RtCamera* out) {
    auto put = [](float* o, const V3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; };
    put(out->origin, cam.origin); put(out->lower_left_corner, cam.lower_left_corner);
    put(out->horizontal, cam.horizontal); put(out->vertical, cam.vertical);
    put(out->u, cam.u); put(out->v, cam.v); put(out->w, cam.w);
    out->lens_radius = cam.lens_radius;
}
void orc_camera(const float* from, const float* at, const float* up, double focus, double vfov, double aspect,
                double aperture, RtCamera* out) {
    Camera cam(V3(from[0], from[1], from[2]), V3(at[0], at[1], at[2]), V3(up[0], up[1], up[2]), focus, vfov, aspect, aperture);
    cam_to_c(cam, out);
}
// camera.npz: n seeded get_ray calls (RefRng seeds sg, s01, s11), st = [n][2]
void orc_camera_rays(const RtCamera* cc, uint32_t sg, uint32_t s01, uint32_t s11, int n, const float* st, float* ro,
                     float* rd, float* dl2) {
    Camera cam(*cc);
    Ctx c; c.rng.seed_ref(sg, s01, s11);
    for (int i = 0; i < n; ++i) {
        Ray r = cam.get_ray(c, st[2 * i], st[2 * i + 1]);
        ro[3 * i] = r.o.x; ro[3 * i + 1] = r.o.y; ro[3 * i + 2] = r.o.z;
        rd[3 * i] = r.d.x; rd[3 * i + 1] = r.d.y; rd[3 * i + 2] = r.d.z;
        dl2[i] = r.dl2;
    }
}

static void put_rec(const std::optional<HitRecord>& h, int i, uint8_t* has, double* t, float* p, float* n, uint8_t* ff,
                    double* uv) {
    has[i] = h.has_value();
    if (h.has_value()) {
        t[i] = h->t; p[3 * i] = h->p.x; p[3 * i + 1] = h->p.y; p[3 * i + 2] = h->p.z;
        n[3 * i] = h->normal.x; n[3 * i + 1] = h->normal.y; n[3 * i + 2] = h->normal.z;
        ff[i] = h->front_face; if (uv) { uv[2 * i] = h->u; uv[2 * i + 1] = h->v; }
    } else {
        t[i] = 0; for (int k = 0; k < 3; ++k) { p[3 * i + k] = 0; n[3 * i + k] = 0; }
        ff[i] = 0; if (uv) { uv[2 * i] = 0; uv[2 * i + 1] = 0; }
    }
}
// shapehits.npz; kind per item: RT_PRIM_*; prm = [n][RT_PRIM_STRIDE] doubles
void orc_shape_hits(int n, const int32_t* kind, const double* prm, const float* ro, const float* rd, double t_min,
                    const double* t_max, uint8_t* has, double* t, float* p, float* nrm, uint8_t* ff, double* uv) {
    for (int i = 0; i < n; ++i) {
        auto s = make_shape(kind[i], prm + size_t(i) * RT_PRIM_STRIDE);
        Ray r(V3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), V3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]));
        Ctx c;
        put_rec(s->hit(c, r, t_min, t_max[i]), i, has, t, p, nrm, ff, uv);
    }
}

// materials.npz: the call sequence of ref_harness.cpp cmd_materials, RNG seeded (21,22,23) once.
struct OrcMatKat {
    int n; int math_mode;
    const float *in_d, *in_n, *in_p; const uint8_t* in_ff; const float* prm; const double* ir; const float* lamb_sd;
    const float* xz_o;   // [n][3] origins used for the xz light
    float *metal_o, *metal_d, *metal_att, *diel_d, *diel_att, *lamb_att;
    double *lamb_spdf, *cos_val; float* cos_gen;
    float* sph_gen; double* sph_val; float* xz_gen; double* xz_val; float* xy_gen; double* xy_val;
    float* yz_gen; double* yz_val; float* mix_gen; double* mix_val;
    float *refl, *refr; double* refr_eta;
};
void orc_materials_kat(const OrcMatKat* k) {
    Ctx c; c.rng.seed_ref(21, 22, 23); c.math_mode = k->math_mode;
    auto g3 = [](const float* a, int i) { return V3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); };
    auto p3 = [](float* a, int i, const V3& v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; };
    Sphere lightS(V3(190, 90, 190), 90);
    Rect lightXZ(1, 213, 343, 227, 332, 554), lightXY(0, 3, 5, 1, 3, -2), lightYZ(2, 0, 2, 0, 2, 5);
    for (int i = 0; i < k->n; ++i) {
        V3 d = g3(k->in_d, i), nf = g3(k->in_n, i), p = g3(k->in_p, i);
        bool front = k->in_ff[i];
        float fuzz = k->prm[5 * i]; V3 alb(k->prm[5 * i + 1], k->prm[5 * i + 2], k->prm[5 * i + 3]);
        double ir = k->ir[i];
        HitRecord rec; rec.p = p; rec.normal = nf; rec.front_face = front; rec.t = 1.0;
        Ray rin(p - d, d);
        { Metal m(alb, fuzz < 1.0f ? fuzz : 1.0f); ScatterRecord s; m.scatter(c, rin, rec, s);
          p3(k->metal_o, i, s.specular_ray.o); p3(k->metal_d, i, s.specular_ray.d); p3(k->metal_att, i, s.attenuation); }
        { Dielectric m(ir); ScatterRecord s; m.scatter(c, rin, rec, s);
          p3(k->diel_d, i, s.specular_ray.d); p3(k->diel_att, i, s.attenuation); }
        { Lambertian m(alb); ScatterRecord s; m.scatter(c, rin, rec, s);
          p3(k->lamb_att, i, s.attenuation);
          V3 sd = g3(k->lamb_sd, i);
          k->lamb_spdf[i] = m.scattering_pdf(rin, rec, Ray(p, sd));
          k->cos_val[i] = s.pdf_ptr->value(c, sd);
          p3(k->cos_gen, i, s.pdf_ptr->generate(c));
          HittablePdf hs(&lightS, p);
          V3 gs = hs.generate(c); p3(k->sph_gen, i, gs);
          k->sph_val[i] = hs.value(c, i % 2 ? gs : sd);
          V3 pr = g3(k->xz_o, i);
          HittablePdf hr(&lightXZ, pr);
          V3 gr = hr.generate(c); p3(k->xz_gen, i, gr);
          k->xz_val[i] = hr.value(c, i % 2 ? gr : sd);
          HittablePdf hxy(&lightXY, p);
          V3 gxy = hxy.generate(c); p3(k->xy_gen, i, gxy);
          k->xy_val[i] = hxy.value(c, i % 2 ? V3(4, 2, -2) - p : sd);
          HittablePdf hyz(&lightYZ, p);
          p3(k->yz_gen, i, hyz.generate(c));
          k->yz_val[i] = hyz.value(c, sd);
          MixturePdf mp(&hs, s.pdf_ptr.get());
          V3 gm = mp.generate(c); p3(k->mix_gen, i, gm);
          k->mix_val[i] = mp.value(c, gm); }
        { V3 ud = d.normalized();
          p3(k->refl, i, reflect(ud, nf));
          double eta = front ? 1.0 / ir : ir;
          k->refr_eta[i] = eta;
          p3(k->refr, i, refract(ud, nf, eta)); }
    }
}

void orc_tonemap(int n, const float* in, const int32_t* ns, float* out) {
    for (int i = 0; i < n; ++i) {
        V3 r = rgbPerSamples(V3(in[3 * i], in[3 * i + 1], in[3 * i + 2]), ns[i]);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
// rgbPerSamples + writeToImg packing (rayview.cpp:57-65,86): RGB32 image mirrored on both axes.
// NaN channels are written as 0 (QColor(int) of a NaN is undefined behaviour in the reference).
void orc_tonemap_rgb32(int w, int h, const float* accum4, int n_samples, uint32_t* rgb32) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* a = accum4 + 4 * (size_t(y) * w + x);
            int n = n_samples > 0 ? n_samples : int(a[3]);
            V3 c = rgbPerSamples(V3(a[0], a[1], a[2]), n);
            auto q = [](float v) { return (v != v) ? 0u : uint32_t(int(v)) & 255u; };
            rgb32[size_t(h - 1 - y) * w + (w - 1 - x)] = 0xff000000u | (q(c.x) << 16) | (q(c.y) << 8) | q(c.z);
        }
}

// hitFromList over caller rays (all outputs optional except prim)
int orc_hit_rays(const RtScene* sc, int64_t n, const float* ro, const float* rd, double t_min, double t_max, int32_t* prim,
                 double* t, float* p, float* nrm, uint8_t* ff) {
    World w; if (!build_world(sc, w)) return 1;
    Ctx c; c.math_mode = 1;
    for (int64_t i = 0; i < n; ++i) {
        Ray r(V3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), V3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]));
        c.rng.key(0, uint32_t(i), 0);   // a constant_medium in the list draws from the ray's own counter stream (as rt_hit_rays)
        auto h = hitFromList(c, w.list, r, t_min, t_max);
        prim[i] = h.has_value() ? h->prim : -1;
        if (t) t[i] = h.has_value() ? h->t : 0.0;
        if (p) for (int k = 0; k < 3; ++k) p[3 * i + k] = h.has_value() ? h->p[k] : 0.0f;
        if (nrm) for (int k = 0; k < 3; ++k) nrm[3 * i + k] = h.has_value() ? h->normal[k] : 0.0f;
        if (ff) ff[i] = h.has_value() ? h->front_face : 0;
    }
    return 0;
}

// calc::ray_color for caller rays; ray i uses CounterRng keyed (seed, i, 0) (as rt_trace_rays).
int orc_trace_rays(const RtScene* sc, int64_t n, const float* ro, const float* rd, int depth, uint64_t seed, int math_mode,
                   uint32_t flags, float* rgb) {
    World w; if (!build_world(sc, w)) return 1;
    Ctx c; c.math_mode = math_mode;
    for (int64_t i = 0; i < n; ++i) {
        c.rng.key(seed, uint32_t(i), 0);
        Ray r(V3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), V3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]));
        V3 col = ray_color(c, r, w, depth, (flags & RT_RENDER_NORMALS) != 0);
        rgb[3 * i] = col.x; rgb[3 * i + 1] = col.y; rgb[3 * i + 2] = col.z;
    }
    return 0;
}

// Pixel-centre primary hits: u = (x + 0.5f) / W, v = (y + 0.5f) / H through camera::get_ray.
// rng_mode 0: RefRng seeded (1,2,3) once, pixels in row-major order (as ref_harness cmd_primary);
// rng_mode 1: CounterRng keyed (seed, pixel, 0) per pixel (as rt_primary_hits).
int orc_primary(const RtScene* sc, const RtCamera* cc, int W, int H, int rng_mode, uint64_t seed, int32_t* prim, double* t) {
    World w; if (!build_world(sc, w)) return 1;
    Camera cam(*cc);
    Ctx c; c.rng.seed_ref(1, 2, 3);
    c.math_mode = rng_mode == 1 ? 1 : 0;   // the GPU's configuration: counter RNG + portable log (a constant_medium calls log in the scan)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (rng_mode == 1) c.rng.key(seed, uint32_t(y * W + x), 0);
            float u = (x + 0.5f) / W, v = (y + 0.5f) / H;
            Ray r = cam.get_ray(c, u, v);
            auto h = hitFromList(c, w.list, r, kSmallest, kInf);
            prim[size_t(y) * W + x] = h.has_value() ? h->prim : -1;
            if (t) t[size_t(y) * W + x] = h.has_value() ? h->t : 0.0;
        }
    return 0;
}

// RayView::renderOneRay's per-pixel contract (rayview.cpp:79-84) over a full frame.
// rng_mode 0: one RefRng seeded (seeds[0..2]) for the whole frame, pixels row-major, samples inner
//             (as ref_harness cmd_render) -- single-threaded by construction;
// rng_mode 1: CounterRng keyed (seed64, pixel, sample); rows split over n_threads.
// accum4: [H][W][4] = (r, g, b sums over the samples in order, number of samples); rows outside
// [y_begin, y_begin + y_count) untouched. flags: RT_RENDER_ACCUMULATE, RT_RENDER_NORMALS.
int orc_render(const RtScene* sc, const RtCamera* cc, int W, int H, int y_begin, int y_count, int spp_begin, int spp_count,
               int depth, int rng_mode, const uint32_t* seeds, uint64_t seed64, int math_mode, uint32_t flags,
               int n_threads, float* accum4, uint64_t* rays_out) {
    World w; if (!build_world(sc, w)) return 1;
    Camera cam(*cc);
    const bool normals = (flags & RT_RENDER_NORMALS) != 0, accumulate = (flags & RT_RENDER_ACCUMULATE) != 0;
    auto rows = [&](int ya, int yb, Ctx& c) {
        for (int y = ya; y < yb; ++y)
            for (int x = 0; x < W; ++x) {
                V3 sum(0, 0, 0);
                for (int s = spp_begin; s < spp_begin + spp_count; ++s) {
                    if (rng_mode == 1) c.rng.key(seed64, uint32_t(y * W + x), uint32_t(s));
                    float u = (x + c.rng.random_double01()) / W;
                    float v = (y + c.rng.random_double01()) / H;
                    V3 col = ray_color(c, cam.get_ray(c, u, v), w, depth, normals);
                    sum = sum + col;
                }
                float* a = accum4 + 4 * (size_t(y) * W + x);
                if (accumulate) { a[0] += sum.x; a[1] += sum.y; a[2] += sum.z; a[3] += float(spp_count); }
                else { a[0] = sum.x; a[1] = sum.y; a[2] = sum.z; a[3] = float(spp_count); }
            }
    };
    uint64_t total = 0;
    if (rng_mode == 0 || n_threads <= 1) {
        Ctx c; c.math_mode = math_mode;
        if (rng_mode == 0) c.rng.seed_ref(seeds[0], seeds[1], seeds[2]);
        rows(y_begin, y_begin + y_count, c);
        total = c.rays;
    } else {
        std::vector<std::thread> th;
        std::vector<uint64_t> cnt(n_threads, 0);
        for (int k = 0; k < n_threads; ++k)
            th.emplace_back([&, k] {
                Ctx c; c.math_mode = math_mode;
                for (int y = y_begin + k; y < y_begin + y_count; y += n_threads) rows(y, y + 1, c);
                cnt[k] = c.rays;
            });
        for (auto& t : th) t.join();
        for (auto v : cnt) total += v;
    }
    if (rays_out) *rays_out = total;
    return 0;
}

}  // extern "C"
