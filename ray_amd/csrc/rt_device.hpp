// Device-side arithmetic of the ray path for gfx950. Everything here is compiled with
// -ffp-contract=off: the reference's arithmetic (g++ on x86-64) is FP32
// QVector3D operators with every multiply and add rounded separately, and FP64 wherever the
// reference calls an unqualified <cmath> function. IEEE +,-,*,/ and sqrt are correctly rounded on
// gfx950 in both precisions, so the same sequence of operations gives the same bits as the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __host__ __device__ __forceinline__

namespace rt {

constexpr double kPi = 3.1415926535897932385;   // reference calc.h:31
constexpr double kInf = __builtin_huge_val();   // reference calc.h:30

// ---- QVector3D arithmetic (Qt5Gui qvector3d.h:184-284; out-of-line length/normalized/dot/cross as
// restated in SURVEY.md 8c) ---------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
RT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(float f, V3 v) { return v3(v.x * f, v.y * f, v.z * f); }
RT_DEV V3 operator*(V3 v, float f) { return v3(v.x * f, v.y * f, v.z * f); }
RT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator-(V3 v) { return v3(-v.x, -v.y, -v.z); }
RT_DEV V3 operator/(V3 v, float d) { return v3(v.x / d, v.y / d, v.z / d); }
// v / d (QVector3D / float, three IEEE divisions by one divisor) with the divisor's part of the work done once. hipcc's FP32
// division on gfx950 is v_div_scale (x2), v_rcp, one Newton step on the reciprocal, the quotient, two residual corrections (the
// last one as v_div_fmas) and v_div_fixup; when the divisor and the numerator are normal numbers whose exponents are less than
// 96 apart, v_div_scale passes both through unscaled, v_div_fmas is a plain fma and v_div_fixup returns its first operand, so the
// quotient is exactly the fma chain below. The guard keeps well inside that: divisor and non-zero numerators in [2^-40, 2^40]
// (radiance and pdf values are O(1)); a zero numerator is let through as well and gives +0 where IEEE gives a signed zero -
// the one caller adds the quotient to +0 (calc.cpp:129, `emitted + ...` with emitted = 0), which erases the sign. A NaN
// numerator stays NaN (operands must be results of arithmetic, as the caller's products and conversions are: a SIGNALING
// NaN would slip through the max / min of the first test). Everything else (infinities, huge, tiny) takes `/`. tools/div_check.hip compares the chain with `/` on 2^32 operand pairs.
RT_DEV bool div3_shared_ok(V3 n, float d) {
    // the usual case first, four compares: no numerator is zero (or all three are), nothing is NaN or infinite
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(n.x), __builtin_fabsf(n.y)), __builtin_fabsf(n.z));
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(n.x), __builtin_fabsf(n.y)), __builtin_fabsf(n.z));
    const bool d_ok = d >= 0x1p-40f && d <= 0x1p40f;
    bool ok = d_ok && hi <= 0x1p40f && (lo >= 0x1p-40f || hi == 0.0f);
#ifdef __HIP_DEVICE_COMPILE__
    if (__ballot(!ok) == 0ull) return true;      // (wave-uniform: the second test runs only for waves that need it)
#endif
    // colours with ONE zero channel are common too ((0.8, 0.8, 0) is the built-in scene's ground): on the bit patterns, sign
    // shifted out, every numerator is 0 or in [2^-40, 2^40] (u - 1 wraps a zero to the top)
    const uint32_t kLo = 0x2b800000u, kHi = 0x53800000u;      // 2^-40, 2^40
    const uint32_t ux = __builtin_bit_cast(uint32_t, n.x) << 1, uy = __builtin_bit_cast(uint32_t, n.y) << 1, uz = __builtin_bit_cast(uint32_t, n.z) << 1;
    const uint32_t top = ux > uy ? (ux > uz ? ux : uz) : (uy > uz ? uy : uz);
    const uint32_t a = ux - 1u, b = uy - 1u, c = uz - 1u;
    const uint32_t bottom = a < b ? (a < c ? a : c) : (b < c ? b : c);
    return ok || (d_ok && top <= (kHi << 1) && bottom >= (kLo << 1) - 1u);
}
#ifdef __HIP_DEVICE_COMPILE__
RT_DEV V3 div3_chain(V3 n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q[3] = {n.x * r, n.y * r, n.z * r};
    const float nn[3] = {n.x, n.y, n.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float rem = __builtin_fmaf(-d, q[k], nn[k]);
        q[k] = __builtin_fmaf(rem, r, q[k]);
        rem = __builtin_fmaf(-d, q[k], nn[k]);
        q[k] = __builtin_fmaf(rem, r, q[k]);
    }
    return V3{q[0], q[1], q[2]};
}
#endif
// (+0, +0, +0) + n / d
RT_DEV V3 div3_zero_plus(V3 n, float d) {
#ifdef __HIP_DEVICE_COMPILE__
    if (div3_shared_ok(n, d)) {
        const V3 q = div3_chain(n, d);
        return V3{0.0f + q.x, 0.0f + q.y, 0.0f + q.z};
    }
#endif
    return V3{0.0f + n.x / d, 0.0f + n.y / d, 0.0f + n.z / d};
}
// n / d, signed zeros included: the chain only when no numerator is zero (the first test of div3_shared_ok without its exception)
RT_DEV V3 div3(V3 n, float d) {
#ifdef __HIP_DEVICE_COMPILE__
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(n.x), __builtin_fabsf(n.y)), __builtin_fabsf(n.z));
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(n.x), __builtin_fabsf(n.y)), __builtin_fabsf(n.z));
    if (d >= 0x1p-40f && d <= 0x1p40f && hi <= 0x1p40f && lo >= 0x1p-40f) return div3_chain(n, d);
#endif
    return V3{n.x / d, n.y / d, n.z / d};
}
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV float length_squared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
RT_DEV double len2_f64(V3 v) { return double(v.x) * double(v.x) + double(v.y) * double(v.y) + double(v.z) * double(v.z); }
RT_DEV float length(V3 v) { return float(__builtin_sqrt(len2_f64(v))); }
// IEEE double division with the reciprocal of the divisor shared between several numerators. The
// operations are the ones hipcc emits for `n / d` on gfx950 (v_rcp_f64, two Newton steps on the
// reciprocal, quotient, exact remainder, one correction) minus v_div_scale / v_div_fixup, which only
// act on operands near the ends of the exponent range; callers guard those and use `/` there.
struct Recip64 {
    double d, r;
};
RT_DEV Recip64 recip64(double d) {
    Recip64 R; R.d = d;
#ifdef __HIP_DEVICE_COMPILE__
    double r0 = __builtin_amdgcn_rcp(d);
    double e0 = __builtin_fma(-d, r0, 1.0);
    double r1 = __builtin_fma(r0, e0, r0);
    double e1 = __builtin_fma(-d, r1, 1.0);
    R.r = __builtin_fma(r1, e1, r1);
#else
    R.r = 1.0 / d;
#endif
    return R;
}
// n / R.d for |n| <= 1e300 (or NaN) and R.d in the guarded range of the caller
RT_DEV double div64(double n, const Recip64& R) {
#ifdef __HIP_DEVICE_COMPILE__
    double q = n * R.r;
    double rem = __builtin_fma(-R.d, q, n);
    return __builtin_fma(rem, R.r, q);
#else
    return n / R.d;
#endif
}
RT_DEV bool recip_range_ok(double d) { return d >= 1.0e-150 && d <= 1.0e150; }   // false for NaN, 0, inf, negative

// IEEE double sqrt for 1e-280 <= x <= 1e280: the operations hipcc emits for __builtin_sqrt on gfx950
// (v_rsq_f64 + Goldschmidt steps + two residual corrections) without the exponent scaling and the
// 0 / inf / NaN selects, which only act outside this range. Callers that cannot prove the range use sqrt64().
RT_DEV double sqrt64_in_range(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return g;
#else
    return __builtin_sqrt(x);
#endif
}
RT_DEV double sqrt64(double x) {
    if (x >= 1.0e-280 && x <= 1.0e280) return sqrt64_in_range(x);
    return __builtin_sqrt(x);
}

RT_DEV V3 normalized(V3 v) {
#if defined(RT_EXPERIMENT_FAST_NORMALIZE) && defined(__HIP_DEVICE_COMPILE__)
    { float r_ = __builtin_amdgcn_rsqf(length_squared(v)); return v3(v.x * r_, v.y * r_, v.z * r_); }   // timing experiment only: wrong bits
#endif
    double len = len2_f64(v);
    if (__builtin_fabs(len - 1.0) <= 0.000000000001) return v;
    if (!(__builtin_fabs(len) <= 0.000000000001)) {
        if (len <= 1.0e280) {   // always, unless a component is inf / NaN; len > 1e-12 here
            double s = sqrt64_in_range(len);
            Recip64 R = recip64(s);
            return v3(float(div64(double(v.x), R)), float(div64(double(v.y), R)), float(div64(double(v.z), R)));
        }
        double s = __builtin_sqrt(len);
        return v3(float(double(v.x) / s), float(double(v.y) / s), float(double(v.z) / s));
    }
    return v3(0.0f, 0.0f, 0.0f);
}

// ---- portable transcendentals: fdlibm k_sin/k_cos + two-term pi/2 reduction for
// 0 <= x <= 2*pi, IEEE double +,-,* only; bit-identical to the CPU restatement the parity tests use.
namespace pm {
constexpr double invpio2 = 6.36619772367581382433e-01;
constexpr double pio2_1 = 1.57079632673412561417e+00;
constexpr double pio2_1t = 6.07710050650619224932e-11;
constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
}  // namespace pm
RT_DEV void sincos_portable(double x, double* s, double* c) {
    using namespace pm;
    int k = int(x * invpio2 + 0.5);
    double fn = double(k);
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double z = y0 * y0, zz = z * z;
    // k_sin
    double rs = S2 + z * (S3 + z * S4) + z * zz * (S5 + z * S6);
    double v = z * y0;
    double sn = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    // k_cos
    double rc = z * (C1 + z * (C2 + z * C3)) + (zz * zz) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    double wc = 1.0 - hz;
    double cs = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
    switch (k & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}
RT_DEV double pow5_portable(double x) { double x2 = x * x; double x4 = x2 * x2; return x4 * x; }
// fdlibm e_log.c (__ieee754_log) for x = 0 or positive normal finite x, with IEEE double +,-,*,/ and integer
// operations on the high word: bit-identical to the oracle's pm::log_pos. Used by constant_medium::hit (shapes.h:425).
namespace pm {
constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
}
RT_DEV double log_portable(double x) {
    using namespace pm;
    if (x == 0.0) return -kInf;
    uint64_t bits = __builtin_bit_cast(uint64_t, x);
    int32_t hx = int32_t(bits >> 32);
    int32_t k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i0 = (hx + 0x95f64) & 0x100000;
    bits = (uint64_t(uint32_t(hx | (i0 ^ 0x3ff00000))) << 32) | (bits & 0xffffffffull);
    x = __builtin_bit_cast(double, bits);
    k += (i0 >> 20);
    const double f = x - 1.0;
    const double dk = double(k);
    if ((0x000fffff & (2 + hx)) < 3) {
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

// ---- counter RNG (DESIGN.md "RNG contract"): PCG-RXS-M-XS-32 stream keyed by (seed, pixel, sample);
// the reference's three minstd_rand0 engines (calc.h:103-108) all map onto it. --------------------
RT_DEV uint32_t pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}
struct Rng {
    uint32_t state;
    RT_DEV void key(uint64_t seed, uint32_t pixel, uint32_t sample) {
        uint32_t lo = uint32_t(seed), hi = uint32_t(seed >> 32);
        state = pcg_hash(pixel + pcg_hash(sample + pcg_hash(lo ^ pcg_hash(hi))));
    }
    RT_DEV float canon() {
        state = state * 747796405u + 2891336453u;
        uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
        uint32_t bits = (word >> 22) ^ word;
        return float(bits >> 8) * 5.9604644775390625e-08f;
    }
    // uniform_real_distribution<float>(a, b): canon * (b - a) + a   (reference calc.cpp:12-36)
    RT_DEV float range(float a, float b) { return canon() * (b - a) + a; }
    RT_DEV float u01() { return canon() * (1.0f - 0.0f) + 0.0f; }
    RT_DEV float u11() { return canon() * (1.0f - -1.0f) + -1.0f; }
};

// reference calc.cpp:54-62 (randomVec11 is a brace-init list: x, y, z drawn in that order)
RT_DEV V3 random_in_unit_sphere(Rng& g) {
    V3 p = v3(-1.0f, -1.0f, -1.0f);
    while (length_squared(p) >= 1.0f) {
        float x = g.u11(); float y = g.u11(); float z = g.u11();
        p = v3(x, y, z);
    }
    return p;
}
// reference calc.cpp:64-72; the reference binary draws y before x (oracle/rt_oracle.cpp, pinned)
RT_DEV V3 random_in_unit_disk(Rng& g) {
    for (;;) {
        float y = g.u11();
        float x = g.u11();
        V3 p = v3(x, y, 0.0f);
        if (!(length_squared(p) >= 1.0f)) return p;
    }
}
// reference calc.h:62-73
RT_DEV V3 random_cosine_direction(Rng& g) {
    float r1 = g.u01();
    float r2 = g.u01();
    double z = __builtin_sqrt(double(1 - r2));
    double phi = 2 * kPi * r1;
    double sn, cs;
    sincos_portable(phi, &sn, &cs);
    double sr2 = __builtin_sqrt(double(r2));
    return v3(float(cs * sr2), float(sn * sr2), float(z));
}
// reference calc.h:75-86
RT_DEV V3 random_to_sphere(Rng& g, double radius, double distance_squared) {
    float r1 = g.u01();
    float r2 = g.u01();
    double z = 1 + r2 * (__builtin_sqrt(1 - radius * radius / distance_squared) - 1);
    double phi = 2 * kPi * r1;
    double sn, cs;
    sincos_portable(phi, &sn, &cs);
    double sq = __builtin_sqrt(1 - z * z);
    return v3(float(cs * sq), float(sn * sq), float(z));
}
// reference calc.cpp:163-174
RT_DEV V3 reflect(V3 v, V3 n) { return v - 2 * dot(v, n) * n; }
RT_DEV V3 refract(V3 uv, V3 n, double etai_over_etat) {
    double cos_theta = __builtin_fmin(double(dot(-uv, n)), 1.0);
    V3 r_out_perp = float(etai_over_etat) * (uv + float(cos_theta) * n);
    V3 r_out_parallel = float(-sqrt64(__builtin_fabs(1.0 - length_squared(r_out_perp)))) * n;
    return r_out_perp + r_out_parallel;
}

// reference hitPosition.h:112-139
struct Onb {
    V3 u, v, w;
    RT_DEV void build_from_w(V3 n) {
        w = normalized(n);
        V3 a = (__builtin_fabs(double(w.x)) > 0.9) ? v3(0, 1, 0) : v3(1, 0, 0);
        v = normalized(cross(w, a));
        u = cross(w, v);
    }
    RT_DEV V3 local(V3 a) const { return a.x * u + a.y * v + a.z * w; }
};

}  // namespace rt
