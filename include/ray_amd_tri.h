/* ray_amd_tri.h -- bounding sphere of RT_PRIM_TRIANGLE (include/ray_amd.h).
 *
 * The reference has no triangle primitive ("parity unpinned"); this build defines it
 * as FP32 Moeller-Trumbore with separately rounded operations PLUS the condition that the hit point
 * p = origin + float(t) * direction lies inside the bounding sphere below (|p - centre|^2 <= radius2
 * in separately rounded FP32). The condition removes the spurious hits a floating-point
 * Moeller-Trumbore test reports for rays that lie in the plane of a far-away triangle (det, u, v
 * all rounding noise), and it is what lets the HIP path cull triangles by a conservative
 * line-sphere filter without changing any result. The CPU oracle and the HIP library both take
 * the sphere from this one function, so that its rounding is part of the primitive's definition.
 * Plain C, host only, double arithmetic.                                                        */
#ifndef RAY_AMD_TRI_H
#define RAY_AMD_TRI_H

#include <math.h>

static inline void rt_triangle_bound(const float v0[3], const float v1[3], const float v2[3], float centre[3], float* radius2) {
    double a[3], b[3], c[3], ab[3], ac[3], bc[3], ctr[3];
    int k;
    for (k = 0; k < 3; ++k) { a[k] = v0[k]; b[k] = v1[k]; c[k] = v2[k]; ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; bc[k] = c[k] - b[k]; }
    double ab2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
    double ac2 = ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2];
    double bc2 = bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2];
    double n[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
    double n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    int use_edge = 0;   /* 1: ab, 2: ac, 3: bc -- the midpoint of the longest edge when the triangle is not acute */
    if (ab2 >= ac2 + bc2) use_edge = 1;
    else if (ac2 >= ab2 + bc2) use_edge = 2;
    else if (bc2 >= ab2 + ac2) use_edge = 3;
    else if (!(n2 > 0.0)) use_edge = (ab2 >= ac2 && ab2 >= bc2) ? 1 : (ac2 >= bc2 ? 2 : 3);
    if (use_edge == 1) for (k = 0; k < 3; ++k) ctr[k] = 0.5 * (a[k] + b[k]);
    else if (use_edge == 2) for (k = 0; k < 3; ++k) ctr[k] = 0.5 * (a[k] + c[k]);
    else if (use_edge == 3) for (k = 0; k < 3; ++k) ctr[k] = 0.5 * (b[k] + c[k]);
    else {   /* circumcentre: a + ((n x ab) |ac|^2 + (ac x n) |ab|^2) / (2 |n|^2) */
        double nxab[3] = {n[1] * ab[2] - n[2] * ab[1], n[2] * ab[0] - n[0] * ab[2], n[0] * ab[1] - n[1] * ab[0]};
        double acxn[3] = {ac[1] * n[2] - ac[2] * n[1], ac[2] * n[0] - ac[0] * n[2], ac[0] * n[1] - ac[1] * n[0]};
        for (k = 0; k < 3; ++k) ctr[k] = a[k] + (nxab[k] * ac2 + acxn[k] * ab2) / (2.0 * n2);
    }
    for (k = 0; k < 3; ++k) centre[k] = (float)ctr[k];
    /* radius from the ROUNDED centre to the farthest vertex, then 1 % + 1e-5 of the coordinates' size of
     * slack: the FP32 hit point of a hit on the triangle's rim must pass the test                 */
    double r2 = 0.0, cn2 = 0.0;
    const double* vs[3] = {a, b, c};
    int i;
    for (i = 0; i < 3; ++i) {
        double d2 = 0.0;
        for (k = 0; k < 3; ++k) { double d = vs[i][k] - (double)centre[k]; d2 += d * d; }
        if (d2 > r2) r2 = d2;
    }
    for (k = 0; k < 3; ++k) cn2 += (double)centre[k] * (double)centre[k];
    double r = sqrt(r2);
    double rr = 1.01 * r + 1.0e-5 * (sqrt(cn2) + r);
    double want = rr * rr;
    float out = (float)want;
    if ((double)out < want) out = nextafterf(out, INFINITY);
    *radius2 = out;   /* +inf / NaN for non-finite vertices: NaN fails the containment test (a miss), as the NaN det does */
}

#endif /* RAY_AMD_TRI_H */
