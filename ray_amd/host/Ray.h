// Mirror of the reference's Ray.h (Ray.h:6-16): same members, same constructor, same at().
#ifndef RAY_H
#define RAY_H
#ifndef RAY_AMD_MIRROR_OUTER   // the header the translation unit included itself pulls in ray_amd_host.h at its end
#define RAY_AMD_MIRROR_OUTER
#define RAY_AMD_MIRROR_OUTER_IS_RAY_H
#endif

#include "qvector3d_standin.h"

struct Ray {
    Ray() {}
    Ray(const QVector3D& origin, const QVector3D& direction)
        : pointOfOrigin(origin), direction(direction), DirlengthSquared(direction.lengthSquared()) {}   // Ray.h:8
    QVector3D at(double t) const { return pointOfOrigin + float(t) * direction; }                        // Ray.h:11: t is narrowed by operator*(float, QVector3D)
    QVector3D pointOfOrigin;
    QVector3D direction;
    float DirlengthSquared = 0.0f;
};

#ifdef RAY_AMD_MIRROR_OUTER_IS_RAY_H
#undef RAY_AMD_MIRROR_OUTER_IS_RAY_H
#undef RAY_AMD_MIRROR_OUTER
#include "ray_amd_host.h"
#endif
#endif  // RAY_H
