// Mirror of the reference's camera.h: same constructor, resetCam and get_ray signatures. resetCam's arithmetic is
// the library's rt_camera_from_lookat (identical to camera.h:32-54, pinned by tests/golden/camera.npz); get_ray is
// the reference's host arithmetic on the reference's own engines (camera.h:21-30, calc::random_in_unit_disk), for
// callers that build rays themselves as RayView::renderOneRay does. Frames render their rays on the GPU.
#ifndef CAMERA_H
#define CAMERA_H
#ifndef RAY_AMD_MIRROR_OUTER   // the header the translation unit included itself pulls in ray_amd_host.h at its end
#define RAY_AMD_MIRROR_OUTER
#define RAY_AMD_MIRROR_OUTER_IS_CAMERA_H
#endif

#include "Ray.h"
#include "calc.h"
#include "hitPosition.h"

class camera {
public:
    camera(QVector3D lookfrom, QVector3D lookat, QVector3D vup, double focus_dist, double vfov = 40.0,
           double aspect_ratio = img::aspect_ratio, double aperture = 0.1) {
        resetCam(lookfrom, lookat, vup, focus_dist, vfov, aspect_ratio, aperture);
    }
    Ray get_ray(double s, double t) const {
        QVector3D rd = float(lens_radius) * calc::random_in_unit_disk();
        QVector3D offset = u * rd.x() + v * rd.y();
        return Ray(origin + offset, lower_left_corner + float(s) * horizontal + float(t) * vertical - origin - offset);
    }
    void resetCam(QVector3D lookfrom, QVector3D lookat, QVector3D vup, double focus_dist, double vfov = 40.0,
                  double aspect_ratio = img::aspect_ratio, double aperture = 0.1) {
        const float f[3] = {lookfrom.x(), lookfrom.y(), lookfrom.z()}, a[3] = {lookat.x(), lookat.y(), lookat.z()},
                    up[3] = {vup.x(), vup.y(), vup.z()};
        rt_host::check(rt_camera_from_lookat(f, a, up, focus_dist, vfov, aspect_ratio, aperture, &c));
        auto v3 = [](const float* p) { return QVector3D(p[0], p[1], p[2]); };
        origin = v3(c.origin); lower_left_corner = v3(c.lower_left_corner); horizontal = v3(c.horizontal); vertical = v3(c.vertical);
        u = v3(c.u); v = v3(c.v); w = v3(c.w); lens_radius = c.lens_radius;
    }
    const RtCamera& rt() const { return c; }   // what rt_set_camera takes
private:
    QVector3D origin;
    QVector3D lower_left_corner;
    QVector3D horizontal;
    QVector3D vertical;
    QVector3D w, u, v;
    double lens_radius = 0;
    RtCamera c;
};

#ifdef RAY_AMD_MIRROR_OUTER_IS_CAMERA_H
#undef RAY_AMD_MIRROR_OUTER_IS_CAMERA_H
#undef RAY_AMD_MIRROR_OUTER
#include "ray_amd_host.h"
#endif
#endif  // CAMERA_H
