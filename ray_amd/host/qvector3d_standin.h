// QVector3D for builds without Qt. The reference's ray path does all its vector arithmetic on Qt5Gui's QVector3D
// (pinned 5.15.2, CMakeLists.txt:10; not vendored). With -DRAY_AMD_USE_QT the real <QVector3D> is used; otherwise
// this stand-in, which restates the semantics of the members the path uses (verified there against
// libQt5Gui and pinned here by tests/golden/vec3.npz through the host test): inline operators are plain FP32 per
// component; length / normalized go through double; scalar * vector takes a FLOAT, so double scalars are narrowed
// first. Compile host code with -ffp-contract=off (x86-64 baseline has no FMA anyway).
#ifndef RAY_AMD_QVECTOR3D_STANDIN_H
#define RAY_AMD_QVECTOR3D_STANDIN_H

#ifdef RAY_AMD_USE_QT
#include <QVector3D>
#else
#include <cmath>

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
    float lengthSquared() const { return xp * xp + yp * yp + zp * zp; }
    float length() const { return float(std::sqrt(len2())); }
    QVector3D normalized() const {
        const double len = len2();
        if (std::fabs(len - 1.0) <= 0.000000000001) return *this;
        if (!(std::fabs(len) <= 0.000000000001)) {
            const double s = std::sqrt(len);
            return QVector3D(float(double(xp) / s), float(double(yp) / s), float(double(zp) / s));
        }
        return QVector3D();
    }
    static float dotProduct(const QVector3D& a, const QVector3D& b) { return a.xp * b.xp + a.yp * b.yp + a.zp * b.zp; }
    static QVector3D crossProduct(const QVector3D& a, const QVector3D& b) {
        return QVector3D(a.yp * b.zp - a.zp * b.yp, a.zp * b.xp - a.xp * b.zp, a.xp * b.yp - a.yp * b.xp);
    }
    QVector3D& operator+=(const QVector3D& o) { xp += o.xp; yp += o.yp; zp += o.zp; return *this; }
    QVector3D& operator-=(const QVector3D& o) { xp -= o.xp; yp -= o.yp; zp -= o.zp; return *this; }
    QVector3D& operator*=(float f) { xp *= f; yp *= f; zp *= f; return *this; }
    friend QVector3D operator+(const QVector3D& a, const QVector3D& b) { return QVector3D(a.xp + b.xp, a.yp + b.yp, a.zp + b.zp); }
    friend QVector3D operator-(const QVector3D& a, const QVector3D& b) { return QVector3D(a.xp - b.xp, a.yp - b.yp, a.zp - b.zp); }
    friend QVector3D operator*(const QVector3D& a, float f) { return QVector3D(a.xp * f, a.yp * f, a.zp * f); }
    friend QVector3D operator*(float f, const QVector3D& a) { return QVector3D(a.xp * f, a.yp * f, a.zp * f); }
    friend QVector3D operator*(const QVector3D& a, const QVector3D& b) { return QVector3D(a.xp * b.xp, a.yp * b.yp, a.zp * b.zp); }
    friend QVector3D operator/(const QVector3D& a, float d) { return QVector3D(a.xp / d, a.yp / d, a.zp / d); }
    friend QVector3D operator-(const QVector3D& a) { return QVector3D(-a.xp, -a.yp, -a.zp); }
    friend bool operator==(const QVector3D& a, const QVector3D& b) { return a.xp == b.xp && a.yp == b.yp && a.zp == b.zp; }
private:
    double len2() const { return double(xp) * double(xp) + double(yp) * double(yp) + double(zp) * double(zp); }
    float xp, yp, zp;
};
#endif  // RAY_AMD_USE_QT
#endif  // RAY_AMD_QVECTOR3D_STANDIN_H
