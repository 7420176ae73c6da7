// Known-answer driver for the HOST arithmetic of the mirror headers (no GPU work): the same call sequences that
// oracle/ref_harness.cpp ran through the reference to produce tests/golden/{rng,vec3,camera,materials}.npz, run
// through ray_amd/host/{calc,camera}.h and the QVector3D stand-in. tests/test_host_mirror_cpu.py feeds it the
// goldens' inputs and compares the outputs bit for bit. Raw little-endian arrays in, raw arrays out.
//   mirror_kat rng SEED OUT            u01, u11, gen(0,1), gen(3,5) [4096 f4 each], W, H [4096 i4 each]
//   mirror_kat samplers OUT            seeds (7,8,9): in_unit_sphere, in_unit_disk, cosine_direction, to_sphere [512x3 f4 each]
//   mirror_kat vec3 N IN OUT           IN: a[N][3], b[N][3]   OUT: lengthSquared[N], length[N], normalized[N][3], dot[N], cross[N][3]
//   mirror_kat camera SEED11 N IN OUT  IN: from, at, up [3 f4 each], focus vfov aspect aperture [4 f8], st[N][2] f4
//                                      OUT: frame[7][3] f4, lens_radius f8, ray_o[N][3], ray_d[N][3], dl2[N]
//   mirror_kat tone N IN OUT           IN: rgb[N][3] f4, n[N] i4   OUT: rgbPerSamples [N][3] f4
//   mirror_kat optics N IN OUT         IN: d[N][3], n[N][3] f4, eta[N] f8   OUT: reflect[N][3], refract[N][3] (of d.normalized())
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <calc.h>
#include <camera.h>

static std::vector<char> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<char> b;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}
struct Out {
    FILE* f;
    explicit Out(const char* p) : f(fopen(p, "wb")) { if (!f) { perror(p); exit(2); } }
    ~Out() { fclose(f); }
    void f4(float v) { fwrite(&v, 4, 1, f); }
    void f8(double v) { fwrite(&v, 8, 1, f); }
    void i4(int32_t v) { fwrite(&v, 4, 1, f); }
    void v3(const QVector3D& v) { f4(v.x()); f4(v.y()); f4(v.z()); }
};
static QVector3D at3(const float* p, size_t i) { return QVector3D(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string c = argv[1];
    if (c == "rng") {
        const unsigned s = unsigned(atol(argv[2]));
        Out o(argv[3]);
        calc::gen.seed(s); calc::gen01.seed(s); calc::gen11.seed(s);
        for (int i = 0; i < 4096; ++i) o.f4(calc::random_double01());
        for (int i = 0; i < 4096; ++i) o.f4(calc::random_double11());
        for (int i = 0; i < 4096; ++i) o.f4(calc::random_double(0.0f, 1.0f));
        for (int i = 0; i < 4096; ++i) o.f4(calc::random_double(3.0f, 5.0f));
        for (int i = 0; i < 4096; ++i) o.i4(calc::random_doubleW());
        for (int i = 0; i < 4096; ++i) o.i4(calc::random_doubleH());
    } else if (c == "samplers") {
        Out o(argv[2]);
        calc::gen.seed(7); calc::gen01.seed(8); calc::gen11.seed(9);
        for (int i = 0; i < 512; ++i) o.v3(calc::random_in_unit_sphere());
        for (int i = 0; i < 512; ++i) o.v3(calc::random_in_unit_disk());
        for (int i = 0; i < 512; ++i) o.v3(calc::random_cosine_direction());
        for (int i = 0; i < 512; ++i) o.v3(calc::random_to_sphere(90.0, 40000.0 + 100.0 * i));
    } else if (c == "vec3") {
        const size_t n = size_t(atol(argv[2]));
        const auto in = slurp(argv[3]);
        const float* a = reinterpret_cast<const float*>(in.data());
        const float* b = a + 3 * n;
        Out o(argv[4]);
        for (size_t i = 0; i < n; ++i) o.f4(at3(a, i).lengthSquared());
        for (size_t i = 0; i < n; ++i) o.f4(at3(a, i).length());
        for (size_t i = 0; i < n; ++i) o.v3(at3(a, i).normalized());
        for (size_t i = 0; i < n; ++i) o.f4(QVector3D::dotProduct(at3(a, i), at3(b, i)));
        for (size_t i = 0; i < n; ++i) o.v3(QVector3D::crossProduct(at3(a, i), at3(b, i)));
    } else if (c == "camera") {
        const unsigned s11 = unsigned(atol(argv[2]));
        const size_t n = size_t(atol(argv[3]));
        const auto in = slurp(argv[4]);
        const float* v = reinterpret_cast<const float*>(in.data());
        double prm[4];
        memcpy(prm, in.data() + 36, 32);
        const float* st = reinterpret_cast<const float*>(in.data() + 68);
        camera cam(at3(v, 0), at3(v, 1), at3(v, 2), prm[0], prm[1], prm[2], prm[3]);
        Out o(argv[5]);
        const RtCamera& rc = cam.rt();
        for (const float* p : {rc.origin, rc.lower_left_corner, rc.horizontal, rc.vertical, rc.u, rc.v, rc.w}) { o.f4(p[0]); o.f4(p[1]); o.f4(p[2]); }
        o.f8(rc.lens_radius);
        calc::gen.seed(11); calc::gen01.seed(12); calc::gen11.seed(s11);
        std::vector<Ray> rays;
        for (size_t i = 0; i < n; ++i) rays.push_back(cam.get_ray(st[2 * i], st[2 * i + 1]));
        for (auto& r : rays) o.v3(r.pointOfOrigin);
        for (auto& r : rays) o.v3(r.direction);
        for (auto& r : rays) o.f4(r.DirlengthSquared);
    } else if (c == "tone") {
        const size_t n = size_t(atol(argv[2]));
        const auto in = slurp(argv[3]);
        const float* rgb = reinterpret_cast<const float*>(in.data());
        const int32_t* ns = reinterpret_cast<const int32_t*>(in.data() + 12 * n);
        Out o(argv[4]);
        for (size_t i = 0; i < n; ++i) o.v3(calc::rgbPerSamples(at3(rgb, i), ns[i]));
    } else if (c == "optics") {
        const size_t n = size_t(atol(argv[2]));
        const auto in = slurp(argv[3]);
        const float* d = reinterpret_cast<const float*>(in.data());
        const float* nn = d + 3 * n;
        const double* eta = reinterpret_cast<const double*>(in.data() + 24 * n);
        Out o(argv[4]);
        for (size_t i = 0; i < n; ++i) o.v3(calc::reflect(at3(d, i).normalized(), at3(nn, i)));
        for (size_t i = 0; i < n; ++i) o.v3(calc::refract(at3(d, i).normalized(), at3(nn, i), eta[i]));
    } else {
        return 2;
    }
    return 0;
}
