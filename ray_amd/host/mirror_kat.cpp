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
//   mirror_kat shapehits N IN OUT      IN: kind[N] i4 (0 sphere, 1 xy, 2 xz, 3 yz), prm[N][5] f8, o[N][3], d[N][3] f4, tmax[N] f8
//                                      OUT: shape::hit_host(Ray, 0.001, tmax): has[N] u1, t[N] f8, p[N][3], n[N][3] f4, ff[N] u1, uv[N][2] f8
//   mirror_kat shapehits_dev N IN OUT  the same through shape::hit (the one mode here that needs the GPU: one upload and one launch per shape)
//   mirror_kat materials N IN OUT      the call sequence of oracle/ref_harness.cpp cmd_materials on engines seeded (21, 22, 23).
//                                      IN: d, n, p [N][3] f4, ff[N] u1, prm[N][5] f4 (fuzz, albedo, -), ir[N] f8, sd[N][3] f4, pr[N][3] f4
//                                      OUT: metal o, d, att; dielectric d, att; lambertian att [N][3] f4; spdf, cos_val [N] f8; cos_gen;
//                                      then per light (sphere, xz, xy, yz): gen [N][3] f4, val [N] f8; mix_gen [N][3] f4, mix_val [N] f8
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <calc.h>
#include <camera.h>
#include <shapes.h>

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
    } else if (c == "shapehits" || c == "shapehits_dev") {
        const bool on_device = c == "shapehits_dev";   // shape::hit: the device's closest-hit kernel, u and v filled in on the host
        const size_t n = size_t(atol(argv[2]));
        const auto in = slurp(argv[3]);
        const char* q = in.data();
        const int32_t* kind = reinterpret_cast<const int32_t*>(q); q += 4 * n;
        const double* prm = reinterpret_cast<const double*>(q); q += 40 * n;
        const float* ro = reinterpret_cast<const float*>(q); q += 12 * n;
        const float* rdv = reinterpret_cast<const float*>(q); q += 12 * n;
        const double* tmax = reinterpret_cast<const double*>(q);
        std::vector<std::optional<hit_record>> recs;
        const std::shared_ptr<material> grey = std::make_shared<lambertian>(QVector3D(0.5, 0.5, 0.5));   // a world shape needs one to be uploaded
        for (size_t i = 0; i < n; ++i) {
            const double* m = prm + 5 * i;
            std::shared_ptr<shape> sh;
            if (kind[i] == 0) sh = std::make_shared<sphere>(QVector3D(float(m[0]), float(m[1]), float(m[2])), float(m[3]), grey);
            else if (kind[i] == 1) sh = std::make_shared<rectangle>(m[0], m[1], m[2], m[3], m[4], grey);
            else if (kind[i] == 2) sh = std::make_shared<xz_rect>(m[0], m[1], m[2], m[3], m[4], grey);
            else sh = std::make_shared<yz_rect>(m[0], m[1], m[2], m[3], m[4], grey);
            const Ray r(at3(ro, i), at3(rdv, i));
            recs.push_back(on_device ? sh->hit(r, 0.001, tmax[i]) : sh->hit_host(r, 0.001, tmax[i]));
        }
        Out o(argv[4]);
        for (auto& r : recs) { uint8_t h = r.has_value(); fwrite(&h, 1, 1, o.f); }
        for (auto& r : recs) o.f8(r ? r->t : 0.0);
        for (auto& r : recs) o.v3(r ? r->p : QVector3D());
        for (auto& r : recs) o.v3(r ? r->normal : QVector3D());
        for (auto& r : recs) { uint8_t h = r && r->front_face; fwrite(&h, 1, 1, o.f); }
        for (auto& r : recs) { o.f8(r ? r->u : 0.0); o.f8(r ? r->v : 0.0); }
    } else if (c == "materials") {
        const size_t n = size_t(atol(argv[2]));
        const auto in = slurp(argv[3]);
        const char* q = in.data();
        const float* d = reinterpret_cast<const float*>(q); q += 12 * n;
        const float* nn = reinterpret_cast<const float*>(q); q += 12 * n;
        const float* pp = reinterpret_cast<const float*>(q); q += 12 * n;
        const uint8_t* ff = reinterpret_cast<const uint8_t*>(q); q += n;
        const float* prm = reinterpret_cast<const float*>(q); q += 20 * n;
        const double* ir = reinterpret_cast<const double*>(q); q += 8 * n;
        const float* sdv = reinterpret_cast<const float*>(q); q += 12 * n;
        const float* prv = reinterpret_cast<const float*>(q);
        std::shared_ptr<shape> lightS = std::make_shared<sphere>(QVector3D(190, 90, 190), 90, nullptr);      // rayview.cpp:49
        std::shared_ptr<shape> lightXZ = std::make_shared<xz_rect>(213, 343, 227, 332, 554, nullptr);
        std::shared_ptr<shape> lightXY = std::make_shared<rectangle>(3, 5, 1, 3, -2, nullptr);
        std::shared_ptr<shape> lightYZ = std::make_shared<yz_rect>(0, 2, 0, 2, 5, nullptr);
        std::vector<QVector3D> met_o, met_d, met_a, die_d, die_a, lam_a, cos_gen, gen[4], mix_gen;
        std::vector<double> spdf, cos_val, val[4], mix_val;
        calc::gen.seed(21); calc::gen01.seed(22); calc::gen11.seed(23);
        for (size_t i = 0; i < n; ++i) {
            const QVector3D dir = at3(d, i), p = at3(pp, i), sd = at3(sdv, i), pr = at3(prv, i);
            const QVector3D alb(prm[5 * i + 1], prm[5 * i + 2], prm[5 * i + 3]);
            hit_record rec;
            rec.p = p; rec.normal = at3(nn, i); rec.front_face = ff[i] != 0; rec.t = 1.0; rec.u = 0; rec.v = 0;
            const Ray rin(p - dir, dir);
            { metal m(alb, prm[5 * i]); scatter_record s; m.scatter(rin, rec, s);
              met_o.push_back(s.specular_ray.pointOfOrigin); met_d.push_back(s.specular_ray.direction); met_a.push_back(s.attenuation); }
            { dielectric m(ir[i]); scatter_record s; m.scatter(rin, rec, s);
              die_d.push_back(s.specular_ray.direction); die_a.push_back(s.attenuation); }
            { lambertian m(alb); scatter_record s; m.scatter(rin, rec, s);
              lam_a.push_back(s.attenuation);
              spdf.push_back(m.scattering_pdf(rin, rec, Ray(p, sd)));
              cos_val.push_back(s.pdf_ptr->value(sd));
              cos_gen.push_back(s.pdf_ptr->generate());
              hittable_pdf hs(lightS, p);
              const QVector3D gs = hs.generate();
              gen[0].push_back(gs); val[0].push_back(hs.value(i % 2 ? gs : sd));
              hittable_pdf hr(lightXZ, pr);
              const QVector3D gr = hr.generate();
              gen[1].push_back(gr); val[1].push_back(hr.value(i % 2 ? gr : sd));
              hittable_pdf hxy(lightXY, p);
              gen[2].push_back(hxy.generate()); val[2].push_back(hxy.value(i % 2 ? QVector3D(4, 2, -2) - p : sd));
              hittable_pdf hyz(lightYZ, p);
              gen[3].push_back(hyz.generate()); val[3].push_back(hyz.value(sd));
              mixture_pdf mp(&hs, s.pdf_ptr.get());
              const QVector3D gm = mp.generate();
              mix_gen.push_back(gm); mix_val.push_back(mp.value(gm)); }
        }
        Out o(argv[4]);
        for (auto* v : {&met_o, &met_d, &met_a, &die_d, &die_a, &lam_a}) for (auto& x : *v) o.v3(x);
        for (double x : spdf) o.f8(x);
        for (double x : cos_val) o.f8(x);
        for (auto& x : cos_gen) o.v3(x);
        for (int k = 0; k < 4; ++k) { for (auto& x : gen[k]) o.v3(x); for (double x : val[k]) o.f8(x); }
        for (auto& x : mix_gen) o.v3(x);
        for (double x : mix_val) o.f8(x);
    } else {
        return 2;
    }
    return 0;
}
