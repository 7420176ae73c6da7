// TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
//
// Driver around the *unmodified* reference sources (/root/reference/{calc,hitPosition}.cpp and
// the headers they include). It is compiled in the build container by oracle/build_ref.sh into
// oracle/_ref/ref_harness (git-ignored) and is used to
//   (1) emit the golden vectors committed under tests/golden/ (see oracle/make_golden.py), and
//   (2) cross-check oracle/rt_oracle.cpp (the Qt-free CPU restatement) bit for bit.
// No reference source text is copied here: this file only *calls* the reference's public
// functions and classes. Private camera members are read with -fno-access-control (a compiler
// flag, not a source edit).
//
// Output container ("RTFX"): a sequence of records, each a text header line
//     <name> <dtype:f4|f8|i4|u1> <ndim> <d0> ... <dn-1>\n
// followed by the raw little-endian array bytes.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <map>
#include <random>
#include <sstream>
#include <string>

#include <calc.h>
#include <hitPosition.h>
#include <shapes.h>
#include <camera.h>

namespace {

struct Out {
    FILE* f;
    explicit Out(const char* path) : f(fopen(path, "wb")) {
        if (!f) { perror(path); exit(2); }
    }
    ~Out() { fclose(f); }
    void put(const std::string& name, const char* dt, const void* data, size_t elt,
             std::initializer_list<size_t> shape) {
        size_t n = 1;
        fprintf(f, "%s %s %zu", name.c_str(), dt, shape.size());
        for (size_t d : shape) { fprintf(f, " %zu", d); n *= d; }
        fprintf(f, "\n");
        fwrite(data, elt, n, f);
    }
    void f4(const std::string& n, const std::vector<float>& v, std::initializer_list<size_t> s) { put(n, "f4", v.data(), 4, s); }
    void f8(const std::string& n, const std::vector<double>& v, std::initializer_list<size_t> s) { put(n, "f8", v.data(), 8, s); }
    void i4(const std::string& n, const std::vector<int32_t>& v, std::initializer_list<size_t> s) { put(n, "i4", v.data(), 4, s); }
    void u1(const std::string& n, const std::vector<uint8_t>& v, std::initializer_list<size_t> s) { put(n, "u1", v.data(), 1, s); }
};

void push3(std::vector<float>& v, const QVector3D& q) { v.push_back(q.x()); v.push_back(q.y()); v.push_back(q.z()); }

void seed_all(unsigned a, unsigned b, unsigned c) {
    calc::gen.seed(a);
    calc::gen01.seed(b);
    calc::gen11.seed(c);
}

// ----------------------------------------------------------------------------------------------
// Scene file (text, hex floats) -> reference objects. Written by ray_amd/scenes.py.
// ----------------------------------------------------------------------------------------------
struct Scene {
    std::vector<std::shared_ptr<material>> mats;
    std::vector<std::shared_ptr<shape>> world;
    std::shared_ptr<shape> light;
    QVector3D from, at, up, background;
    double focus = 1, vfov = 40, aspect = 16.0 / 9.0, aperture = 0;
};

double rd(std::istream& in) {
    std::string s;
    in >> s;
    return strtod(s.c_str(), nullptr);   // accepts C99 hex floats
}
QVector3D rd3(std::istream& in) {
    float x = (float)rd(in), y = (float)rd(in), z = (float)rd(in);
    return QVector3D(x, y, z);
}

std::shared_ptr<shape> read_shape(const std::string& kind, std::istream& in, Scene& sc, bool with_mat) {
    auto mat = [&]() -> std::shared_ptr<material> {
        if (!with_mat) return std::shared_ptr<material>();
        int m; in >> m;
        return sc.mats.at(m);
    };
    if (kind == "sphere") {
        QVector3D c = rd3(in);
        float r = (float)rd(in);
        return std::make_shared<sphere>(c, r, mat());
    }
    if (kind == "xy_rect" || kind == "xz_rect" || kind == "yz_rect") {
        double a0 = rd(in), a1 = rd(in), b0 = rd(in), b1 = rd(in), k = rd(in);
        auto m = mat();
        if (kind == "xy_rect") return std::make_shared<rectangle>(a0, a1, b0, b1, k, m);
        if (kind == "xz_rect") return std::make_shared<xz_rect>(a0, a1, b0, b1, k, m);
        return std::make_shared<yz_rect>(a0, a1, b0, b1, k, m);
    }
    if (kind == "box") {
        QVector3D p0 = rd3(in), p1 = rd3(in);
        return std::make_shared<box>(p0, p1, mat());
    }
    if (kind == "medium_sphere" || kind == "medium_box") {   // constant_medium(boundary, density, colour): shapes.h:383-388
        std::shared_ptr<shape> b;
        if (kind == "medium_sphere") { QVector3D c = rd3(in); float r = (float)rd(in); b = std::make_shared<sphere>(c, r, std::shared_ptr<material>()); }
        else { QVector3D p0 = rd3(in), p1 = rd3(in); b = std::make_shared<box>(p0, p1, std::shared_ptr<material>()); }
        double d = rd(in);
        QVector3D col = rd3(in);
        return std::make_shared<constant_medium>(b, d, col);
    }
    fprintf(stderr, "unknown shape kind '%s'\n", kind.c_str());
    exit(2);
}

Scene load_scene(const char* path) {
    std::ifstream in(path);
    if (!in) { perror(path); exit(2); }
    Scene sc;
    std::string tok;
    while (in >> tok) {
        if (tok == "rtscene") { int v; in >> v; }
        else if (tok == "lambertian") sc.mats.push_back(std::make_shared<lambertian>(rd3(in)));
        else if (tok == "metal") { QVector3D a = rd3(in); double f = rd(in); sc.mats.push_back(std::make_shared<metal>(a, f)); }
        else if (tok == "dielectric") sc.mats.push_back(std::make_shared<dielectric>(rd(in)));
        else if (tok == "diffuse_light") sc.mats.push_back(std::make_shared<diffuse_light>(rd3(in)));
        else if (tok == "light") { std::string k; in >> k; sc.light = read_shape(k, in, sc, false); }
        else if (tok == "camera") {
            sc.from = rd3(in); sc.at = rd3(in); sc.up = rd3(in);
            sc.focus = rd(in); sc.vfov = rd(in); sc.aspect = rd(in); sc.aperture = rd(in);
        }
        else if (tok == "background") sc.background = rd3(in);
        else sc.world.push_back(read_shape(tok, in, sc, true));
    }
    return sc;
}

// Closest-primitive id by driving shape::hit exactly the way hitFromList does
// (hitPosition.cpp:6-22), and checking the result against hitFromList itself.
int scan_id(const std::vector<std::shared_ptr<shape>>& list, const Ray& r, double t_min, double t_max,
            std::optional<hit_record>* out = nullptr) {
    int id = -1;
    double closest = t_max;
    std::optional<hit_record> best;
    for (size_t i = 0; i < list.size(); ++i) {
        auto h = list[i]->hit(r, t_min, closest);
        if (h.has_value()) { closest = h->t; id = (int)i; best = h; }
    }
    auto ref = hitFromList(list, r, t_min, t_max);
    if (ref.has_value() != best.has_value() ||
        (ref.has_value() && (memcmp(&ref->t, &best->t, 8) != 0 || ref->mat_ptr != best->mat_ptr))) {
        fprintf(stderr, "scan_id disagrees with hitFromList\n");
        exit(3);
    }
    if (out) *out = best;
    return id;
}

// ----------------------------------------------------------------------------------------------
int cmd_rng(const char* outp) {
    Out o(outp);
    const unsigned seeds[4] = {0u, 1u, 2u, 12345u};
    const size_t N = 4096;
    for (unsigned s : seeds) {
        std::vector<float> a(N), b(N), c(N), d(N);
        std::vector<int32_t> w(N), h(N);
        seed_all(s, s, s);
        for (size_t i = 0; i < N; ++i) a[i] = calc::random_double01();
        for (size_t i = 0; i < N; ++i) b[i] = calc::random_double11();
        for (size_t i = 0; i < N; ++i) c[i] = calc::random_double(0.0f, 1.0f);
        for (size_t i = 0; i < N; ++i) d[i] = calc::random_double(3.0f, 5.0f);
        for (size_t i = 0; i < N; ++i) w[i] = calc::random_doubleW();
        for (size_t i = 0; i < N; ++i) h[i] = calc::random_doubleH();
        std::string p = "seed" + std::to_string(s) + "_";
        o.f4(p + "u01", a, {N});
        o.f4(p + "u11", b, {N});
        o.f4(p + "gen_0_1", c, {N});
        o.f4(p + "gen_3_5", d, {N});
        o.i4(p + "W", w, {N});
        o.i4(p + "H", h, {N});
    }
    // derived samplers (calc.cpp:54-72, calc.h:62-86)
    {
        const size_t M = 512;
        std::vector<float> sph, disk, cosd, tosph;
        seed_all(7, 8, 9);
        for (size_t i = 0; i < M; ++i) push3(sph, calc::random_in_unit_sphere());
        for (size_t i = 0; i < M; ++i) push3(disk, calc::random_in_unit_disk());
        for (size_t i = 0; i < M; ++i) push3(cosd, calc::random_cosine_direction());
        for (size_t i = 0; i < M; ++i) push3(tosph, calc::random_to_sphere(90.0, 40000.0 + 100.0 * i));
        o.f4("in_unit_sphere_s9", sph, {M, 3});
        o.f4("in_unit_disk_s9", disk, {M, 3});
        o.f4("cosine_direction_s8", cosd, {M, 3});
        o.f4("to_sphere_s8", tosph, {M, 3});
    }
    return 0;
}

int cmd_vec3(const char* outp) {
    Out o(outp);
    const size_t N = 10000;
    std::mt19937 g(42);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    std::uniform_int_distribution<int> e(-20, 20);
    std::vector<float> a, b, lsq(N), len(N), nrm, dot(N), crs, dist_pl;
    for (size_t i = 0; i < N; ++i) {
        float sa = std::ldexp(1.0f, e(g)), sb = std::ldexp(1.0f, e(g));
        QVector3D va(u(g) * sa, u(g) * sa, u(g) * sa), vb(u(g) * sb, u(g) * sb, u(g) * sb);
        if (i % 7 == 0) va = va.normalized();              // near-unit inputs (1e-12 shortcut)
        if (i % 97 == 0) va = QVector3D(0, 0, 0);
        if (i % 101 == 0) va = QVector3D(1e-20f * u(g), 1e-21f, 0.0f);   // |len|<=1e-12 -> zero
        push3(a, va); push3(b, vb);
        lsq[i] = va.lengthSquared();
        len[i] = va.length();
        push3(nrm, va.normalized());
        dot[i] = QVector3D::dotProduct(va, vb);
        push3(crs, QVector3D::crossProduct(va, vb));
    }
    o.f4("a", a, {N, 3}); o.f4("b", b, {N, 3});
    o.f4("lengthSquared", lsq, {N}); o.f4("length", len, {N});
    o.f4("normalized", nrm, {N, 3}); o.f4("dot", dot, {N}); o.f4("cross", crs, {N, 3});
    return 0;
}

void dump_camera(Out& o, const std::string& p, const camera& c) {
    std::vector<float> v;
    push3(v, c.origin); push3(v, c.lower_left_corner); push3(v, c.horizontal); push3(v, c.vertical);
    push3(v, c.u); push3(v, c.v); push3(v, c.w);
    o.f4(p + "frame", v, {7, 3});
    o.f8(p + "lens_radius", {c.lens_radius}, {1});
}

int cmd_camera(const char* outp) {
    Out o(outp);
    struct Cfg { QVector3D from, at, up; double focus, vfov, aspect, aperture; };
    QVector3D f0(0, 0, 1), a0(0, 0, -1);
    std::vector<Cfg> cfgs = {
        {f0, a0, QVector3D(0, 1, 0), (double)(f0 - a0).length(), 40.0, (double)img::aspect_ratio, 0.1},  // rayview.cpp:18-27
        {QVector3D(13, 2, 3), QVector3D(0, 0, 0), QVector3D(0, 1, 0), 10.0, 20.0, 16.0 / 9.0, 0.1},
        {QVector3D(278, 278, -800), QVector3D(278, 278, 0), QVector3D(0, 1, 0), 10.0, 40.0, 1.0, 0.0},
        {QVector3D(-2, 2, 1), QVector3D(0, 0, -1), QVector3D(0.1f, 1, 0.2f), 3.4, 90.0, 2.0, 2.0},
    };
    std::vector<double> params;
    std::vector<float> vecs;
    for (size_t i = 0; i < cfgs.size(); ++i) {
        const Cfg& c = cfgs[i];
        camera cam(c.from, c.at, c.up, c.focus, c.vfov, c.aspect, c.aperture);
        std::string p = "cam" + std::to_string(i) + "_";
        dump_camera(o, p, cam);
        push3(vecs, c.from); push3(vecs, c.at); push3(vecs, c.up);
        params.insert(params.end(), {c.focus, c.vfov, c.aspect, c.aperture});
        // seeded get_ray
        const size_t N = 1000;
        std::mt19937 g(100 + (unsigned)i);
        std::uniform_real_distribution<float> u(0.0f, 1.0f);
        std::vector<float> st, ro, rdv, dl;
        seed_all(11, 12, 13 + (unsigned)i);
        for (size_t k = 0; k < N; ++k) {
            float s = u(g), t = u(g);
            Ray r = cam.get_ray(s, t);
            st.push_back(s); st.push_back(t);
            push3(ro, r.pointOfOrigin); push3(rdv, r.direction); dl.push_back(r.DirlengthSquared);
        }
        o.f4(p + "st", st, {N, 2}); o.f4(p + "ray_o", ro, {N, 3}); o.f4(p + "ray_d", rdv, {N, 3});
        o.f4(p + "ray_dl2", dl, {N});
        o.i4(p + "seed11", {(int32_t)(13 + i)}, {1});
    }
    o.f4("cfg_vecs", vecs, {cfgs.size(), 3, 3});
    o.f8("cfg_params", params, {cfgs.size(), 4});
    return 0;
}

void push_rec(const std::optional<hit_record>& h, std::vector<uint8_t>& has, std::vector<double>& t,
              std::vector<float>& p, std::vector<float>& n, std::vector<uint8_t>& ff, std::vector<double>& uv) {
    has.push_back(h.has_value());
    if (h.has_value()) {
        t.push_back(h->t); push3(p, h->p); push3(n, h->normal); ff.push_back(h->front_face);
        uv.push_back(h->u); uv.push_back(h->v);
    } else {
        t.push_back(0); push3(p, QVector3D()); push3(n, QVector3D()); ff.push_back(0); uv.push_back(0); uv.push_back(0);
    }
}

int cmd_shapehits(const char* outp) {
    Out o(outp);
    const size_t N = 24000;
    std::mt19937 g(4242);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    std::uniform_real_distribution<float> u01(0.0f, 1.0f);
    // spheres -------------------------------------------------------------
    {
        std::vector<float> ro, rdv, prm;
        std::vector<double> tm, t, uv;
        std::vector<uint8_t> has, ff;
        std::vector<float> p, n;
        for (size_t i = 0; i < N; ++i) {
            QVector3D c(4 * u(g), 4 * u(g), 4 * u(g));
            float r = 0.05f + 2.0f * u01(g);
            QVector3D org(6 * u(g), 6 * u(g), 6 * u(g));
            QVector3D dir;
            switch (i % 8) {
            case 0: dir = QVector3D(u(g), u(g), u(g)); break;                          // random, mostly miss
            case 1: org = c + QVector3D(u(g), u(g), u(g)) * (0.5f * r);               // origin inside
                    dir = QVector3D(u(g), u(g), u(g)); break;
            case 2: {                                                                // exactly tangent, axis aligned
                    c = QVector3D(std::round(4 * u(g)), std::round(4 * u(g)), std::round(4 * u(g)));
                    r = 0.5f * (1 + (int)(3 * u01(g)));
                    org = QVector3D(c.x(), c.y() - r, c.z() + 4.0f); dir = QVector3D(0, 0, -1); break; }
            case 3: {                                                                // origin on surface (t ~ t_min)
                    QVector3D nn = QVector3D(u(g), u(g), u(g)).normalized();
                    org = c + nn * r; dir = QVector3D(u(g), u(g), u(g)); break; }
            default: {                                                               // aimed at the sphere
                    QVector3D tgt = c + QVector3D(u(g), u(g), u(g)) * r;
                    dir = (tgt - org) * (0.1f + 3.0f * u01(g)); break; }
            }
            double tmax = (i % 5 == 0) ? (double)(20.0f * u01(g)) : calc::infinity;
            sphere s(c, r, std::shared_ptr<material>());
            Ray ray(org, dir);
            auto h = s.hit(ray, 0.001, tmax);
            push3(ro, org); push3(rdv, dir); push3(prm, c); prm.push_back(r); tm.push_back(tmax);
            push_rec(h, has, t, p, n, ff, uv);
        }
        o.f4("sph_ray_o", ro, {N, 3}); o.f4("sph_ray_d", rdv, {N, 3}); o.f4("sph_prm", prm, {N, 4});
        o.f8("sph_tmax", tm, {N}); o.u1("sph_has", has, {N}); o.f8("sph_t", t, {N});
        o.f4("sph_p", p, {N, 3}); o.f4("sph_n", n, {N, 3}); o.u1("sph_ff", ff, {N}); o.f8("sph_uv", uv, {N, 2});
    }
    // rects: axis 0 = xy (k on z), 1 = xz (k on y), 2 = yz (k on x) --------
    {
        std::vector<float> ro, rdv;
        std::vector<double> prm, tm, t, uv;
        std::vector<int32_t> ax;
        std::vector<uint8_t> has, ff;
        std::vector<float> p, n;
        for (size_t i = 0; i < N; ++i) {
            int axis = (int)(i % 3);
            double a0 = 3 * u(g), a1 = a0 + 0.1 + 3 * u01(g), b0 = 3 * u(g), b1 = b0 + 0.1 + 3 * u01(g), k = 3 * u(g);
            if (i % 11 == 0) { a0 = std::round(a0); a1 = a0 + 2; b0 = std::round(b0); b1 = b0 + 1; k = std::round(k); }
            QVector3D org(5 * u(g), 5 * u(g), 5 * u(g));
            QVector3D dir(u(g), u(g), u(g));
            if (i % 4 == 1) {   // aimed into the rectangle
                float aa = (float)(a0 + (a1 - a0) * u01(g)), bb = (float)(b0 + (b1 - b0) * u01(g));
                QVector3D tgt = axis == 0 ? QVector3D(aa, bb, (float)k) : axis == 1 ? QVector3D(aa, (float)k, bb) : QVector3D((float)k, aa, bb);
                dir = (tgt - org) * (0.2f + 2 * u01(g));
            }
            if (i % 11 == 0 && i % 4 == 1) {   // exactly on an edge
                QVector3D tgt = axis == 0 ? QVector3D((float)a0, (float)b1, (float)k) : axis == 1 ? QVector3D((float)a0, (float)k, (float)b1) : QVector3D((float)k, (float)a0, (float)b1);
                org = tgt + QVector3D(axis == 2 ? 2.f : 0.f, axis == 1 ? 2.f : 0.f, axis == 0 ? 2.f : 0.f);
                dir = tgt - org;
            }
            if (i % 53 == 0) { if (axis == 0) dir.setZ(0); else if (axis == 1) dir.setY(0); else dir.setX(0); }   // parallel: t = +-inf / nan
            double tmax = (i % 5 == 0) ? (double)(10.0f * u01(g)) : calc::infinity;
            std::shared_ptr<shape> s;
            if (axis == 0) s = std::make_shared<rectangle>(a0, a1, b0, b1, k, std::shared_ptr<material>());
            else if (axis == 1) s = std::make_shared<xz_rect>(a0, a1, b0, b1, k, std::shared_ptr<material>());
            else s = std::make_shared<yz_rect>(a0, a1, b0, b1, k, std::shared_ptr<material>());
            Ray ray(org, dir);
            auto h = s->hit(ray, 0.001, tmax);
            ax.push_back(axis); push3(ro, org); push3(rdv, dir);
            prm.insert(prm.end(), {a0, a1, b0, b1, k}); tm.push_back(tmax);
            push_rec(h, has, t, p, n, ff, uv);
        }
        o.i4("rect_axis", ax, {N}); o.f4("rect_ray_o", ro, {N, 3}); o.f4("rect_ray_d", rdv, {N, 3});
        o.f8("rect_prm", prm, {N, 5}); o.f8("rect_tmax", tm, {N}); o.u1("rect_has", has, {N}); o.f8("rect_t", t, {N});
        o.f4("rect_p", p, {N, 3}); o.f4("rect_n", n, {N, 3}); o.u1("rect_ff", ff, {N}); o.f8("rect_uv", uv, {N, 2});
    }
    return 0;
}

// Material / pdf known-answer vectors with seeded streams.
int cmd_materials(const char* outp) {
    Out o(outp);
    const size_t N = 4000;
    std::mt19937 g(777);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    std::uniform_real_distribution<float> u01(0.0f, 1.0f);

    std::vector<float> in_d, in_n, in_p, prm;
    std::vector<uint8_t> in_ff;
    std::vector<float> met_o, met_d, met_a, die_d, die_a, lam_a, lam_sd;
    std::vector<double> lam_spdf, cos_val, sl_val, rl_val, mix_val;
    std::vector<float> cos_gen, sl_gen, rl_gen, mix_gen, refl, refr;
    std::vector<double> refr_eta, irs;

    auto lightS = std::make_shared<sphere>(QVector3D(190, 90, 190), 90, std::shared_ptr<material>());   // rayview.cpp:49
    std::shared_ptr<shape> lightXZ = std::make_shared<xz_rect>(213, 343, 227, 332, 554, std::shared_ptr<material>());
    std::shared_ptr<shape> lightXY = std::make_shared<rectangle>(3, 5, 1, 3, -2, std::shared_ptr<material>());
    std::shared_ptr<shape> lightYZ = std::make_shared<yz_rect>(0, 2, 0, 2, 5, std::shared_ptr<material>());
    std::vector<double> xy_val, yz_val;
    std::vector<float> xy_gen, yz_gen;

    seed_all(21, 22, 23);
    for (size_t i = 0; i < N; ++i) {
        QVector3D d(u(g), u(g), u(g));
        d *= (0.2f + 3 * u01(g));
        QVector3D nrm = QVector3D(u(g), u(g), u(g)).normalized();
        if (i % 9 == 0) nrm = QVector3D(0.95f, 0.1f * u(g), 0.1f * u(g)).normalized();     // |w.x|>0.9 onb branch
        QVector3D p(3 * u(g), 3 * u(g), 3 * u(g));
        bool front = QVector3D::dotProduct(d, nrm) < 0.0f;
        QVector3D n_face = front ? nrm : -nrm;
        float fuzz = (i % 3 == 0) ? 0.0f : 0.5f * u01(g);
        if (i % 50 == 0) fuzz = 1.5f;   // clamps to 1
        QVector3D alb(u01(g), u01(g), u01(g));
        double ir = (i % 4 == 0) ? 1.5 : 1.0 + 1.5 * u01(g);
        push3(in_d, d); push3(in_n, n_face); push3(in_p, p); in_ff.push_back(front);
        prm.push_back(fuzz); push3(prm, alb); prm.push_back((float)ir); irs.push_back(ir);

        hit_record rec;
        rec.p = p; rec.normal = n_face; rec.front_face = front; rec.t = 1.0; rec.u = 0; rec.v = 0;
        Ray rin(p - d, d);

        { metal m(alb, fuzz); scatter_record s; m.scatter(rin, rec, s);
          push3(met_o, s.specular_ray.pointOfOrigin); push3(met_d, s.specular_ray.direction); push3(met_a, s.attenuation); }
        { dielectric m(ir); scatter_record s; m.scatter(rin, rec, s);
          push3(die_d, s.specular_ray.direction); push3(die_a, s.attenuation); }
        { lambertian m(alb); scatter_record s; m.scatter(rin, rec, s);
          push3(lam_a, s.attenuation);
          QVector3D sd(u(g), u(g), u(g)); sd *= (0.1f + 2 * u01(g));
          push3(lam_sd, sd);
          lam_spdf.push_back(m.scattering_pdf(rin, rec, Ray(p, sd)));
          cos_val.push_back(s.pdf_ptr->value(sd));
          push3(cos_gen, s.pdf_ptr->generate());
          // light pdfs evaluated from p
          hittable_pdf hs(lightS, p);
          QVector3D gs = hs.generate();
          push3(sl_gen, gs);
          sl_val.push_back(hs.value(i % 2 ? gs : sd));
          QVector3D pr(278 + 200 * u(g), 100 + 200 * u01(g), 278 + 200 * u(g));
          hittable_pdf hr(lightXZ, pr);
          QVector3D gr = hr.generate();
          push3(rl_gen, gr); push3(rl_gen, pr);
          rl_val.push_back(hr.value(i % 2 ? gr : sd));
          hittable_pdf hxy(lightXY, p);
          QVector3D gxy = hxy.generate();
          push3(xy_gen, gxy);
          xy_val.push_back(hxy.value(i % 2 ? QVector3D(4, 2, -2) - p : sd));
          hittable_pdf hyz(lightYZ, p);
          push3(yz_gen, hyz.generate());
          yz_val.push_back(hyz.value(sd));
          mixture_pdf mp(&hs, s.pdf_ptr.get());
          QVector3D gm = mp.generate();
          push3(mix_gen, gm);
          mix_val.push_back(mp.value(gm));
        }
        { QVector3D ud = d.normalized();
          push3(refl, calc::reflect(ud, n_face));
          double eta = front ? 1.0 / ir : ir;
          refr_eta.push_back(eta);
          push3(refr, calc::refract(ud, n_face, eta)); }
    }
    o.f4("in_d", in_d, {N, 3}); o.f4("in_n", in_n, {N, 3}); o.f4("in_p", in_p, {N, 3}); o.u1("in_ff", in_ff, {N});
    o.f4("prm", prm, {N, 5});
    o.f4("metal_o", met_o, {N, 3}); o.f4("metal_d", met_d, {N, 3}); o.f4("metal_att", met_a, {N, 3});
    o.f4("diel_d", die_d, {N, 3}); o.f4("diel_att", die_a, {N, 3});
    o.f4("lamb_att", lam_a, {N, 3}); o.f4("lamb_sd", lam_sd, {N, 3}); o.f8("lamb_spdf", lam_spdf, {N});
    o.f8("cos_val", cos_val, {N}); o.f4("cos_gen", cos_gen, {N, 3});
    o.f4("sphlight_gen", sl_gen, {N, 3}); o.f8("sphlight_val", sl_val, {N});
    o.f4("xzlight_gen_and_o", rl_gen, {N, 2, 3}); o.f8("xzlight_val", rl_val, {N});
    o.f4("xylight_gen", xy_gen, {N, 3}); o.f8("xylight_val", xy_val, {N});
    o.f4("yzlight_gen", yz_gen, {N, 3}); o.f8("yzlight_val", yz_val, {N});
    o.f4("mix_gen", mix_gen, {N, 3}); o.f8("mix_val", mix_val, {N});
    o.f4("reflect", refl, {N, 3}); o.f4("refract", refr, {N, 3}); o.f8("refract_eta", refr_eta, {N});
    o.f8("ir", irs, {N});
    o.i4("seeds", {21, 22, 23}, {3});

    // tonemap (calc.cpp:133-154)
    {
        std::vector<float> tin, tout;
        std::vector<int32_t> ns;
        for (size_t i = 0; i < 2000; ++i) {
            int n = 1 + (int)(i % 64);
            QVector3D c(n * 1.3f * u01(g), n * u01(g), n * 0.2f * u01(g));
            if (i % 100 == 0) c.setX(0);
            if (i % 101 == 0) c.setY(1e6f);
            push3(tin, c); ns.push_back(n); push3(tout, calc::rgbPerSamples(c, n));
        }
        o.f4("tone_in", tin, {2000, 3}); o.i4("tone_n", ns, {2000}); o.f4("tone_out", tout, {2000, 3});
    }
    return 0;
}

// constant_medium::hit known answers (shapes.h:390-444): seeded gen01, sphere and box boundaries, rays through,
// inside, grazing and missing, several densities and t ranges.
int cmd_medium(const char* outp) {
    Out o(outp);
    const size_t N = 6000;
    std::mt19937 g(9191);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    std::uniform_real_distribution<float> u01(0.0f, 1.0f);
    std::vector<float> ro, rdv, bnd;
    std::vector<double> dens, tmin, tmax, t;
    std::vector<int32_t> kind;
    std::vector<uint8_t> has, ff;
    std::vector<float> p, n;
    seed_all(31, 32, 33);
    for (size_t i = 0; i < N; ++i) {
        const bool is_box = (i % 2) == 1;
        QVector3D c(3 * u(g), 3 * u(g), 3 * u(g));
        float r = 0.2f + 1.5f * u01(g);
        QVector3D half(0.2f + 1.5f * u01(g), 0.2f + 1.5f * u01(g), 0.2f + 1.5f * u01(g));
        QVector3D p0 = c - half, p1 = c + half;
        std::shared_ptr<shape> b;
        if (is_box) b = std::make_shared<box>(p0, p1, std::shared_ptr<material>());
        else b = std::make_shared<sphere>(c, r, std::shared_ptr<material>());
        double d = (i % 7 == 0) ? 100.0 : (i % 7 == 1 ? 0.01 : 0.05 + 3.0 * u01(g));
        constant_medium m(b, d, QVector3D(1, 1, 1));
        QVector3D org(5 * u(g), 5 * u(g), 5 * u(g));
        QVector3D dir;
        switch (i % 5) {
        case 0: dir = QVector3D(u(g), u(g), u(g)); break;                                       // random
        case 1: org = c + QVector3D(u(g), u(g), u(g)) * 0.15f; dir = QVector3D(u(g), u(g), u(g)); break;   // origin inside
        default: dir = (c + QVector3D(u(g), u(g), u(g)) * 0.5f * (is_box ? 0.2f : r) - org) * (0.2f + 3 * u01(g)); break;   // aimed through
        }
        if (is_box && i % 31 == 1) { dir.setX(0); }                                             // parallel to two faces
        double t0 = (i % 3 == 0) ? 0.001 : (i % 3 == 1 ? -2.0 : (double)(2.0f * u01(g)));
        double t1 = (i % 4 == 0) ? (double)(1.0f + 8.0f * u01(g)) : calc::infinity;
        auto h = m.hit(Ray(org, dir), t0, t1);
        kind.push_back(is_box ? 1 : 0);
        if (is_box) { push3(bnd, p0); push3(bnd, p1); } else { push3(bnd, c); bnd.push_back(r); bnd.push_back(0); bnd.push_back(0); }
        dens.push_back(d); push3(ro, org); push3(rdv, dir); tmin.push_back(t0); tmax.push_back(t1);
        has.push_back(h.has_value());
        if (h.has_value()) { t.push_back(h->t); push3(p, h->p); push3(n, h->normal); ff.push_back(h->front_face); }
        else { t.push_back(0); push3(p, QVector3D()); push3(n, QVector3D()); ff.push_back(0); }
    }
    o.i4("kind", kind, {N}); o.f4("bound", bnd, {N, 6}); o.f8("density", dens, {N}); o.f4("ray_o", ro, {N, 3}); o.f4("ray_d", rdv, {N, 3});
    o.f8("t_min", tmin, {N}); o.f8("t_max", tmax, {N}); o.u1("has", has, {N}); o.f8("t", t, {N}); o.f4("p", p, {N, 3});
    o.f4("n", n, {N, 3}); o.u1("ff", ff, {N});
    o.i4("seeds", {31, 32, 33}, {3});
    return 0;
}

camera make_cam(const Scene& sc) { return camera(sc.from, sc.at, sc.up, sc.focus, sc.vfov, sc.aspect, sc.aperture); }

// Pixel-centre primary rays: u=(x+0.5)/W, v=(y+0.5)/H through camera::get_ray.
int cmd_primary(const char* scene, int W, int H, const char* outp) {
    Scene sc = load_scene(scene);
    camera cam = make_cam(sc);
    Out o(outp);
    std::vector<int32_t> id((size_t)W * H), side((size_t)W * H, -1);
    std::vector<double> tt((size_t)W * H, 0.0);
    seed_all(1, 2, 3);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float u = (x + 0.5f) / W, v = (y + 0.5f) / H;
            Ray r = cam.get_ray(u, v);
            std::optional<hit_record> rec;
            int i = scan_id(sc.world, r, calc::smallestVal, calc::infinity, &rec);
            id[(size_t)y * W + x] = i;
            if (i >= 0) {
                tt[(size_t)y * W + x] = rec->t;
                if (auto* b = dynamic_cast<box*>(sc.world[i].get()))
                    side[(size_t)y * W + x] = scan_id(b->sides, r, calc::smallestVal, calc::infinity);
            }
        }
    o.i4("prim_id", id, {(size_t)H, (size_t)W});
    o.i4("box_side", side, {(size_t)H, (size_t)W});
    o.f8("t", tt, {(size_t)H, (size_t)W});
    return 0;
}

// Rays of a render = calls of hitFromList from calc::ray_color (calc.cpp:99). The reference is compiled unmodified, so the
// count comes from the linker: build_ref.sh passes -Wl,--wrap=<mangled hitFromList>, which sends every call from another
// translation unit to the __wrap_ function below and keeps the reference's definition reachable as __real_.
#ifdef RT_HFL_SYM
}  // namespace (the wrapper needs external linkage)
static unsigned long long g_hit_from_list_calls = 0;
std::optional<hit_record> rt_real_hitFromList(const std::vector<std::shared_ptr<shape>>&, const Ray&, double, double) asm("__real_" RT_HFL_SYM);
std::optional<hit_record> rt_wrap_hitFromList(const std::vector<std::shared_ptr<shape>>& l, const Ray& r, double t0, double t1) asm("__wrap_" RT_HFL_SYM);
std::optional<hit_record> rt_wrap_hitFromList(const std::vector<std::shared_ptr<shape>>& l, const Ray& r, double t0, double t1) {
    ++g_hit_from_list_calls;
    return rt_real_hitFromList(l, r, t0, t1);
}
namespace {
#endif

// Full-frame render in row-major pixel order with the per-pixel contract of
// RayView::renderOneRay (rayview.cpp:79-84), RNG seeded (a,b,c).
int cmd_render(const char* scene, int W, int H, int spp, int depth, unsigned a, unsigned b, unsigned c,
               const char* outp) {
    Scene sc = load_scene(scene);
    camera cam = make_cam(sc);
    seed_all(a, b, c);
    std::vector<float> img;
    img.reserve((size_t)W * H * 3);
    auto t0 = std::chrono::steady_clock::now();
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            QVector3D sum(0, 0, 0);
            for (int s = 0; s < spp; ++s) {
                float u = (x + calc::random_double01()) / W;
                float v = (y + calc::random_double01()) / H;
                sum += calc::ray_color(cam.get_ray(u, v), sc.background, sc.world, sc.light, depth, false);
            }
            push3(img, sum);
        }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
#ifdef RT_HFL_SYM
    fprintf(stderr, "rays %llu\n", g_hit_from_list_calls);   // before the timing line, whose last field bench.py parses
#endif
    fprintf(stderr, "render %dx%d spp=%d depth=%d: %.3f s (%.3f Msamples/s)\n", W, H, spp, depth, sec,
            1e-6 * W * H * spp / sec);
    if (outp) {
        Out o(outp);
        o.f4("sum", img, {(size_t)H, (size_t)W, 3});
        o.i4("params", {W, H, spp, depth, (int32_t)a, (int32_t)b, (int32_t)c}, {7});
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s rng|vec3|camera|shapehits|materials|medium OUT\n"
                        "       %s primary SCENE W H OUT\n"
                        "       %s render SCENE W H SPP DEPTH S S01 S11 [OUT]\n", argv[0], argv[0], argv[0]);
        return 1;
    }
    std::string c = argv[1];
    if (c == "rng") return cmd_rng(argv[2]);
    if (c == "vec3") return cmd_vec3(argv[2]);
    if (c == "camera") return cmd_camera(argv[2]);
    if (c == "shapehits") return cmd_shapehits(argv[2]);
    if (c == "materials") return cmd_materials(argv[2]);
    if (c == "medium") return cmd_medium(argv[2]);
    if (c == "primary" && argc == 6) return cmd_primary(argv[2], atoi(argv[3]), atoi(argv[4]), argv[5]);
    if (c == "render" && argc >= 10)
        return cmd_render(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]),
                          (unsigned)atol(argv[7]), (unsigned)atol(argv[8]), (unsigned)atol(argv[9]),
                          argc > 10 ? argv[10] : nullptr);
    fprintf(stderr, "bad command\n");
    return 1;
}
