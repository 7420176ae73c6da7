// Every entry point of include/ray_amd.h called once or more from plain C++ on small inputs, for a run under the host-side
// AddressSanitizer (tools/asan_run.sh): the Python tests check the results, this checks that the library's HOST code
// (validation, flattening, uploads, workspaces that grow, the context's life cycle) touches no memory it should not.
// It also compares rt_render_noalloc / rt_render_bands / rt_render_pixels with rt_render bit for bit, as the tests do.
// Built by __graft_entry__.build() like the other callers (tests/test_gpu_parity.py runs it); the sanitizer build:
//   clang++ -fsanitize=address -I include ray_amd/host/abi_tour.cpp -L tools/asan -lray_amd -o tools/asan/abi_tour
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ray_amd.h"

#define OK(call)                                                                                  \
    do {                                                                                          \
        if ((call) != 0) {                                                                        \
            std::fprintf(stderr, "%s:%d %s failed: %s\n", __FILE__, __LINE__, #call, rt_last_error()); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
#define MUST_FAIL(call)                                                                           \
    do {                                                                                          \
        if ((call) == 0) {                                                                        \
            std::fprintf(stderr, "%s:%d %s should have been rejected\n", __FILE__, __LINE__, #call); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

struct SceneArrays {
    std::vector<int32_t> kind, mat, outer, side, mat_kind;
    std::vector<double> data, ir;
    std::vector<float> rgb, fuzz;
    RtScene s{};
    void prim(int k, int m, std::initializer_list<double> d) {
        kind.push_back(k); mat.push_back(m); outer.push_back(int(kind.size()) - 1); side.push_back(-1);
        size_t at = data.size();
        data.resize(at + RT_PRIM_STRIDE, 0.0);
        size_t i = 0;
        for (double v : d) data[at + i++] = v;
    }
    void material(int k, float r, float g, float b, float fz, double index) {
        mat_kind.push_back(k); rgb.insert(rgb.end(), {r, g, b}); fuzz.push_back(fz); ir.push_back(index);
    }
    const RtScene* finish(int light_kind, std::initializer_list<double> light) {
        s.n_prims = int(kind.size()); s.prim_kind = kind.data(); s.prim_mat = mat.data(); s.prim_outer = outer.data();
        s.prim_side = side.data(); s.prim_data = data.data();
        s.n_mats = int(mat_kind.size()); s.mat_kind = mat_kind.data(); s.mat_rgb = rgb.data(); s.mat_fuzz = fuzz.data(); s.mat_ir = ir.data();
        s.light_kind = light_kind;
        size_t i = 0;
        for (double v : light) s.light_data[i++] = v;
        s.background[0] = 0.7f; s.background[1] = 0.8f; s.background[2] = 1.0f;
        return &s;
    }
};

static uint32_t lcg(uint32_t& x) { return x = x * 1664525u + 1013904223u; }
static float unit(uint32_t& x) { return float(lcg(x) >> 8) / 16777216.0f; }

int main() {
    int count = 0;
    OK(rt_device_count(&count));
    MUST_FAIL(rt_init(count));   // no such device
    OK(rt_init(0));
    char info[256];
    OK(rt_device_info(info, sizeof info));
    MUST_FAIL(rt_device_info(info, 0));
    MUST_FAIL(rt_device_info(nullptr, 16));
    char tiny[2];
    OK(rt_device_info(tiny, 2));   // (truncated)
    std::printf("%s\n", info);

    // a scene with every primitive and material kind; a few hundred spheres and triangles so that the filter groups have batches
    SceneArrays sc;
    sc.material(RT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f, 0.0);
    sc.material(RT_MAT_METAL, 0.8f, 0.7f, 0.6f, 0.2f, 0.0);
    sc.material(RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 0.0f, 1.5);
    sc.material(RT_MAT_DIFFUSE_LIGHT, 4.0f, 4.0f, 4.0f, 0.0f, 0.0);
    sc.material(RT_MAT_ISOTROPIC, 0.9f, 0.9f, 0.9f, 0.0f, 0.0);
    sc.prim(RT_PRIM_SPHERE, 0, {0.0, -1000.0, 0.0, 1000.0});
    uint32_t x = 7u;
    for (int i = 0; i < 300; ++i)
        sc.prim(RT_PRIM_SPHERE, i % 3, {double(float(unit(x) * 16 - 8)), 0.2, double(float(unit(x) * 16 - 8)), double(0.2f)});
    for (int i = 0; i < 400; ++i) {
        float px = unit(x) * 8 - 4, py = unit(x) * 3, pz = unit(x) * 8 - 4;
        sc.prim(RT_PRIM_TRIANGLE, i % 3, {px, py, pz, px + 0.4f * unit(x), py + 0.4f * unit(x), pz, px, py + 0.3f * unit(x), pz + 0.4f * unit(x)});
    }
    sc.prim(RT_PRIM_XY_RECT, 1, {-2.0, 2.0, 0.0, 2.0, -5.0});
    sc.prim(RT_PRIM_XZ_RECT, 3, {-1.0, 1.0, -1.0, 1.0, 5.0});
    sc.prim(RT_PRIM_YZ_RECT, 0, {0.0, 2.0, -2.0, 2.0, 6.0});
    const RtScene* scene = sc.finish(RT_PRIM_XZ_RECT, {-1.0, 1.0, -1.0, 1.0, 5.0});
    {   // rejected scenes leave the context usable
        RtScene bad = *scene;
        bad.n_prims = -1;
        MUST_FAIL(rt_upload_scene(&bad));
        bad = *scene; bad.prim_kind = nullptr;
        MUST_FAIL(rt_upload_scene(&bad));
        std::vector<int32_t> m2 = sc.mat; m2[5] = 99;
        bad = *scene; bad.prim_mat = m2.data();
        MUST_FAIL(rt_upload_scene(&bad));
        MUST_FAIL(rt_upload_scene(nullptr));
    }
    OK(rt_upload_scene(scene));
    const float from[3] = {13.0f, 2.0f, 3.0f}, at[3] = {0.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
    RtCamera cam;
    OK(rt_camera_from_lookat(from, at, up, 10.0, 20.0, 16.0 / 9.0, 0.1, &cam));
    OK(rt_set_camera(&cam));
    MUST_FAIL(rt_set_camera(nullptr));

    const int W = 203, H = 77, SPP = 24, DEPTH = 9;
    const size_t npx = size_t(W) * H;
    float *a0 = nullptr, *a1 = nullptr;
    uint64_t* d_rays = nullptr;
    OK(rt_malloc(reinterpret_cast<void**>(&a0), npx * 16));
    OK(rt_malloc(reinterpret_cast<void**>(&a1), npx * 16));
    OK(rt_malloc(reinterpret_cast<void**>(&d_rays), 8));
    std::vector<float> h0(npx * 4), h1(npx * 4);
    auto same = [&](const char* what) {
        if (rt_memcpy_d2h(h0.data(), a0, npx * 16) != 0 || rt_memcpy_d2h(h1.data(), a1, npx * 16) != 0) return false;
        if (std::memcmp(h0.data(), h1.data(), npx * 16) != 0) { std::fprintf(stderr, "%s differs from rt_render\n", what); return false; }
        return true;
    };
    OK(rt_memset(d_rays, 0, 8));
    OK(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, 0, a0, d_rays));
    float ms; int launches;
    OK(rt_last_kernel_ms(&ms, &launches));
    uint64_t rays = 0;
    OK(rt_memcpy_d2h(&rays, d_rays, 8));
    std::printf("rt_render %dx%d %d spp: %llu rays, %.3f ms in %d launches\n", W, H, SPP, (unsigned long long)rays, ms, launches);

    // the knobs: every setting leaves the same bits
    const int limits[] = {0, 1, 64}, smins[] = {0, 3, 8};
    for (int k = 0; k < 3; ++k) {
        OK(rt_set_tail_limit(limits[k]));
        OK(rt_set_sample_slices(smins[k], 2 + 7 * k));
        OK(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, 0, a1, nullptr));
        if (!same("rt_render with other work units")) return 1;
    }
    OK(rt_set_tail_limit(1400));
    OK(rt_set_sample_slices(8, 16));
    MUST_FAIL(rt_set_sample_slices(-1, 4));
    MUST_FAIL(rt_set_tail_limit(-5));
    for (int mode : {0, 2, 1}) {   // tiles in image order, in order of cost for every call, the default rule
        OK(rt_set_tile_order(mode));
        OK(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, 0, a1, nullptr));
        if (!same("rt_render, tile order")) return 1;
        OK(rt_render_bands(W, H, 1, 2, 0, SPP, DEPTH, 5, 0, a1, nullptr));   // (rows of the other bands keep what they hold)
        if (!same("rt_render_bands, tile order")) return 1;
    }
    MUST_FAIL(rt_set_tile_order(3));
    OK(rt_set_shading_sort(1));
    OK(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    if (!same("rt_render, shading sort")) return 1;
    OK(rt_set_shading_sort(0));

    OK(rt_render_noalloc(W, H, 0, H, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    if (!same("rt_render_noalloc")) return 1;
    MUST_FAIL(rt_render_noalloc(W, H, 0, H, 0, SPP, RT_NOALLOC_MAX_DEPTH + 1, 5, 0, a1, nullptr));
    // passes that continue in sample order, the second without waiting
    OK(rt_render(W, H, 0, H, 0, 10, DEPTH, 5, 0, a1, nullptr));
    OK(rt_render(W, H, 0, H, 10, SPP - 10, DEPTH, 5, RT_RENDER_CONTINUE | RT_RENDER_NOWAIT, a1, nullptr));
    OK(rt_sync());
    if (!same("two passes with RT_RENDER_CONTINUE")) return 1;
    MUST_FAIL(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, RT_RENDER_CONTINUE | RT_RENDER_ACCUMULATE, a1, nullptr));
    // bands of three shares and row ranges
    OK(rt_memset(a1, 0xff, npx * 16));
    for (int r = 0; r < 3; ++r) OK(rt_render_bands(W, H, r, 3, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    if (!same("rt_render_bands")) return 1;
    OK(rt_memset(a1, 0xff, npx * 16));
    OK(rt_render(W, H, 0, 30, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    OK(rt_render(W, H, 30, H - 30, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    if (!same("rt_render in two row ranges")) return 1;
    MUST_FAIL(rt_render(W, H, 10, H, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    MUST_FAIL(rt_render(0, H, 0, H, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    MUST_FAIL(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, 0, nullptr, nullptr));
    // pixel lists: short (rendered as listed), long (sorted by tile on the device, a workspace that grows), with strays
    for (int n : {1, 100, 5000, 40000}) {
        std::vector<int32_t> px(static_cast<size_t>(n), 0);
        for (auto& p : px) p = int32_t(lcg(x) % uint32_t(npx));
        if (n >= 100) { px[3] = -1; px[7] = int32_t(npx); }
        int32_t* d_px = nullptr;
        OK(rt_malloc(reinterpret_cast<void**>(&d_px), size_t(n) * 4));
        OK(rt_memcpy_h2d(d_px, px.data(), size_t(n) * 4));
        OK(rt_memset(a1, 0, npx * 16));
        OK(rt_render_pixels(W, H, n, d_px, 0, SPP, DEPTH, 5, 0, a1, nullptr));
        OK(rt_memcpy_d2h(h1.data(), a1, npx * 16));
        OK(rt_memcpy_d2h(h0.data(), a0, npx * 16));
        std::vector<char> listed(npx, 0);
        for (int32_t p : px) if (p >= 0 && size_t(p) < npx) listed[size_t(p)] = 1;
        for (size_t p = 0; p < npx; ++p) {
            const bool eq = std::memcmp(&h1[4 * p], &h0[4 * p], 16) == 0, zero = h1[4 * p + 3] == 0.0f;
            if (listed[p] ? !eq : !zero) { std::fprintf(stderr, "rt_render_pixels(%d): pixel %zu\n", n, p); return 1; }
        }
        OK(rt_free(d_px));
    }
    MUST_FAIL(rt_render_pixels(W, H, 5, nullptr, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    OK(rt_render(W, H, 0, H, 0, 2, DEPTH, 5, RT_RENDER_VERIFY, a1, nullptr));
    uint64_t bad = 1;
    OK(rt_verify_mismatches(&bad));
    if (bad != 0) { std::fprintf(stderr, "%llu scans differ from the list-order loop\n", (unsigned long long)bad); return 1; }
    OK(rt_render(W, H, 0, H, 0, 2, DEPTH, 5, RT_RENDER_NORMALS, a1, nullptr));

    // primary hits, caller rays (device buffers and page-locked host memory), tonemap
    int32_t* d_prim = nullptr; double* d_t = nullptr; uint32_t* d_img = nullptr;
    OK(rt_malloc(reinterpret_cast<void**>(&d_prim), npx * 4));
    OK(rt_malloc(reinterpret_cast<void**>(&d_t), npx * 8));
    OK(rt_malloc(reinterpret_cast<void**>(&d_img), npx * 4));
    OK(rt_primary_hits(W, H, 5, d_prim, d_t));
    OK(rt_primary_hits(W, H, 5, nullptr, d_t));
    OK(rt_primary_hits(W, H, 5, d_prim, nullptr));
    OK(rt_tonemap(W, H, a0, 0, d_img));
    OK(rt_tonemap(W, H, a0, SPP, d_img));
    MUST_FAIL(rt_tonemap(W, H, nullptr, SPP, d_img));
    for (int n : {1, 63, 4097}) {
        char* pin = nullptr;
        const size_t per = 12 + 12 + 4 + 8 + 12 + 12 + 1 + 12, bytes = size_t(n) * per + 64;
        OK(rt_malloc_host(reinterpret_cast<void**>(&pin), bytes));
        float* o = reinterpret_cast<float*>(pin);
        float* d = o + 3 * size_t(n);
        double* t = reinterpret_cast<double*>(pin + (size_t(n) * 24 + 7) / 8 * 8);
        int32_t* pr = reinterpret_cast<int32_t*>(t + n);
        float* p = reinterpret_cast<float*>(pr + n);
        float* nn = p + 3 * size_t(n);
        float* rgb = nn + 3 * size_t(n);
        uint8_t* ff = reinterpret_cast<uint8_t*>(rgb + 3 * size_t(n));
        for (int i = 0; i < n; ++i) {
            o[3 * i] = 13.0f; o[3 * i + 1] = 2.0f; o[3 * i + 2] = 3.0f;
            d[3 * i] = -13.0f + unit(x) * 4 - 2; d[3 * i + 1] = -2.0f + unit(x) * 2 - 1; d[3 * i + 2] = -3.0f + unit(x) * 4 - 2;
        }
        OK(rt_hit_rays(n, o, d, 0.001, INFINITY, pr, t, p, nn, ff));
        OK(rt_hit_rays(n, o, d, -3.0, 25.0, pr, t, nullptr, nullptr, nullptr));
        OK(rt_hit_rays(n, o, d, 0.0011, 40.0, pr, t, p, nullptr, ff));   // a t_min that is no float: the list-order scan
        OK(rt_trace_rays(n, o, d, DEPTH, 9, 0, rgb));
        OK(rt_trace_rays(n, o, d, DEPTH, 9, RT_RENDER_NORMALS, rgb));
        int hits = 0;
        for (int i = 0; i < n; ++i) hits += pr[i] >= 0;
        std::printf("%d caller rays through page-locked memory: %d hit\n", n, hits);
        OK(rt_free_host(pin));
    }
    MUST_FAIL(rt_hit_rays(4, nullptr, nullptr, 0.0, 1.0, nullptr, nullptr, nullptr, nullptr, nullptr));
    (void)rt_hit_rays(0, nullptr, nullptr, 0.0, 1.0, nullptr, nullptr, nullptr, nullptr, nullptr);   // (n = 0: either answer, no access)

    // a caller's stream slot and back, a second upload (buffers are replaced), deep paths (the HBM part of the stack grows)
    OK(rt_set_stream(nullptr));
    OK(rt_render(W, H, 0, H, 0, 4, 40, 5, 0, a1, nullptr));
    OK(rt_reset_stream());
    sc.data[4 * RT_PRIM_STRIDE + 1] = 0.7;   // a sphere moves (the reference's slider)
    OK(rt_upload_scene(scene));
    OK(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    // two constant_medium volumes join the list (such a scene is scanned in list order; RT_RENDER_VERIFY has nothing to compare)
    sc.prim(RT_PRIM_MEDIUM_SPHERE, 4, {2.0, 1.0, 2.0, 1.0, -1.0 / 0.8});
    sc.prim(RT_PRIM_MEDIUM_BOX, 4, {-3.0, 0.0, -3.0, -2.0, 1.0, -2.0, -1.0 / 0.5});
    scene = sc.finish(RT_PRIM_XZ_RECT, {-1.0, 1.0, -1.0, 1.0, 5.0});
    OK(rt_upload_scene(scene));
    OK(rt_render(W, H, 0, H, 0, 4, DEPTH, 5, 0, a1, nullptr));
    OK(rt_render_noalloc(W, H, 0, H, 0, 4, DEPTH, 5, 0, a0, nullptr));
    if (!same("rt_render_noalloc with media")) return 1;
    MUST_FAIL(rt_render(W, H, 0, H, 0, 2, DEPTH, 5, RT_RENDER_VERIFY, a1, nullptr));
    OK(rt_primary_hits(W, H, 5, d_prim, d_t));
    // spheres only, then an empty list: other kernel instances, empty groups
    SceneArrays only;
    only.material(RT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f, 0.0);
    only.prim(RT_PRIM_SPHERE, 0, {0.0, 0.0, -1.0, 0.5});
    OK(rt_upload_scene(only.finish(RT_LIGHT_BASE, {})));
    OK(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, 0, a1, nullptr));
    SceneArrays none;
    none.material(RT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f, 0.0);
    if (rt_upload_scene(none.finish(RT_LIGHT_BASE, {})) == 0) OK(rt_render(W, H, 0, H, 0, 2, DEPTH, 5, 0, a1, nullptr));

    OK(rt_free(d_prim)); OK(rt_free(d_t)); OK(rt_free(d_img)); OK(rt_free(a0)); OK(rt_free(a1)); OK(rt_free(d_rays));
    MUST_FAIL(rt_set_device(count));
    OK(rt_set_device(0));
    rt_shutdown();
    MUST_FAIL(rt_render(W, H, 0, H, 0, SPP, DEPTH, 5, 0, nullptr, nullptr));   // no context
    OK(rt_init(0));                                                            // and a second life
    OK(rt_upload_scene(scene));
    OK(rt_set_camera(&cam));
    OK(rt_malloc(reinterpret_cast<void**>(&a0), npx * 16));
    OK(rt_render(W, H, 0, H, 0, 2, DEPTH, 5, 0, a0, nullptr));
    OK(rt_free(a0));
    rt_shutdown();
    std::printf("abi tour ok\n");
    return 0;
}
