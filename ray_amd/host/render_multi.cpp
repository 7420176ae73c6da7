// One C++ process, one host thread, every GPU of the node: the multi-GPU split of a frame below
// Python, on nothing but the C ABI. One rt_host::Session (= one library context) per device; every share is queued
// with RT_RENDER_NOWAIT so all devices render at the same time; then the shares are collected:
//   spp   share g renders samples [spp*g/N, spp*(g+1)/N) of every pixel; the accumulators are summed in share order
//   tile  share g renders the 8-row bands g, g+N, ... (rt_render_bands); the bands are disjoint and are merged
// With fewer devices than shares (a one-GPU box) the shares are dealt round-robin onto the devices and a device
// renders its shares one after the other -- same arithmetic, which is how tests/test_gpu_parity.py checks it.
// (bench.py does the same across PROCESSES with RCCL; this is the path for a C++ host like RayView.)
//   render_multi WIDTH HEIGHT SPP DEPTH SEED SHARES spp|tile [accum.bin]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include <calc.h>
#include <camera.h>
#include <hitPosition.h>
#include <shapes.h>

static std::vector<std::shared_ptr<shape>> normal_scene(float sphereY) {   // rayview.cpp:233-260
    std::vector<std::shared_ptr<shape>> world;
    auto material_ground = std::make_shared<lambertian>(QVector3D(0.8, 0.8, 0.0));
    auto material_center = std::make_shared<lambertian>(QVector3D(0.7, 0.3, 0.3));
    auto material_left = std::make_shared<dielectric>(1.5);
    auto material_right = std::make_shared<metal>(QVector3D(0.6, 0.6, 0.6), 0.0);
    auto difflight = std::make_shared<diffuse_light>(QVector3D(2, 2, 2));
    world.push_back(std::make_shared<sphere>(QVector3D{0.0f, sphereY, -1.0f}, 0.5f, material_center));
    world.push_back(std::make_shared<sphere>(QVector3D{0.0f, -100.2f, -1.0f}, 100.0f, material_ground));
    world.push_back(std::make_shared<sphere>(QVector3D{1.0f, 0.0f, -1.0f}, 0.6f, material_right));
    world.push_back(std::make_shared<sphere>(QVector3D{-1.0f, 0.0f, -1.0f}, 0.6f, material_left));
    world.push_back(std::make_shared<rectangle>(3, 5, 1, 3, -2, difflight));
    return world;
}

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s WIDTH HEIGHT SPP DEPTH SEED SHARES spp|tile [accum.bin]\n", argv[0]); return 2; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), spp = atoi(argv[3]), depth = atoi(argv[4]);
    const uint64_t seed = strtoull(argv[5], nullptr, 0);
    const int shares = atoi(argv[6]);
    const bool tile = strcmp(argv[7], "tile") == 0;
    try {
        int n_dev = 0;
        rt_host::check(rt_device_count(&n_dev));
        if (n_dev > shares) n_dev = shares;
        auto world = normal_scene(0.5f);
        std::shared_ptr<shape> lights = std::make_shared<sphere>(QVector3D(190, 90, 190), 90, std::shared_ptr<material>());
        camera cam(QVector3D(0.0f, 0.0f, 1.0f), QVector3D(0.0f, 0.0f, -1.0f), QVector3D(0.0f, 1.0f, 0.0f), 2.0);
        const uint64_t n = uint64_t(W) * H;
        std::vector<std::unique_ptr<rt_host::Session>> dev;
        for (int d = 0; d < n_dev; ++d) {                       // one context per device, scene and camera on each
            dev.push_back(std::make_unique<rt_host::Session>(d));
            dev[d]->upload(world, lights, img::gradientBgVec);
            rt_host::check(rt_set_camera(&cam.rt()));
        }
        std::vector<std::unique_ptr<rt_host::DevBuf>> acc(shares), cnt(shares);
        for (int g = 0; g < shares; ++g) {                      // buffers live on the device that renders the share
            dev[g % n_dev]->bind();
            acc[g] = std::make_unique<rt_host::DevBuf>(n * 16); cnt[g] = std::make_unique<rt_host::DevBuf>(8);
            rt_host::check(rt_memset(acc[g]->p, 0, n * 16)); rt_host::check(rt_memset(cnt[g]->p, 0, 8));
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<float> total(n * 4, 0.0f), part(n * 4);
        uint64_t rays = 0;
        for (int g0 = 0; g0 < shares; g0 += n_dev) {            // one round: every device gets one share, queued without waiting
            for (int g = g0; g < shares && g < g0 + n_dev; ++g) {
                dev[g % n_dev]->bind();
                if (tile) rt_host::check(rt_render_bands(W, H, g, shares, 0, spp, depth, seed, RT_RENDER_NOWAIT, (float*)acc[g]->p, (uint64_t*)cnt[g]->p));
                else rt_host::check(rt_render(W, H, 0, H, spp * g / shares, spp * (g + 1) / shares - spp * g / shares, depth, seed, RT_RENDER_NOWAIT,
                                              (float*)acc[g]->p, (uint64_t*)cnt[g]->p));
            }
            for (int g = g0; g < shares && g < g0 + n_dev; ++g) {   // collect in share order (the sum order is part of the result)
                dev[g % n_dev]->bind();
                rt_host::check(rt_sync());
                rt_host::check(rt_memcpy_d2h(part.data(), acc[g]->p, n * 16));
                uint64_t r = 0;
                rt_host::check(rt_memcpy_d2h(&r, cnt[g]->p, 8));
                rays += r;
                for (uint64_t i = 0; i < n * 4; ++i) total[i] += part[i];   // tile: everything outside the share's bands is zero
            }
        }
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("render_multi %dx%d spp=%d depth=%d: %d shares (%s) on %d device(s), %.3f s, %llu rays, %.1f Mrays/s\n", W, H, spp, depth, shares,
               tile ? "tile" : "spp", n_dev, sec, (unsigned long long)rays, rays / sec * 1e-6);
        if (argc > 8) {
            FILE* f = fopen(argv[8], "wb");
            if (!f) { perror(argv[8]); return 2; }
            fwrite(total.data(), sizeof(float), total.size(), f);
            fclose(f);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "render_multi: %s\n", e.what());
        return 1;
    }
    return 0;
}
