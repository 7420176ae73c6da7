// Headless stand-in for main.cpp + RayView (reference main.cpp:6-18, rayview.cpp:13-50): builds the
// reference's normal_scene with the reference's own class names, renders one frame through
// renderAll() on the GPU and writes out.ppm (+ optional raw accumulators for the parity test).
//   render_demo [width height samples depth seed [accum.bin]]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <calc.h>
#include <camera.h>
#include <hitPosition.h>
#include <shapes.h>

// rayview.cpp:233-260, verbatim structure (m_shpereY = 0.5f, rayview.cpp:31)
static std::vector<std::shared_ptr<shape>> normal_scene(float sphereY) {
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
    int W = argc > 1 ? atoi(argv[1]) : 640, H = argc > 2 ? atoi(argv[2]) : 360;
    int spp = argc > 3 ? atoi(argv[3]) : 16, depth = argc > 4 ? atoi(argv[4]) : 8;
    uint64_t seed = argc > 5 ? strtoull(argv[5], nullptr, 0) : 0;
    try {
        rt_host::Session session(0);
        QVector3D lookFrom(0.0f, 0.0f, 1.0f), lookAt(0.0f, 0.0f, -1.0f), vup(0.0f, 1.0f, 0.0f);   // rayview.cpp:18-20
        camera cam(lookFrom, lookAt, vup, 2.0);                                                    // rayview.cpp:27
        auto world = normal_scene(0.5f);
        std::shared_ptr<shape> lights = std::make_shared<sphere>(QVector3D(190, 90, 190), 90, std::shared_ptr<material>());  // rayview.cpp:49
        std::vector<float> accum;
        uint64_t rays = 0;
        auto t0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> img = renderAll(session, W, H, spp, cam, world, depth, lights, img::gradientBgVec, seed, &accum, &rays);
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // single-ray entry points, as calc.cpp:99 calls them
        Ray centre(lookFrom, QVector3D(0, 0, -2));
        auto rec = hitFromList(session, centre, calc::smallestVal, calc::infinity);
        QVector3D col = calc::ray_color(session, centre, depth);
        printf("rendered %dx%d spp=%d depth=%d: %.3f s, %llu rays, %.1f Mrays/s; centre ray hits prim %d t=%.9g colour (%g %g %g)\n",
               W, H, spp, depth, sec, (unsigned long long)rays, rays / sec * 1e-6, rec ? rec->rt_prim : -1, rec ? rec->t : 0.0,
               col.x(), col.y(), col.z());
        // RayView::go: three 16 ms slices of progressive refinement, then a slider move (rayview.cpp:130-134)
        ProgressiveView view(session, W, H, depth, seed);
        view.reset(world, lights, cam);
        int n = 0;
        for (int slice = 0; slice < 3; ++slice) n = view.go(16.0);
        std::vector<float> pacc = view.accumulators();
        world[0]->center.setY(0.8f);
        view.reset(world, lights, cam);
        int n2 = view.go(16.0);
        printf("progressive: %d samples/pixel after 3 slices (accumulator count %.0f), %d after the slider moved\n", n, pacc[3], n2);
        FILE* f = fopen("out.ppm", "wb");
        if (f) {
            fprintf(f, "P6\n%d %d\n255\n", W, H);
            for (uint32_t px : img) { unsigned char c[3] = {(unsigned char)(px >> 16), (unsigned char)(px >> 8), (unsigned char)px}; fwrite(c, 1, 3, f); }
            fclose(f);
        }
        if (argc > 6) {
            FILE* g = fopen(argv[6], "wb");
            if (!g) { perror(argv[6]); return 2; }
            fwrite(accum.data(), sizeof(float), accum.size(), g);
            fclose(g);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "render_demo: %s\n", e.what());
        return 1;
    }
    return 0;
}
