// How fast are the reference-signature entry points? RayView::renderOneRay's inner loop (rayview.cpp:79-84) on the built-in scene
// (rayview.cpp:233-260): cam.get_ray + the six-argument calc::ray_color per sample, timed on the host clock; then the batch overload
// (not in the reference) with the same rays in one launch. The CPU reference traces 0.51 Mrays/s on one core (bench.py cpu_baseline).
//   tick_bench SAMPLES [BATCH ...]        -DTICK_BENCH_NO_BATCH: single-ray part only (for the round-2 headers, which have no batch overload)
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <hitPosition.h>

#include <calc.h>
#include <camera.h>
#include <shapes.h>

int main(int argc, char** argv) {
    const int samples = argc > 1 ? atoi(argv[1]) : 2000;
    try {
        std::vector<std::shared_ptr<shape>> world;
        world.push_back(std::make_shared<sphere>(QVector3D{0.0f, 0.5f, -1.0f}, 0.5f, std::make_shared<lambertian>(QVector3D(0.7, 0.3, 0.3))));
        world.push_back(std::make_shared<sphere>(QVector3D{0.0f, -100.2f, -1.0f}, 100.0f, std::make_shared<lambertian>(QVector3D(0.8, 0.8, 0.0))));
        world.push_back(std::make_shared<sphere>(QVector3D{1.0f, 0.0f, -1.0f}, 0.6f, std::make_shared<metal>(QVector3D(0.6, 0.6, 0.6), 0.0)));
        world.push_back(std::make_shared<sphere>(QVector3D{-1.0f, 0.0f, -1.0f}, 0.6f, std::make_shared<dielectric>(1.5)));
        world.push_back(std::make_shared<rectangle>(3, 5, 1, 3, -2, std::make_shared<diffuse_light>(QVector3D(2, 2, 2))));
        std::shared_ptr<shape> lights = std::make_shared<sphere>(QVector3D(190, 90, 190), 90, std::shared_ptr<material>());
        const QVector3D from{0.0f, 0.0f, 1.0f}, at{0.0f, 0.0f, -1.0f}, up{0.0f, 1.0f, 0.0f};
        camera cam{from, at, up, (from - at).length()};
        calc::gen.seed(1); calc::gen01.seed(2); calc::gen11.seed(3);
        auto pixel_ray = [&]() {
            const int x = calc::random_doubleW(), y = calc::random_doubleH();
            return cam.get_ray((x + calc::random_double01()) / img::width, (y + calc::random_double01()) / img::height);
        };
        QVector3D sum;
        for (int i = 0; i < 20; ++i) sum += calc::ray_color(pixel_ray(), img::gradientBgVec, world, lights, 10);   // warm-up: context, upload, first launches
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < samples; ++i) sum += calc::ray_color(pixel_ray(), img::gradientBgVec, world, lights, 10);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("single-ray calc::ray_color, built-in scene, depth 10: %d calls in %.3f s = %.1f us per call, %.4f Msamples/s (sum %g)\n", samples, dt,
               1e6 * dt / samples, samples / dt / 1e6, sum.x());
#ifndef TICK_BENCH_NO_BATCH
        for (int a = 2; a < argc; ++a) {
            const size_t n = size_t(atol(argv[a]));
            std::vector<Ray> rays(n);
            for (auto& r : rays) r = pixel_ray();
            calc::ray_color(rays, img::gradientBgVec, world, lights, 10);
            t0 = std::chrono::steady_clock::now();
            const int reps = n < 100000 ? 20 : 3;
            for (int k = 0; k < reps; ++k) { auto c = calc::ray_color(rays, img::gradientBgVec, world, lights, 10); sum += c[0]; }
            dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
            printf("batch calc::ray_color, %zu rays per call: %.1f us per call, %.3f Msamples/s\n", n, 1e6 * dt, n / dt / 1e6);
        }
#endif
    } catch (const std::exception& e) {
        fprintf(stderr, "tick_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
