// The reference's only live caller of the path, RayView::renderOneRay (rayview.cpp:67-87), written against the
// mirror headers exactly as the reference writes it against its own: cam.get_ray(u, v), the six-argument
// calc::ray_color, calc::rgbPerSamples -- no session, no device handle in sight. Proof that the headers are a
// source-level drop-in for that caller. Besides doing what renderOneRay does, it logs every tick
// (pixel, (u, v) and ray of every sample, colour sum, tonemapped pixel) so that tests/test_gpu_parity.py can check
// the host arithmetic against the oracle's restatement of the reference and the colours against the oracle's ray_color.
//   rayview_tick N_TICKS SAMPLES DEPTH SEED SEED01 SEED11 OUT.bin
#include <cstdio>
#include <cstdlib>

#include <hitPosition.h>

#include <calc.h>
#include <camera.h>
#include <shapes.h>

// rayview.cpp:233-260 (m_shpereY = 0.5f, rayview.cpp:31)
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

struct RayViewCore {   // the members of RayView that renderOneRay touches (rayview.h:62-93)
    std::vector<std::shared_ptr<shape>> m_worldObjects = normal_scene(0.5f);
    std::shared_ptr<shape> worldLights = std::make_shared<sphere>(QVector3D(190, 90, 190), 90, std::shared_ptr<material>());   // rayview.cpp:49
    QVector3D m_lookAt{0.0f, 0.0f, -1.0f}, m_lookFrom{0.0f, 0.0f, 1.0f}, m_vup{0.0f, 1.0f, 0.0f};
    camera cam{m_lookFrom, m_lookAt, m_vup, (m_lookFrom - m_lookAt).length()};                                                  // rayview.cpp:27
    QVector3D m_default_pixel_color;
    int m_numSamples = 1, m_depth = 10;
    bool m_isColorOnly = false;
    FILE* log = nullptr;

    void put(const void* p, size_t n) { fwrite(p, 1, n, log); }
    void put3(const QVector3D& v) { const float f[3] = {v.x(), v.y(), v.z()}; put(f, 12); }

    void renderOneRay() {                                           // rayview.cpp:67-87
        int randX = calc::random_doubleW();
        int randY = calc::random_doubleH();
        m_default_pixel_color.setX(0);
        m_default_pixel_color.setY(0);
        m_default_pixel_color.setZ(0);
        int numSamples = m_numSamples;
        const int32_t head[2] = {randX, randY};
        put(head, 8);
        for (int s = 0; s < numSamples; ++s) {
            float u = (randX + calc::random_double01()) / img::width;
            float v = (randY + calc::random_double01()) / img::height;
            const Ray ray = cam.get_ray(u, v);
            const uint64_t stream = rt_host::ray_color_stream();    // which counter-RNG stream this call will draw from
            const QVector3D c = calc::ray_color(ray, img::gradientBgVec, m_worldObjects, worldLights, m_depth, m_isColorOnly);
            m_default_pixel_color += c;
            const float uv[2] = {u, v};
            put(uv, 8); put3(ray.pointOfOrigin); put3(ray.direction); put(&ray.DirlengthSquared, 4); put(&stream, 8); put3(c);
        }
        put3(m_default_pixel_color);
        put3(calc::rgbPerSamples(m_default_pixel_color, numSamples));   // what writeToImg stores at (W - randX - 1, H - randY - 1)
    }
};

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s N_TICKS SAMPLES DEPTH SEED SEED01 SEED11 OUT.bin\n", argv[0]); return 2; }
    try {
        RayViewCore view;
        const int ticks = atoi(argv[1]);
        view.m_numSamples = atoi(argv[2]);
        view.m_depth = atoi(argv[3]);
        calc::gen.seed(unsigned(atol(argv[4]))); calc::gen01.seed(unsigned(atol(argv[5]))); calc::gen11.seed(unsigned(atol(argv[6])));
        view.log = fopen(argv[7], "wb");
        if (!view.log) { perror(argv[7]); return 2; }
        for (int i = 0; i < ticks; ++i) view.renderOneRay();
        // hitFromList with the reference's four arguments, and shape::hit / bounding_box on one shape
        Ray centre(view.m_lookFrom, QVector3D(0, 0, -2));
        auto rec = hitFromList(view.m_worldObjects, centre, calc::smallestVal, calc::infinity);
        auto one = view.m_worldObjects.at(0)->hit(centre, calc::smallestVal, calc::infinity);
        aabb bb;
        const bool has_box = boundingBoxFromList(view.m_worldObjects, bb);
        // the slider (rayview.cpp:130-134) mutates the first shape; the next calls must see the moved sphere
        view.m_worldObjects.at(0)->center.setY(0.8f);
        for (int i = 0; i < ticks; ++i) view.renderOneRay();
        fclose(view.log);
        printf("ticks %d x2, uploads %d, centre ray: list prim %d t=%.9g, first sphere alone t=%.9g, world box %d (%g %g %g)-(%g %g %g) hit %d\n",
               ticks, rt_host::default_session().uploads, rec ? rec->rt_prim : -1, rec ? rec->t : 0.0, one ? one->t : 0.0, int(has_box),
               bb.min().x(), bb.min().y(), bb.min().z(), bb.max().x(), bb.max().y(), bb.max().z(), int(bb.hit(centre, 0.001, 1e30)));
    } catch (const std::exception& e) {
        fprintf(stderr, "rayview_tick: %s\n", e.what());
        return 1;
    }
    return 0;
}
