#include "scene_builders.hpp"

#include <cmath>

uint64_t SplitMix64::NextU64() {
    s_ += 0x9E3779B97F4A7C15ull;
    uint64_t z = s_;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

double SplitMix64::Uniform(double lo, double hi) {
    const double u = static_cast<double>(NextU64() >> 11) * (1.0 / 9007199254740992.0);
    return lo + (hi - lo) * u;
}

// Same objects, lights and literals as the reference's SetupScene (src/main.cc:153-206).
void SetupScene(ObjectList& objs, LightList& lights, Color& bgr) {
    const Light ambient{bgr, 0.3f};
    const Color grey = Color{1.0, 1.0, 1.0} * 0.5;
    const Color violet{0.5, 0.313, 0.64};
    const Color orange{0.9, 0.4, 0.298};

    objs.push_back(std::make_unique<Plane>(gmtl::Vec3d{0, -3, 0}, gmtl::Vec3d{0, 1, 0},
                                           DiffuseMaterial{grey, grey, Color{0, 0, 0} * 0.5, 0, 0}, ambient));
    objs.push_back(std::make_unique<Sphere>(gmtl::Vec3d{2, 0.5, 10}, 1,
                                            DiffuseMaterial{violet, violet, violet, .7, .9f, 1.1}, ambient));
    objs.push_back(std::make_unique<Sphere>(gmtl::Vec3d{1.5, 0, 13}, 1,
                                            DiffuseMaterial{orange, orange, orange, 0.2f, 0.f}, ambient));

    lights.push_back(std::make_unique<PointLight>(Color{1, 1, 1}, 1.f, gmtl::Vec3d{0, 4, 7}));
    lights.push_back(std::make_unique<PointLight>(Color{1, 1, 1}, .3f, gmtl::Vec3d{10, 8, 3}));
}

namespace {

const double kPi = 3.141592653589793;

void RingLights(SplitMix64& rng, LightList& lights, int count, double radius, double cz, double lo, double hi) {
    for (int i = 0; i < count; ++i) {
        const double ang = 2.0 * kPi * i / count;
        const double y = rng.Uniform(6.0, 12.0);
        const double inten = rng.Uniform(lo, hi);
        lights.push_back(std::make_unique<PointLight>(
            Color{1, 1, 1}, static_cast<float>(inten),
            gmtl::Vec3d{radius * std::cos(ang), y, cz + radius * std::sin(ang)}));
    }
}

void Floor(ObjectList& objs, const Light& ambient) {
    const Color half{0.5, 0.5, 0.5};
    objs.push_back(std::make_unique<Plane>(gmtl::Vec3d{0, -3, 0}, gmtl::Vec3d{0, 1, 0},
                                           DiffuseMaterial{half, half, half, 0.0, 0.0}, ambient));
}

}  // namespace

void SetupRandomSpheresScene(ObjectList& objs, LightList& lights, Color& bgr, int n_spheres, int n_lights, uint64_t seed) {
    SplitMix64 rng(seed);
    const Light ambient{bgr, 0.3f};
    Floor(objs, ambient);
    for (int i = 0; i < n_spheres; ++i) {
        const double cx = rng.Uniform(-20, 20), cy = rng.Uniform(-2.5, 6), cz = rng.Uniform(6, 60);
        const double rad = rng.Uniform(0.15, 0.6);
        const double kr = rng.Uniform(0.1, 1), kg = rng.Uniform(0.1, 1), kb = rng.Uniform(0.1, 1);
        const double refl = rng.Uniform(0, 1);
        const bool opaque = rng.Uniform() < 0.7;
        const double tr_draw = rng.Uniform(0.3, 0.95);
        const double tr = opaque ? 0.0 : tr_draw;
        const double ri = rng.Uniform(1.05, 1.5);
        const Color kd{kr, kg, kb};
        objs.push_back(std::make_unique<Sphere>(gmtl::Vec3d{cx, cy, cz}, static_cast<float>(rad),
                                                DiffuseMaterial{kd, kd, kd, refl, tr, ri}, ambient));
    }
    RingLights(rng, lights, n_lights, 18.0, 30.0, 0.1, 0.5);
}

void SetupGlassScene(ObjectList& objs, LightList& lights, Color& bgr, int n_spheres, int n_lights, uint64_t seed) {
    SplitMix64 rng(seed);
    const Light ambient{bgr, 0.3f};
    Floor(objs, ambient);
    for (int i = 0; i < n_spheres; ++i) {
        const double cx = rng.Uniform(-6, 6), cy = rng.Uniform(-2, 3.5), cz = rng.Uniform(8, 22);
        const double rad = rng.Uniform(0.6, 1.4);
        const double kr = rng.Uniform(0.2, 1), kg = rng.Uniform(0.2, 1), kb = rng.Uniform(0.2, 1);
        const double refl = rng.Uniform(0.2, 0.9);
        const double tr = rng.Uniform(0.8, 0.95);
        const double ri = rng.Uniform(1.3, 1.6);
        const Color kd{kr, kg, kb};
        objs.push_back(std::make_unique<Sphere>(gmtl::Vec3d{cx, cy, cz}, static_cast<float>(rad),
                                                DiffuseMaterial{kd, kd, kd, refl, tr, ri}, ambient));
    }
    RingLights(rng, lights, n_lights, 10.0, 14.0, 0.03, 0.12);
}

void OrbitScene(ObjectList& objs, LightList& lights, double angle_rad, gmtl::Vec3d pivot) {
    const double c = std::cos(-angle_rad), s = std::sin(-angle_rad);
    const auto rot = [c, s](const gmtl::Vec3d& v, const gmtl::Vec3d& about) {
        const double x = v[0] - about[0], y = v[1] - about[1], z = v[2] - about[2];
        return gmtl::Vec3d{c * x + s * z + about[0], y + about[1], -s * x + c * z + about[2]};
    };
    for (auto& o : objs) o = o->Moved(rot, pivot);
    for (auto& l : lights) l = std::make_unique<PointLight>(l->light_color, l->intensity, rot(l->pos, pivot));
}
