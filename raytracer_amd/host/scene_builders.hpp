// Scene construction for the host program: the reference's SetupScene (src/main.cc:144-207)
// and the synthetic scenes of the larger benchmark configurations, all written with the
// reference's own construction calls (Plane, Sphere, PointLight, DiffuseMaterial).
// The synthetic scenes draw from a fixed splitmix64 stream, so every run builds the same
// objects, byte for byte the ones raytracer_amd/scenes.py builds.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "scene_model.hpp"

using ObjectList = std::vector<std::unique_ptr<Object>>;
using LightList = std::vector<std::unique_ptr<PointLight>>;

// splitmix64; uniform() = lo + (hi - lo) * (53 random bits / 2^53).
class SplitMix64 {
public:
    explicit SplitMix64(uint64_t seed) : s_(seed) {}
    uint64_t NextU64();
    double Uniform(double lo = 0.0, double hi = 1.0);

private:
    uint64_t s_;
};

// The reference's scene: floor, glassy violet sphere, opaque orange sphere, two lights.
void SetupScene(ObjectList& objs, LightList& lights, Color& bgr);

// Floor (as in the default scene) at index 0 + n_spheres random spheres; lights on a ring.
void SetupRandomSpheresScene(ObjectList& objs, LightList& lights, Color& bgr, int n_spheres = 1024, int n_lights = 8,
                             uint64_t seed = 0x5EED1024);

// Refraction-heavy scene: floor + transparent spheres in front of the camera.
void SetupGlassScene(ObjectList& objs, LightList& lights, Color& bgr, int n_spheres = 24, int n_lights = 16,
                     uint64_t seed = 0x61A55);

// Camera orbit about a vertical axis through `pivot`, applied as the inverse rigid rotation
// of the scene (the reference's camera is fixed at the origin looking down +z).
void OrbitScene(ObjectList& objs, LightList& lights, double angle_rad, gmtl::Vec3d pivot = gmtl::Vec3d{1.75, 0.0, 11.5});
