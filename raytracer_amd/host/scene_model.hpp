// Host-side scene model with the reference renderer's names and constructor signatures,
// so that scene-construction code written against the reference (its SetupScene,
// src/main.cc:144-207) compiles unchanged:
//   Color, DiffuseMaterial            src/color.h:7-74
//   Light, PointLight                 src/common.h:47-64
//   Object, Plane, Sphere             src/objects.h:11-147
//   ViewFrustum                       src/view_frustum.h:12-59
//   Screen                            src/screen.h:12-125
// What differs: the per-ray work (TestCollision, calcNormal, Shade, GetRayByPixel) is not
// here at all - it runs on the GPU behind the C ABI in include/rt_hip.h - and every object
// can flatten itself into the rt_prim record that ABI takes (the reference keeps its
// geometry private with no accessors, src/objects.h:107-110,143-146).  Screen presents
// into a BMP/PPM file instead of a Windows device context.
#pragma once

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "vec3d.hpp"

struct Color {
    Color(double r, double g, double b) : r_(r), g_(g), b_(b) {}

    // Negative light does not exist: setters clamp at zero (a NaN becomes zero too, as
    // with the `(a > b) ? a : b` macro the reference's setters expand to).
    void r(double v) { r_ = (v > 0.0) ? v : 0.0; }
    void g(double v) { g_ = (v > 0.0) ? v : 0.0; }
    void b(double v) { b_ = (v > 0.0) ? v : 0.0; }
    double r() const { return r_; }
    double g() const { return g_; }
    double b() const { return b_; }

    // 8-bit channel: clamp to [0,1], scale by 255, truncate.
    uint8_t red() const { return quantize(r_); }
    uint8_t green() const { return quantize(g_); }
    uint8_t blue() const { return quantize(b_); }

    Color operator*(double k) const { return Color(r_ * k, g_ * k, b_ * k); }
    Color operator*(const Color& o) const { return Color(r_ * o.r_, g_ * o.g_, b_ * o.b_); }
    Color operator+(const Color& o) const { return Color(r_ + o.r_, g_ + o.g_, b_ + o.b_); }
    Color& operator+=(const Color& o) {
        r_ += o.r_; g_ += o.g_; b_ += o.b_;
        return *this;
    }

private:
    static uint8_t quantize(double v) {
        const double s = std::clamp(v, 0.0, 1.0) * 255;
        return (s == s) ? static_cast<uint8_t>(s) : uint8_t{0};
    }
    double r_, g_, b_;
};

struct DiffuseMaterial {
    Color kd;  // diffuse coefficient
    Color ka;  // ambient coefficient (carried, never used by the shading)
    Color ks;  // specular coefficient (carried, never used by the shading)
    double reflectivity = 1.0;
    double transparency = 0.0;
    double refraction_index = 1.0;
};

struct Light {
    Light(Color c, float intensity) : light_color(c), intensity(intensity) {}
    Color light_color;
    float intensity;
};

struct PointLight : public Light {
    PointLight(Color c, float intensity, gmtl::Vec3d pos) : Light(c, intensity), pos(pos) {}
    gmtl::Vec3d pos;

    rt_light Flatten() const {
        rt_light l{};
        for (unsigned i = 0; i < 3; ++i) l.pos[i] = pos[i];
        l.color[0] = light_color.r(); l.color[1] = light_color.g(); l.color[2] = light_color.b();
        l.intensity = static_cast<double>(intensity);
        return l;
    }
};

class Object {
public:
    Object(DiffuseMaterial material, Light ambient_light) : material_(material), ambient_light_(ambient_light) {}
    virtual ~Object() = default;

    DiffuseMaterial material() const { return material_; }
    const Light& GetAmbient() const { return ambient_light_; }

    // The record the device library takes for this collider.
    virtual rt_prim Flatten() const = 0;

    // The same object after a rigid motion: positions go through move(v, pivot), directions
    // through move(v, 0) (and are normalised again by the constructor).
    using Motion = std::function<gmtl::Vec3d(const gmtl::Vec3d&, const gmtl::Vec3d&)>;
    virtual std::unique_ptr<Object> Moved(const Motion& move, const gmtl::Vec3d& pivot) const = 0;

protected:
    rt_prim FlattenMaterial(int32_t type) const {
        rt_prim p{};
        p.type = type;
        p.kd[0] = material_.kd.r(); p.kd[1] = material_.kd.g(); p.kd[2] = material_.kd.b();
        p.reflectivity = material_.reflectivity;
        p.transparency = material_.transparency;
        p.refraction_index = material_.refraction_index;
        p.ambient_intensity = static_cast<double>(ambient_light_.intensity);
        return p;
    }
    DiffuseMaterial material_;
    Light ambient_light_;
};

class Plane : public Object {
public:
    Plane(gmtl::Vec3d origin, gmtl::Vec3d normal, DiffuseMaterial material, Light ka)
        : Object(material, ka), origin_(origin), normal_(gmtl::makeNormal(normal)) {}

    rt_prim Flatten() const override {
        rt_prim p = FlattenMaterial(RT_PRIM_PLANE);
        for (unsigned i = 0; i < 3; ++i) { p.p[i] = origin_[i]; p.q[i] = normal_[i]; }
        return p;
    }
    std::unique_ptr<Object> Moved(const Motion& move, const gmtl::Vec3d& pivot) const override {
        return std::make_unique<Plane>(move(origin_, pivot), move(normal_, gmtl::Vec3d{0, 0, 0}), material_,
                                       ambient_light_);
    }

private:
    gmtl::Vec3d origin_;
    gmtl::Vec3d normal_;  // unit length from construction on
};

class Sphere : public Object {
public:
    // The radius arrives as float, as in the reference, and is widened to double.
    Sphere(gmtl::Vec3d center, float radius, DiffuseMaterial material, Light ka)
        : Object(material, ka), center_(center), radius_(radius) {}

    rt_prim Flatten() const override {
        rt_prim p = FlattenMaterial(RT_PRIM_SPHERE);
        for (unsigned i = 0; i < 3; ++i) p.p[i] = center_[i];
        p.q[0] = radius_;
        return p;
    }
    std::unique_ptr<Object> Moved(const Motion& move, const gmtl::Vec3d& pivot) const override {
        return std::make_unique<Sphere>(move(center_, pivot), static_cast<float>(radius_), material_, ambient_light_);
    }

private:
    gmtl::Vec3d center_;
    double radius_;
};

// Camera description.  The pixel -> ray arithmetic itself runs on the device from
// (width, height, angle); the physical-pixel figures are kept for inspection.
class ViewFrustum {
public:
    ViewFrustum(std::size_t scr_width, std::size_t scr_height, double angle, double view_plane_distance);
    std::size_t Width() const { return scr_width_; }
    std::size_t Height() const { return scr_height_; }
    double Angle() const { return angle_; }
    double PixelWidth() const { return pixel_width_; }
    double PixelHeight() const { return pixel_height_; }

private:
    std::size_t scr_width_, scr_height_;
    double angle_, view_plane_distance_;
    double pixel_width_, pixel_height_;
};

// Double-buffered 24-bit framebuffer, rows bottom-up, bytes B,G,R (the layout of the
// reference's Screen, src/screen.h:12-125).  DrawScreen() swaps and writes the front buffer
// to the output path given at construction (".ppm" -> binary PPM, anything else -> BMP).
// As in the reference the depth buffer is never reset: from the second frame on a store at
// the same depth merges with a per-channel max into the (zeroed) back buffer, which gives the
// same pixels as an overwrite.
// The two pixel buffers are page-locked when the device library can provide such memory
// (rt_host_alloc), so that a frame arrives in them at the full PCIe rate, and ordinary
// memory otherwise.
class Screen {
public:
    Screen(std::size_t width, std::size_t height, std::string output_path = "");
    ~Screen();
    Screen(const Screen&) = delete;
    Screen& operator=(const Screen&) = delete;

    void ClearScreenColor(const Color& c = {255, 255, 255});
    void StorePixel(std::size_t x, std::size_t y, uint8_t z, const Color& c = {255, 255, 255});
    // StorePixel(x, y, z, colour) for whole rows of already-quantised pixels (what the
    // device returns), same visibility rule, row-wise where a whole row passes the `<` test.
    void StoreRows(std::size_t first_row, std::size_t n_rows, uint8_t z, const uint8_t* bgr);
    // When storing a whole frame at depth z would replace every pixel (every depth below z,
    // or equal to z over a back buffer that is still all zero), mark the frame as stored and
    // return the back buffer for the caller to fill directly; nullptr otherwise.
    uint8_t* BeginFrameStore(uint8_t z);
    uint8_t* GetScreenData() { return buffers_[front_]; }
    void DrawScreen();
    std::size_t Width() const { return width_; }
    std::size_t Height() const { return height_; }
    bool PageLocked() const { return page_locked_; }

private:
    void MaterialiseDepth();
    std::size_t width_, height_;
    std::size_t front_ = 1;
    std::array<uint8_t*, 2> buffers_{{nullptr, nullptr}};
    bool page_locked_ = false;
    bool back_is_zero_ = true;          // nothing stored into the back buffer since it was cleared
    uint8_t depth_floor_ = 0;           // depth of every pixel while depth_ is empty
    std::vector<uint8_t> depth_;        // per-pixel depths once they stop being uniform
    std::string output_path_;
};

// Where a Tracer() call spent its time (seconds); the first call on a device also creates the
// device context, a call with a new scene uploads it.
struct TracerTimings {
    double create_s = 0.0;   // rt_create (0 when the context existed)
    double upload_s = 0.0;   // rt_upload_scene (0 when the scene was resident)
    double render_s = 0.0;   // rt_render: trace + copy into the Screen
};

struct TracerOptions {
    int max_depth = 4;    // the reference's MAX_DEPTH (src/main.cc:17)
    double fov_deg = 60;  // src/main.cc:126
    int device = 0;
    TracerTimings* timings = nullptr;  // filled when given
};

// Drop-in for the reference's Tracer() (src/main.cc:115-120): same arguments, same
// effect on `scr` (every pixel stored at depth 1).  Throws std::runtime_error when the
// device library reports an error.  The device context lives on between calls (one per
// device and process) and the scene is only uploaded again when it has changed, so a
// sequence of frames pays for set-up once; ReleaseTracerDevices() lets go of it.
void Tracer(const Color& bgr, Screen& scr, const std::vector<std::unique_ptr<Object>>& objs,
            const std::vector<std::unique_ptr<PointLight>>& lights);
void Tracer(const Color& bgr, Screen& scr, const std::vector<std::unique_ptr<Object>>& objs,
            const std::vector<std::unique_ptr<PointLight>>& lights, const TracerOptions& opt);
void ReleaseTracerDevices();

// Writers used by Screen::DrawScreen.
void WriteBmp(const std::string& path, const uint8_t* bgr, std::size_t width, std::size_t height);
void WritePpm(const std::string& path, const uint8_t* bgr, std::size_t width, std::size_t height);
