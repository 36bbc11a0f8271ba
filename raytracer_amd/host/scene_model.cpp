// Host side of the drop-in: camera bookkeeping, the headless Screen and the Tracer()
// call that hands the flattened scene to the device library.
#include "scene_model.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>

ViewFrustum::ViewFrustum(std::size_t scr_width, std::size_t scr_height, double angle, double view_plane_distance)
    : scr_width_(scr_width), scr_height_(scr_height), angle_(angle), view_plane_distance_(view_plane_distance) {
    // src/view_frustum.h:18,28-34
    const double to_rad = 3.141592653589793 / 180.;
    const double aspect = static_cast<double>(scr_height) / static_cast<double>(scr_width);
    const double phy_width = 2 * 1 * std::tan(0.5 * angle * to_rad);
    const double phy_height = aspect * phy_width;
    pixel_width_ = phy_width / scr_width;
    pixel_height_ = phy_height / scr_height;
    (void)view_plane_distance_;
}

Screen::Screen(std::size_t width, std::size_t height, std::string output_path)
    : width_(width), height_(height), output_path_(std::move(output_path)) {
    const std::size_t bytes = width * height * 3;
    void* a = nullptr;
    void* b = nullptr;
    if (bytes && rt_host_alloc(bytes, &a) == RT_OK && rt_host_alloc(bytes, &b) == RT_OK) {
        page_locked_ = true;
    } else {
        if (a) rt_host_free(a);
        a = new uint8_t[bytes ? bytes : 1];
        b = new uint8_t[bytes ? bytes : 1];
    }
    buffers_[0] = static_cast<uint8_t*>(a);
    buffers_[1] = static_cast<uint8_t*>(b);
    std::memset(buffers_[0], 0, bytes);
    std::memset(buffers_[1], 0, bytes);
}

Screen::~Screen() {
    for (uint8_t* p : buffers_) {
        if (page_locked_) rt_host_free(p);
        else delete[] p;
    }
}

void Screen::MaterialiseDepth() {
    if (depth_.empty()) depth_.assign(width_ * height_, depth_floor_);
}

// The reference walks (row, column) but passes them to StorePixel as (x, y)
// (src/screen.h:47-53), which only works for square frames; here x is the column.
void Screen::ClearScreenColor(const Color& c) {
    if (depth_.empty()) {
        // every pixel at one depth: a store at depth 0 does nothing over a deeper frame and
        // is the equal-depth merge over a frame at depth 0
        if (depth_floor_ > 0) return;
        uint8_t* back = buffers_[(front_ + 1) % 2];
        const uint8_t v[3] = {c.blue(), c.green(), c.red()};
        const std::size_t n = width_ * height_;
        if (back_is_zero_) {
            for (std::size_t i = 0; i < n; ++i) { back[3 * i] = v[0]; back[3 * i + 1] = v[1]; back[3 * i + 2] = v[2]; }
        } else {
            for (std::size_t i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) back[3 * i + k] = std::max(v[k], back[3 * i + k]);
        }
        back_is_zero_ = false;
        return;
    }
    for (std::size_t y = 0; y < height_; ++y)
        for (std::size_t x = 0; x < width_; ++x) StorePixel(x, y, 0, c);
}

// src/screen.h:55-73: nearer (larger z) wins, equal z keeps the brighter channel.
void Screen::StorePixel(std::size_t x, std::size_t y, uint8_t z, const Color& c) {
    MaterialiseDepth();
    uint8_t* back = buffers_[(front_ + 1) % 2];
    const std::size_t px = y * width_ + x;
    uint8_t* dst = &back[3 * px];
    const uint8_t v[3] = {c.blue(), c.green(), c.red()};
    if (depth_[px] < z) {
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
        depth_[px] = z;
        back_is_zero_ = false;
    } else if (depth_[px] == z) {
        for (int i = 0; i < 3; ++i) dst[i] = std::max(v[i], dst[i]);
        back_is_zero_ = false;
    }
}

uint8_t* Screen::BeginFrameStore(uint8_t z) {
    if (!depth_.empty()) return nullptr;
    if (!(depth_floor_ < z || (depth_floor_ == z && back_is_zero_))) return nullptr;
    depth_floor_ = z;
    back_is_zero_ = false;
    return buffers_[(front_ + 1) % 2];
}

void Screen::StoreRows(std::size_t first_row, std::size_t n_rows, uint8_t z, const uint8_t* bgr) {
    uint8_t* back = buffers_[(front_ + 1) % 2];
    if (first_row == 0 && n_rows == height_) {
        if (uint8_t* dst = BeginFrameStore(z)) {
            std::memcpy(dst, bgr, width_ * height_ * 3);
            return;
        }
    }
    MaterialiseDepth();
    back_is_zero_ = false;
    for (std::size_t r = 0; r < n_rows; ++r) {
        const std::size_t y = first_row + r;
        uint8_t* drow = &depth_[y * width_];
        bool all_below = true;
        for (std::size_t x = 0; x < width_ && all_below; ++x) all_below = drow[x] < z;
        if (all_below) {  // the `<` branch for the whole row
            std::memcpy(&back[3 * y * width_], bgr + 3 * r * width_, 3 * width_);
            std::memset(drow, z, width_);
            continue;
        }
        for (std::size_t x = 0; x < width_; ++x) {
            const std::size_t px = y * width_ + x;
            const uint8_t* src = bgr + 3 * (r * width_ + x);
            if (depth_[px] < z) {
                std::memcpy(&back[3 * px], src, 3);
                depth_[px] = z;
            } else if (depth_[px] == z) {
                for (int i = 0; i < 3; ++i) back[3 * px + i] = std::max(src[i], back[3 * px + i]);
            }
        }
    }
}

// src/screen.h:81-99: clear the old front buffer, make the back buffer the front one,
// present it.  Presenting = writing the file (the reference blits to the console).  The
// depth buffer is left alone, as in the reference.
void Screen::DrawScreen() {
    std::memset(buffers_[front_], 0, width_ * height_ * 3);
    front_ = (front_ + 1) % 2;
    back_is_zero_ = true;
    if (output_path_.empty()) return;
    const bool ppm = output_path_.size() > 4 && output_path_.compare(output_path_.size() - 4, 4, ".ppm") == 0;
    if (ppm) WritePpm(output_path_, buffers_[front_], width_, height_);
    else WriteBmp(output_path_, buffers_[front_], width_, height_);
}

namespace {
void put_u32(uint8_t* p, uint32_t v) { for (int i = 0; i < 4; ++i) p[i] = static_cast<uint8_t>(v >> (8 * i)); }
void put_u16(uint8_t* p, uint16_t v) { p[0] = static_cast<uint8_t>(v); p[1] = static_cast<uint8_t>(v >> 8); }
}  // namespace

// 24-bit BI_RGB bitmap with a positive height (bottom-up), the header the reference
// fills in src/screen.h:102-112; rows padded to 4 bytes, biSizeImage set correctly.
void WriteBmp(const std::string& path, const uint8_t* bgr, std::size_t width, std::size_t height) {
    const std::size_t pad = (4 - (3 * width) % 4) % 4;
    const std::size_t row = 3 * width + pad;
    uint8_t hdr[54] = {0};
    hdr[0] = 'B'; hdr[1] = 'M';
    put_u32(hdr + 2, static_cast<uint32_t>(54 + row * height));
    put_u32(hdr + 10, 54);
    put_u32(hdr + 14, 40);
    put_u32(hdr + 18, static_cast<uint32_t>(width));
    put_u32(hdr + 22, static_cast<uint32_t>(height));
    put_u16(hdr + 26, 1);
    put_u16(hdr + 28, 24);
    put_u32(hdr + 34, static_cast<uint32_t>(row * height));
    put_u32(hdr + 38, 2835);
    put_u32(hdr + 42, 2835);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    f.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
    if (pad == 0) {  // the frame is the file's pixel array as it stands
        f.write(reinterpret_cast<const char*>(bgr), static_cast<std::streamsize>(3 * width * height));
        return;
    }
    const char zeros[4] = {0, 0, 0, 0};
    for (std::size_t y = 0; y < height; ++y) {
        f.write(reinterpret_cast<const char*>(bgr + 3 * width * y), static_cast<std::streamsize>(3 * width));
        f.write(zeros, static_cast<std::streamsize>(pad));
    }
}

// Binary PPM is top-down RGB: flip the rows, swap the channels.
void WritePpm(const std::string& path, const uint8_t* bgr, std::size_t width, std::size_t height) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    f << "P6\n" << width << " " << height << "\n255\n";
    std::vector<uint8_t> row(3 * width);
    for (std::size_t y = height; y-- > 0;) {
        const uint8_t* src = bgr + 3 * width * y;
        for (std::size_t x = 0; x < width; ++x) {
            row[3 * x + 0] = src[3 * x + 2];
            row[3 * x + 1] = src[3 * x + 1];
            row[3 * x + 2] = src[3 * x + 0];
        }
        f.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size()));
    }
}

void Tracer(const Color& bgr, Screen& scr, const std::vector<std::unique_ptr<Object>>& objs,
            const std::vector<std::unique_ptr<PointLight>>& lights) {
    Tracer(bgr, scr, objs, lights, TracerOptions{});
}

namespace {
// One device context per device for the life of the process, with the scene it holds.
struct DeviceSlot {
    rt_ctx* ctx = nullptr;
    std::vector<rt_prim> prims;
    std::vector<rt_light> lights;
    std::vector<uint8_t> staging;  // for frames that cannot go straight into the Screen
};
std::map<int, DeviceSlot>& device_slots() {
    static std::map<int, DeviceSlot> slots;
    return slots;
}
template <typename T>
bool same_bytes(const std::vector<T>& a, const std::vector<T>& b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
}
}  // namespace

void ReleaseTracerDevices() {
    for (auto& kv : device_slots()) rt_destroy(kv.second.ctx);
    device_slots().clear();
}

void Tracer(const Color& bgr, Screen& scr, const std::vector<std::unique_ptr<Object>>& objs,
            const std::vector<std::unique_ptr<PointLight>>& lights, const TracerOptions& opt) {
    // The view frustum of src/main.cc:126; the device recomputes the pixel pitch from the
    // same three numbers.
    ViewFrustum vfr{scr.Width(), scr.Height(), opt.fov_deg, 1000.0};

    std::vector<rt_prim> prims;
    prims.reserve(objs.size());
    for (const auto& o : objs) prims.push_back(o->Flatten());
    std::vector<rt_light> lts;
    lts.reserve(lights.size());
    for (const auto& l : lights) lts.push_back(l->Flatten());

    using clock = std::chrono::steady_clock;
    const auto since = [](clock::time_point t) { return std::chrono::duration<double>(clock::now() - t).count(); };
    TracerTimings tm;
    DeviceSlot& slot = device_slots()[opt.device];
    if (!slot.ctx) {
        const auto t = clock::now();
        if (rt_create(opt.device, &slot.ctx) != RT_OK) {
            slot.ctx = nullptr;
            throw std::runtime_error(std::string("rt_create: ") + rt_last_error(nullptr));
        }
        tm.create_s = since(t);
    }
    rt_ctx* ctx = slot.ctx;
    auto fail = [&](const char* what) { throw std::runtime_error(std::string(what) + ": " + rt_last_error(ctx)); };
    if (!same_bytes(prims, slot.prims) || !same_bytes(lts, slot.lights)) {
        slot.prims.clear();
        const auto t = clock::now();
        if (rt_upload_scene(ctx, prims.data(), static_cast<int32_t>(prims.size()), lts.data(),
                            static_cast<int32_t>(lts.size())) != RT_OK)
            fail("rt_upload_scene");
        slot.prims = prims;
        slot.lights = lts;
        tm.upload_s = since(t);
    }

    rt_params p{};
    p.width = static_cast<int32_t>(vfr.Width());
    p.height = static_cast<int32_t>(vfr.Height());
    p.max_depth = opt.max_depth;
    p.band_rows = p.height;
    p.band_first = 0;
    p.band_step = 1;
    p.band_count = 1;
    p.fov_deg = vfr.Angle();
    p.bgr[0] = bgr.r(); p.bgr[1] = bgr.g(); p.bgr[2] = bgr.b();

    // StorePixel(x, y, 1, colour) for every pixel: when that replaces the whole back buffer
    // the device's frame is copied straight into it
    const auto t = clock::now();
    if (uint8_t* dst = scr.BeginFrameStore(1)) {
        if (rt_render(ctx, &p, dst, nullptr, nullptr, nullptr, nullptr) != RT_OK) fail("rt_render");
    } else {
        slot.staging.resize(static_cast<std::size_t>(p.width) * p.height * 3);
        if (rt_render(ctx, &p, slot.staging.data(), nullptr, nullptr, nullptr, nullptr) != RT_OK) fail("rt_render");
        scr.StoreRows(0, scr.Height(), 1, slot.staging.data());
    }
    tm.render_s = since(t);
    if (opt.timings) *opt.timings = tm;
}
