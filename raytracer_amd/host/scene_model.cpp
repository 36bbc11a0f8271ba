// Host side of the drop-in: camera bookkeeping, the headless Screen and the Tracer()
// call that hands the flattened scene to the device library.
#include "scene_model.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
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
    for (auto& b : buffers_) b.assign(width * height * 3, 0);
    depth_.assign(width * height, 0);
}

// The reference walks (row, column) but passes them to StorePixel as (x, y)
// (src/screen.h:47-53), which only works for square frames; here x is the column.
void Screen::ClearScreenColor(const Color& c) {
    for (std::size_t y = 0; y < height_; ++y)
        for (std::size_t x = 0; x < width_; ++x) StorePixel(x, y, 0, c);
}

// src/screen.h:55-73: nearer (larger z) wins, equal z keeps the brighter channel.
void Screen::StorePixel(std::size_t x, std::size_t y, uint8_t z, const Color& c) {
    std::vector<uint8_t>& back = buffers_[(front_ + 1) % 2];
    const std::size_t px = y * width_ + x;
    uint8_t* dst = &back[3 * px];
    const uint8_t v[3] = {c.blue(), c.green(), c.red()};
    if (depth_[px] < z) {
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
        depth_[px] = z;
    } else if (depth_[px] == z) {
        for (int i = 0; i < 3; ++i) dst[i] = std::max(v[i], dst[i]);
    }
}

void Screen::StoreRows(std::size_t first_row, std::size_t n_rows, uint8_t z, const uint8_t* bgr) {
    std::vector<uint8_t>& back = buffers_[(front_ + 1) % 2];
    for (std::size_t r = 0; r < n_rows; ++r) {
        const std::size_t y = first_row + r;
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
// present it.  Presenting = writing the file (the reference blits to the console).
void Screen::DrawScreen() {
    std::fill(buffers_[front_].begin(), buffers_[front_].end(), 0);
    front_ = (front_ + 1) % 2;
    std::fill(depth_.begin(), depth_.end(), 0);
    if (output_path_.empty()) return;
    const bool ppm = output_path_.size() > 4 && output_path_.compare(output_path_.size() - 4, 4, ".ppm") == 0;
    if (ppm) WritePpm(output_path_, buffers_[front_].data(), width_, height_);
    else WriteBmp(output_path_, buffers_[front_].data(), width_, height_);
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

    rt_ctx* ctx = nullptr;
    if (rt_create(opt.device, &ctx) != RT_OK) throw std::runtime_error(std::string("rt_create: ") + rt_last_error(nullptr));
    auto fail = [&](const char* what) {
        std::string msg = std::string(what) + ": " + rt_last_error(ctx);
        rt_destroy(ctx);
        throw std::runtime_error(msg);
    };
    if (rt_upload_scene(ctx, prims.data(), static_cast<int32_t>(prims.size()), lts.data(),
                        static_cast<int32_t>(lts.size())) != RT_OK)
        fail("rt_upload_scene");

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

    std::vector<uint8_t> frame(static_cast<std::size_t>(p.width) * p.height * 3);
    if (rt_render(ctx, &p, frame.data(), nullptr, nullptr, nullptr, nullptr) != RT_OK) fail("rt_render");
    rt_destroy(ctx);
    scr.StoreRows(0, scr.Height(), 1, frame.data());  // StorePixel(x, y, 1, colour) for every pixel
}
