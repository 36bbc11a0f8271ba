// Host-side screen bounds of a scene: which 8x8 pixel tiles of a strip can hold a hit at all.
//
// Every primary ray starts at the camera (the origin, looking along +z, src/view_frustum.h:37-46),
// so the pixels in which a sphere or a plane can be hit form a region of the image that follows
// from the object alone.  For every row of tiles the table below holds the range of tile
// columns outside of which no primary ray of the row hits anything: a wave whose tile lies
// outside writes the background colour (src/main.cc:49-53) and is done, without tracing.  The
// ranges only have to be supersets; inside them every pixel is traced as usual.
//
// Sphere (centre c, radius r): a ray with direction d, d_z > 0, that meets the sphere projects
// along y to a ray of the (z, x) plane that meets the projected disc, so X = d_x / d_z lies
// between the tangents of that disc as seen from the origin, and Y = d_y / d_z likewise in the
// (z, y) plane.  The radius is taken as sqrt(r^2 + 1e-13 |c|_1^2) (1 + 1e-9): what the
// reference's rounded test (src/objects.h:124-131) can accept beyond the exact sphere.
// Plane (point p, unit normal n): hit only where t = (p . n) / (d . n) > 0 (src/objects.h:91-101),
// i.e. where X n_x + Y n_y + n_z has the sign of p . n: a half plane of the image.
// Pixel ranges are widened by one pixel on every side.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "../../include/rt_hip.h"
#include "whitted_device.h"

namespace rt {

namespace detail {

// Range [lo, hi] of tan(angle from the +a axis) over the rays from the origin, pointing into
// a > 0, that meet the disc of centre (a, b) and radius r; false = none.  Unbounded ends are
// returned as infinities.
inline bool forward_slope_range(double a, double b, double r, double* lo, double* hi) {
    const double half_pi = 1.57079632679489661923;
    *lo = -INFINITY;
    *hi = INFINITY;
    const double rho = std::hypot(a, b);
    if (!(rho > r * (1.0 + 1e-9))) return true;  // the origin lies in the disc: every ray
    const double t0 = std::atan2(b, a);
    const double al = std::asin(std::min(1.0, r / rho)) + 1e-9;
    const double l = t0 - al, h = t0 + al;
    if (h <= -half_pi || l >= half_pi) return false;
    if (l > -half_pi + 1e-7) *lo = std::tan(l);
    if (h < half_pi - 1e-7) *hi = std::tan(h);
    return true;
}

// Frame row of strip row k (the kernel's frame_row()).
inline int64_t strip_to_frame_row(const rt_params& p, int64_t k) {
    const int64_t band = k / p.band_rows;
    return (p.band_first + band * p.band_step) * static_cast<int64_t>(p.band_rows) + (k - band * p.band_rows);
}

}  // namespace detail

// Smallest shift that brings every tile column of a strip of this width below 256.
inline int sky_unit_shift(int width) {
    const int tiles_x = (width + kTile - 1) / kTile;
    int s = 0;
    while (((tiles_x - 1) >> s) > 255) ++s;
    return s;
}

// The order in which a launch with one wave per tile goes through its rows of tiles
// (KArgs::row_order): block row b renders tile row (first + b) mod tiles_y, or (first - b) mod
// tiles_y with bit 30 set.  A wave on a sphere walks a ray tree, a wave on a plane or on the sky
// is short; a launch whose last blocks are long waves ends with most of the chip idle, so the
// rows that hold spheres go first and the launch ends on short waves.  `sphere_tiles[ty]` = how
// many tiles of row ty some sphere's bounds cover.  The rows with spheres form a band [lo, hi]
// of the strip (interleaved strips: of its rows); the launch starts at the end of the band that
// has the shorter rest behind it and runs through the band, then through the longer rest, so
// that as many short waves as possible are still to come when the last long ones start.
inline int32_t tile_row_order(const uint32_t* sphere_tiles, int tiles_y) {
    int lo = -1, hi = -1;
    for (int ty = 0; ty < tiles_y; ++ty)
        if (sphere_tiles[ty] > 0) {
            if (lo < 0) lo = ty;
            hi = ty;
        }
    if (lo < 0 || tiles_y > 0xffff) return 0;
    const int below = lo, above = tiles_y - 1 - hi;
    // upwards from lo: band, rows above, rows below last; downwards from hi: band, below, above last
    return (below >= above) ? lo : (hi | 0x40000000);
}
inline int tile_row_of_block(int32_t order, int b, int tiles_y) {
    int ty = (order & 0x40000000) ? (order & 0xffff) - b : (order & 0xffff) + b;
    if (ty < 0) ty += tiles_y;
    if (ty >= tiles_y) ty -= tiles_y;
    return ty;
}

// Fills `out` for the strip `p` describes (rows = rt_strip_rows(p), pixel pitch pw x ph), indexed
// by tile row, and - when asked - `sphere_tiles[ty]` (kSkyRows entries) with the number of tiles
// of row ty inside some sphere's bounds.  A strip of more than kSkyRows tile rows gets the table
// that leaves every tile in.
inline void build_sky_table(const rt_prim* prims, int n_prims, const rt_params& p, int rows, double pw, double ph,
                            SkyTable* out, uint32_t* sphere_tiles = nullptr) {
    using namespace detail;
    const int tiles_y = (rows + kTile - 1) / kTile;
    const int shift = sky_unit_shift(p.width);
    for (int e = 0; e < kSkyRows; ++e) out->cols[e] = 0xff00u;  // lo = 0, hi = 255: every column
    if (sphere_tiles)
        for (int e = 0; e < kSkyRows; ++e) sphere_tiles[e] = 0;
    if (tiles_y > kSkyRows) return;
    for (int ty = 0; ty < tiles_y; ++ty) {
        out->cols[ty] = 0x0001u;  // lo = 1, hi = 0: none
        const int64_t k0 = static_cast<int64_t>(ty) * kTile, k1 = std::min<int64_t>(rows, k0 + kTile) - 1;
        // image-plane Y of the tile row's pixel rows, one pixel wider on both sides (rows ascend with k)
        const double ya = (static_cast<double>(strip_to_frame_row(p, k0) - p.height / 2) + 0.5 - 1.0) * ph;
        const double yb = (static_cast<double>(strip_to_frame_row(p, k1) - p.height / 2) + 0.5 + 1.0) * ph;
        double xlo = INFINITY, xhi = -INFINITY;  // union of the objects' X ranges over this tile row
        // (once the row is covered from end to end only the spheres' shares are still of interest)
        for (int i = 0; i < n_prims && (sphere_tiles || !(xlo == -INFINITY && xhi == INFINITY)); ++i) {
            const rt_prim& s = prims[i];
            double lo = -INFINITY, hi = INFINITY;
            if (s.type == RT_PRIM_SPHERE) {
                const double c1 = std::fabs(s.p[0]) + std::fabs(s.p[1]) + std::fabs(s.p[2]);
                const double rg = std::sqrt(s.q[0] * s.q[0] + 1.0e-13 * c1 * c1) * (1.0 + 1e-9);
                double vlo, vhi;
                if (!forward_slope_range(s.p[2], s.p[1], rg, &vlo, &vhi)) continue;
                if (vhi < ya || vlo > yb) continue;
                if (!forward_slope_range(s.p[2], s.p[0], rg, &lo, &hi)) continue;
                if (sphere_tiles) {  // its share of this row, in tiles
                    const double half = static_cast<double>(p.width / 2);
                    const double a = std::max(0.0, lo / pw + half), b = std::min(static_cast<double>(p.width), hi / pw + half);
                    if (b > a) sphere_tiles[ty] += static_cast<uint32_t>((b - a) / kTile) + 1u;
                }
            } else {
                // hit only where sg (X n_x + Y n_y + n_z) > 0, sg = sign(p . n); an uncertain sign of
                // p . n leaves the whole row in
                const double num = s.p[0] * s.q[0] + s.p[1] * s.q[1] + s.p[2] * s.q[2];
                const double p1 = std::fabs(s.p[0]) + std::fabs(s.p[1]) + std::fabs(s.p[2]);
                if (std::fabs(num) > 1e-12 * p1) {
                    const double sg = num > 0 ? 1.0 : -1.0;
                    const double A = sg * s.q[0];
                    // the largest value over the row's Y, and a margin for the rounding of the
                    // reference's own dot products
                    const double B = sg * s.q[2] + std::max(sg * s.q[1] * ya, sg * s.q[1] * yb) +
                                     1e-9 * (1.0 + std::fabs(ya) + std::fabs(yb));
                    if (std::fabs(A) <= 1e-12) {
                        if (B <= 0.0) continue;  // nowhere in the row
                    } else if (A > 0) {
                        lo = -B / A;  // hit only at larger X
                        lo -= 1e-9 * (1.0 + std::fabs(lo));
                    } else {
                        hi = -B / A;  // hit only at smaller X
                        hi += 1e-9 * (1.0 + std::fabs(hi));
                    }
                }
            }
            xlo = std::min(xlo, lo);
            xhi = std::max(xhi, hi);
        }
        if (!(xlo <= xhi)) continue;  // nothing can be hit in this row
        // X = (xi + 0.5) pw, xi = x - width / 2; one pixel wider on both sides
        const double half = static_cast<double>(p.width / 2);
        double fx0 = std::floor(xlo / pw - 0.5) + half - 1.0, fx1 = std::ceil(xhi / pw - 0.5) + half + 1.0;
        if (!(fx0 > 0.0)) fx0 = 0.0;                      // also for -inf and NaN
        if (!(fx1 < p.width - 1.0)) fx1 = p.width - 1.0;
        if (fx0 > fx1) continue;
        const uint32_t t0 = (static_cast<uint32_t>(fx0) / kTile) >> shift, t1 = (static_cast<uint32_t>(fx1) / kTile) >> shift;
        out->cols[ty] = static_cast<uint16_t>((t0 & 0xffu) | ((t1 & 0xffu) << 8));
    }
}

}  // namespace rt
