// Host-side builder of the anchor grids.
//
// Every shadow feeler of a light runs along a line through that light, every primary ray
// along a line through the camera.  For such an "anchor" point A the lines through it are a
// two-parameter family; they are sorted into the cells of three cube faces (face k holds the
// lines whose direction d has its largest component along axis k, cell coordinates
// u = d_i / d_k, v = d_j / d_k in [-1, 1]), and every cell lists the spheres that some line of
// the cell may touch - in container order, so that the first sphere of the list a feeler
// hits is the first occluder among the spheres the way the reference finds it
// (src/objects.h:35-46).  The kernels then test the few spheres of one cell instead of
// walking the whole scene.  A list only has to be a superset: every entry still goes through
// the kernel's own filters and the reference's double test.
//
// What a list must hold.  The reference's test of a ray (origin o, unit direction d, both
// double) against a sphere (c, r) returns a value only if its rounded d2 = L.L - tca^2 is at
// most r^2, and then the distance D of c to the line through the anchor with direction d
// satisfies D^2 <= r^2 + 1e-13 Sb^2, Sb >= |o|_1 + |c|_1 (the bound the kernel's
// light-anchored line test is built on: rounding of d2, and the 4e-16 |A - o| by which a
// feeler's line passes its light).  The grids are built for Sb <= sb_max: spheres are
// rasterised with r_g^2 = r^2 + 1e-13 sb_max^2, and the kernel consults a grid only for rays
// within that bound (KArgs::grid_eray_max), the others take the general path.
//
// Footprint of a sphere on face k: a line that meets the sphere projects, along axis j, to a
// line through the origin of the (k, i) plane that meets the projected disc (centre
// (L_k, L_i), L = c - A, radius r_g), so u = tan(angle) lies within the two tangents of that
// disc, and likewise v: the rectangle of the two ranges contains every line of the face that
// meets the sphere.  The kernel finds its cell from the float image of d (error below 1e-6 in
// u and v); the rectangle is widened by 4e-6 before it is cut into cells.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "whitted_device.h"

namespace rt {

struct GridSphere {
    double c[3];
    double r2;       // radius^2 as the reference computes it
    int32_t index;   // position in the caller's object list; the input is ascending in it
    float transparency;
};

struct AnchorGrids {
    int res = 0;
    float eray_max = 0.0f;
    std::vector<GridCell> cells;     // [anchor][face][v][u]
    std::vector<GridEntry> entries;
};

// The constants of the light-anchored tests for the sphere (centre - anchor = L, r2) - see
// line_miss_from_light() and classify_sphere() in whitted_exact.inc.h (U = 6e-8 >= 2^-24):
//   out[0..2] L in float
//   out[3]    thr  = |L|^2 - r^2 - 24 U |L|_1^2, rounded down: the float value of thr - (L.d)^2 is
//             off by at most 21.3 U |L|_1^2 for a direction within 6 U of the true one
//   cold[0]   r^2 in float
//   cold[1]   ehit = bound on the error of r^2 - |L - (L.d) d|^2 evaluated in float, for lines
//             that pass within sqrt(r^2 + 50 U |L|_1^2) of the centre (all that thr lets through):
//             |P~ - P| <= 19.1 U |L|_1, so the squared length is off by at most
//             38.2 U rr |L|_1 + 365 U^2 |L|_1^2 + 3 U rr^2, plus U r^2 for cold[0]; stored with
//             a factor 1.5 to spare
inline void line_test_record(const double L[3], double r2, float out[4], float* cold = nullptr) {
    const double U = 6.0e-8;
    double n2 = 0.0, n1 = 0.0;
    for (int c = 0; c < 3; ++c) {
        n2 += L[c] * L[c];
        n1 += std::fabs(L[c]);
        out[c] = static_cast<float>(L[c]);
    }
    const double thr = n2 - r2 - 24.0 * U * n1 * n1;
    float tf = static_cast<float>(thr);
    out[3] = std::nextafter(std::nextafter(tf, -INFINITY), -INFINITY);
    if (cold) {
        const double rr = std::sqrt(r2 + 50.0 * U * n1 * n1);
        const double ehit = 1.5 * (38.2 * U * rr * n1 + 365.0 * U * U * n1 * n1 + 3.0 * U * rr * rr + U * r2);
        cold[0] = static_cast<float>(r2);
        cold[1] = std::nextafter(static_cast<float>(ehit), INFINITY);
    }
}

namespace detail {

constexpr double kQuarter = 0.78539816339744830962;  // pi / 4
constexpr double kPi = 3.14159265358979323846;

// Range [lo, hi] of tan(angle) over the lines through the origin of a plane that meet the
// disc of centre (a, b) and radius r, cut to |tan| <= 1 (a little more); false = none.
inline bool slope_range(double a, double b, double r, double* lo, double* hi) {
    const double win = kQuarter + 1e-5;
    const double rho = std::hypot(a, b);
    *lo = -1.0;
    *hi = 1.0;
    if (!(rho > r * (1.0 + 1e-9))) return true;  // the origin lies in the disc: every line
    const double t0 = std::atan2(b, a);
    const double al = std::asin(std::min(1.0, r / rho)) + 1e-9;
    int found = 0;
    double flo = 0.0, fhi = 0.0;
    for (int m = -1; m <= 1; ++m) {  // lines, not rays: angles count modulo pi
        const double c = t0 + m * kPi;
        const double l = std::max(c - al, -win), h = std::min(c + al, win);
        if (l <= h) {
            if (found++) return true;  // two pieces: the whole face
            flo = l;
            fhi = h;
        }
    }
    if (!found) return false;
    *lo = std::tan(flo);
    *hi = std::tan(fhi);
    return true;
}

struct Footprint {
    int face, u0, u1, v0, v1;
};

inline int cell_of(double t, int res) {
    const double f = std::floor((t + 1.0) * (0.5 * res));
    return f < 0.0 ? 0 : (f > res - 1 ? res - 1 : static_cast<int>(f));
}

}  // namespace detail

// Grid resolution for a scene: 128 cells per sphere and face (the lists of a cell are walked by
// every lane on its own, so short lists count for more than a small table: 24 large spheres seen
// from 17 anchors take 17.5 ms per 4K depth-8 frame at 32 cells per edge, 17.0 at 56), at least
// 32, at most 128, and no more than keeps the cell table of all anchors within 1.4 MiB - the
// lanes of a wave read cells and lists of their own, and a table that does not stay in the
// 4 MiB L2 of an XCD next to the lists turns into memory traffic (1024 spheres, 9 anchors:
// 600 MB per 4K frame at 128 cells per edge, 26 MB at 64; 64 and 80 take the same time).
inline int anchor_grid_resolution(size_t n_spheres, size_t n_anchors) {
    int res = 32;
    while (res < 128 && static_cast<size_t>(res) * res < 128 * n_spheres) res += 8;
    while (res > 32 && 3ull * res * res * sizeof(GridCell) * n_anchors > 1468006ull) res -= 8;
    return res;
}

// Builds the grids of all anchors.  Returns false (and leaves *out empty) when the lists
// would exceed max_entries: the scene is then rendered without grids.
inline bool build_anchor_grids(const std::vector<GridSphere>& spheres, const std::vector<std::array<double, 3>>& anchors,
                               double sb_max, int res, size_t max_entries, AnchorGrids* out) {
    using namespace detail;
    out->res = 0;
    out->cells.clear();
    out->entries.clear();
    if (spheres.empty() || anchors.empty() || res < 1) return false;
    const double grow = 1.0e-13 * sb_max * sb_max;
    const double eps = 4.0e-6;
    const size_t per_anchor = 3ull * res * res;
    std::vector<GridCell> cells(per_anchor * anchors.size());
    std::vector<GridEntry> entries;
    std::vector<Footprint> fp;
    std::vector<uint32_t> owner;  // sphere of each footprint
    std::vector<uint32_t> fill(per_anchor);
    for (size_t ai = 0; ai < anchors.size(); ++ai) {
        GridCell* cell = cells.data() + ai * per_anchor;
        for (size_t i = 0; i < per_anchor; ++i) cell[i] = GridCell{0u, 0u};
        fp.clear();
        owner.clear();
        for (size_t si = 0; si < spheres.size(); ++si) {
            const GridSphere& s = spheres[si];
            const double L[3] = {s.c[0] - anchors[ai][0], s.c[1] - anchors[ai][1], s.c[2] - anchors[ai][2]};
            const double rg = std::sqrt(s.r2 + grow) * (1.0 + 1e-9);
            for (int face = 0; face < 3; ++face) {
                // face 0: k = x, (i, j) = (y, z); face 1: k = y, (x, z); face 2: k = z, (x, y)
                const int k = face, i = (face == 0) ? 1 : 0, j = (face == 2) ? 1 : 2;
                double ulo, uhi, vlo, vhi;
                if (!slope_range(L[k], L[i], rg, &ulo, &uhi)) continue;
                if (!slope_range(L[k], L[j], rg, &vlo, &vhi)) continue;
                Footprint f{face, cell_of(ulo - eps, res), cell_of(uhi + eps, res), cell_of(vlo - eps, res),
                            cell_of(vhi + eps, res)};
                fp.push_back(f);
                owner.push_back(static_cast<uint32_t>(si));
                for (int v = f.v0; v <= f.v1; ++v)
                    for (int u = f.u0; u <= f.u1; ++u) ++cell[(static_cast<size_t>(face) * res + v) * res + u].count;
            }
        }
        size_t total = entries.size();
        for (size_t i = 0; i < per_anchor; ++i) {
            cell[i].start = static_cast<uint32_t>(total);
            total += cell[i].count;
            fill[i] = 0;
        }
        if (total > max_entries) return false;
        entries.resize(total);
        // footprints come in sphere order, so every list ascends in the container index
        for (size_t n = 0; n < fp.size(); ++n) {
            const Footprint& f = fp[n];
            const GridSphere& s = spheres[owner[n]];
            const double L[3] = {s.c[0] - anchors[ai][0], s.c[1] - anchors[ai][1], s.c[2] - anchors[ai][2]};
            GridEntry e;
            std::memset(&e, 0, sizeof e);
            float rec[4], cold[2];
            line_test_record(L, s.r2, rec, cold);
            e.l[0] = rec[0]; e.l[1] = rec[1]; e.l[2] = rec[2];
            e.thr = rec[3];
            e.index = s.index;
            e.r2 = cold[0];
            e.ehit = cold[1];
            e.transparency = s.transparency;
            for (int v = f.v0; v <= f.v1; ++v)
                for (int u = f.u0; u <= f.u1; ++u) {
                    const size_t ci = (static_cast<size_t>(f.face) * res + v) * res + u;
                    entries[cell[ci].start + fill[ci]++] = e;
                }
        }
    }
    out->res = res;
    out->eray_max = static_cast<float>(grow) * 0.999f;
    out->cells.swap(cells);
    out->entries.swap(entries);
    return true;
}

}  // namespace rt
