// Host-side builder of the sphere tree the kernels walk for large scenes.
//
// The reference tests every object against every ray in container order
// (src/main.cc:37-47, src/objects.h:31-47).  For many spheres the kernels instead walk a
// binary tree of bounding boxes, stored in depth-first order with a skip index per node
// ("if this subtree cannot be hit, continue at node skip"), which needs no stack and keeps
// the node index uniform across a wave.  The walk only decides which exact tests can be left
// out; container-order semantics are restored from the original indices the leaves carry
// (smallest index wins a tie in t; the smallest-index occluder is the "first" one).
// How the tree is split only changes how much the walk can skip, never a result.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "whitted_device.h"

namespace rt {

struct TreeSphere {
    double c[3];
    double r;
    int32_t index;  // position in the caller's object list
};

namespace detail {

constexpr int kLeafGroup = 4;

inline float up(float v) { return std::nextafter(v, INFINITY); }

inline void put_int(float& slot, int32_t v) { std::memcpy(&slot, &v, sizeof v); }

// Node record, 32 bytes like FGeom (see whitted_device.h):
//   interior  v[0..2] box centre, v[3..5] half extents, v[6] skip index (bits, > 0),
//             v[7] smallest original index below the node (bits, >= 0)
//   group     a box directly over 1..kLeafGroup spheres, whose leaf records follow it:
//             as an interior node, but v[6] = -(number of leaves) (bits, < 0)
//   leaf      v[0..2] sphere centre, v[3] radius^2 (as in the linear table),
//             v[7] = ~index of the sphere in the caller's list (bits, < 0)
// Leaves only ever follow a group node; the walk reads a group's leaves four at a time, so the
// table ends with kLeafGroup - 1 spare records.
// Returns the smallest original index below the node; *extent1 grows to |centre|_1 +
// |half extents|_1 of the largest box (the filter's per-ray error bound needs it).
inline int build(std::vector<TreeSphere>& s, int lo, int hi, std::vector<FGeom>& out, double* extent1) {
    const int self = static_cast<int>(out.size());
    out.emplace_back();
    std::memset(&out[self], 0, sizeof(FGeom));
    // split at the median of the centres along the widest axis
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double bmn[3] = {INFINITY, INFINITY, INFINITY}, bmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; ++i)
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], s[i].c[k]);
            mx[k] = std::max(mx[k], s[i].c[k]);
            bmn[k] = std::min(bmn[k], s[i].c[k] - s[i].r);
            bmx[k] = std::max(bmx[k], s[i].c[k] + s[i].r);
        }
    // Split: binned surface-area heuristic over the centres (16 bins per axis, cost = area
    // of the box of each side x number of spheres in it); median of the widest axis when
    // no bin boundary separates the centres.
    int axis = 0;
    for (int k = 1; k < 3; ++k)
        if (mx[k] - mn[k] > mx[axis] - mn[axis]) axis = k;
    int mid = lo + (hi - lo) / 2;
    bool by_median = true;
    if (hi - lo > 2) {
        constexpr int kBins = 16;
        double best = INFINITY, best_pos = 0.0;
        int best_axis = -1;
        for (int k = 0; k < 3; ++k) {
            const double ext = mx[k] - mn[k];
            if (!(ext > 0.0)) continue;
            struct Bin { double mn[3], mx[3]; int n; } bins[kBins];
            for (Bin& b : bins) { b.n = 0; for (int j = 0; j < 3; ++j) { b.mn[j] = INFINITY; b.mx[j] = -INFINITY; } }
            for (int i = lo; i < hi; ++i) {
                int bi = static_cast<int>((s[i].c[k] - mn[k]) / ext * kBins);
                bi = std::min(std::max(bi, 0), kBins - 1);
                Bin& b = bins[bi];
                ++b.n;
                for (int j = 0; j < 3; ++j) {
                    b.mn[j] = std::min(b.mn[j], s[i].c[j] - s[i].r);
                    b.mx[j] = std::max(b.mx[j], s[i].c[j] + s[i].r);
                }
            }
            const auto area = [](const double* a, const double* b) {
                const double x = b[0] - a[0], y = b[1] - a[1], z = b[2] - a[2];
                return x * y + y * z + z * x;
            };
            double rmn[kBins][3], rmx[kBins][3];
            int rn[kBins];
            double amn[3] = {INFINITY, INFINITY, INFINITY}, amx[3] = {-INFINITY, -INFINITY, -INFINITY};
            int an = 0;
            for (int b = kBins - 1; b >= 0; --b) {  // suffix boxes
                an += bins[b].n;
                for (int j = 0; j < 3; ++j) { amn[j] = std::min(amn[j], bins[b].mn[j]); amx[j] = std::max(amx[j], bins[b].mx[j]); }
                rn[b] = an;
                for (int j = 0; j < 3; ++j) { rmn[b][j] = amn[j]; rmx[b][j] = amx[j]; }
            }
            double lmn[3] = {INFINITY, INFINITY, INFINITY}, lmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            int ln = 0;
            for (int b = 0; b + 1 < kBins; ++b) {  // split between bin b and b + 1
                ln += bins[b].n;
                for (int j = 0; j < 3; ++j) { lmn[j] = std::min(lmn[j], bins[b].mn[j]); lmx[j] = std::max(lmx[j], bins[b].mx[j]); }
                if (ln == 0 || rn[b + 1] == 0) continue;
                const double cost = area(lmn, lmx) * ln + area(rmn[b + 1], rmx[b + 1]) * rn[b + 1];
                if (cost < best) { best = cost; best_axis = k; best_pos = mn[k] + ext * (b + 1) / kBins; }
            }
        }
        if (best_axis >= 0) {
            const auto it = std::partition(s.begin() + lo, s.begin() + hi, [&](const TreeSphere& t) {
                int bi = static_cast<int>((t.c[best_axis] - mn[best_axis]) / (mx[best_axis] - mn[best_axis]) * kBins);
                bi = std::min(std::max(bi, 0), kBins - 1);
                return mn[best_axis] + (mx[best_axis] - mn[best_axis]) * (bi + 1) / kBins <= best_pos;
            });
            const int m = static_cast<int>(it - s.begin());
            // keep both sides above an eighth of the range: bounds the depth of the tree
            if (m - lo >= (hi - lo) / 8 && hi - m >= (hi - lo) / 8 && m > lo && m < hi) { mid = m; by_median = false; }
        }
    }
    if (by_median)
        std::nth_element(s.begin() + lo, s.begin() + mid, s.begin() + hi,
                         [axis](const TreeSphere& a, const TreeSphere& b) { return a.c[axis] < b.c[axis]; });
    // box of the member spheres around a FLOAT centre (so that the centre is exact on the
    // device), half extents measured from it and rounded up
    FGeom g;
    std::memset(&g, 0, sizeof g);
    double e1 = 0.0;
    for (int k = 0; k < 3; ++k) {
        const float cf = static_cast<float>(0.5 * (bmn[k] + bmx[k]));
        const double h = std::max(bmx[k] - cf, cf - bmn[k]);
        const float hf = up(static_cast<float>(h * (1.0 + 1e-6)));
        g.v[k] = cf;
        g.v[3 + k] = hf;
        e1 += std::fabs(static_cast<double>(cf)) + hf;
    }
    *extent1 = std::max(*extent1, e1);
    int min_index;
    if (hi - lo <= kLeafGroup) {
        // a box directly over up to kLeafGroup spheres: cheaper than two more levels of boxes
        min_index = s[lo].index;
        for (int i = lo; i < hi; ++i) {
            const TreeSphere& t = s[i];
            FGeom leaf;
            std::memset(&leaf, 0, sizeof leaf);
            for (int k = 0; k < 3; ++k) leaf.v[k] = static_cast<float>(t.c[k]);
            leaf.v[3] = static_cast<float>(t.r * t.r);  // the filter allows for this rounding
            put_int(leaf.v[7], ~t.index);
            out.push_back(leaf);
            min_index = std::min(min_index, t.index);
        }
        put_int(g.v[6], -(hi - lo));
    } else {
        const int min_l = build(s, lo, mid, out, extent1);
        const int min_r = build(s, mid, hi, out, extent1);
        min_index = std::min(min_l, min_r);
        put_int(g.v[6], static_cast<int32_t>(out.size()));
    }
    put_int(g.v[7], min_index);
    out[self] = g;
    return min_index;
}

}  // namespace detail

// Nodes in depth-first order (followed by kLeafGroup - 1 spare records, not counted in
// *n_nodes); empty when there are no spheres.  *extent1 receives the largest |centre|_1 +
// |half extents|_1 over the boxes.
inline std::vector<FGeom> build_sphere_tree(std::vector<TreeSphere> spheres, double* extent1, int* n_nodes) {
    std::vector<FGeom> out;
    *extent1 = 0.0;
    *n_nodes = 0;
    if (spheres.empty()) return out;
    out.reserve(2 * spheres.size() + detail::kLeafGroup);
    detail::build(spheres, 0, static_cast<int>(spheres.size()), out, extent1);
    *n_nodes = static_cast<int>(out.size());
    for (int i = 1; i < detail::kLeafGroup; ++i) out.push_back(out.back());
    return out;
}

}  // namespace rt
