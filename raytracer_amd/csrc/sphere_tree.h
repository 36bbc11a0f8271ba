// Host-side builder of the sphere tree the kernels walk for large scenes.
//
// The reference tests every object against every ray in container order
// (src/main.cc:37-47, src/objects.h:31-47).  For many spheres the kernels instead walk a
// binary tree of bounding spheres, stored in depth-first order with a skip index per node
// ("if this subtree cannot be hit, continue at node skip"), which needs no stack and keeps
// the node index uniform across a wave.  The walk only decides which exact tests can be left
// out; container-order semantics are restored from the original indices the leaves carry
// (smallest index wins a tie in t; the smallest-index occluder is the "first" one).
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

inline float up(float v) { return std::nextafter(v, INFINITY); }

inline void put_int(float& slot, int32_t v) { std::memcpy(&slot, &v, sizeof v); }

// Node record (same 32-byte layout as FGeom, see whitted_device.h):
//   v[0..2] centre, v[3] radius^2, v[4] skip index (bits), v[5] original index of the leaf's
//   sphere or -1 for an interior node (bits), v[6] smallest original index below the node (bits)
inline int build(std::vector<TreeSphere>& s, int lo, int hi, std::vector<FGeom>& out) {
    const int self = static_cast<int>(out.size());
    out.emplace_back();
    std::memset(&out[self], 0, sizeof(FGeom));
    if (hi - lo == 1) {
        const TreeSphere& t = s[lo];
        FGeom g;
        std::memset(&g, 0, sizeof g);
        for (int k = 0; k < 3; ++k) g.v[k] = static_cast<float>(t.c[k]);
        g.v[3] = static_cast<float>(t.r * t.r);  // as in the linear table: the filter allows for its rounding
        put_int(g.v[4], self + 1);
        put_int(g.v[5], t.index);
        put_int(g.v[6], t.index);
        out[self] = g;
        return t.index;
    }
    // split at the median of the centres along the widest axis
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; ++i)
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], s[i].c[k]);
            mx[k] = std::max(mx[k], s[i].c[k]);
        }
    int axis = 0;
    for (int k = 1; k < 3; ++k)
        if (mx[k] - mn[k] > mx[axis] - mn[axis]) axis = k;
    const int mid = lo + (hi - lo) / 2;
    std::nth_element(s.begin() + lo, s.begin() + mid, s.begin() + hi,
                     [axis](const TreeSphere& a, const TreeSphere& b) { return a.c[axis] < b.c[axis]; });
    // bounding sphere of the member spheres: centre of their box, radius to the farthest surface
    double bmn[3] = {INFINITY, INFINITY, INFINITY}, bmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; ++i)
        for (int k = 0; k < 3; ++k) {
            bmn[k] = std::min(bmn[k], s[i].c[k] - s[i].r);
            bmx[k] = std::max(bmx[k], s[i].c[k] + s[i].r);
        }
    double C[3], c1 = 0.0;
    float Cf[3];
    for (int k = 0; k < 3; ++k) {
        C[k] = 0.5 * (bmn[k] + bmx[k]);
        Cf[k] = static_cast<float>(C[k]);
        c1 += std::fabs(C[k]);
    }
    double R = 0.0;
    for (int i = lo; i < hi; ++i) {
        // distance from the FLOAT centre, so that its rounding is inside the bound
        double dd = 0.0;
        for (int k = 0; k < 3; ++k) dd += (s[i].c[k] - Cf[k]) * (s[i].c[k] - Cf[k]);
        R = std::max(R, std::sqrt(dd) + s[i].r);
    }
    R = R * (1.0 + 1e-6) + 1e-12 * c1;            // rounding of the computation above, generously
    float Rf = up(static_cast<float>(R));
    float R2 = up(up(Rf * Rf));
    const int min_l = build(s, lo, mid, out);
    const int min_r = build(s, mid, hi, out);
    FGeom g;
    std::memset(&g, 0, sizeof g);
    for (int k = 0; k < 3; ++k) g.v[k] = Cf[k];
    g.v[3] = R2;
    put_int(g.v[4], static_cast<int32_t>(out.size()));
    put_int(g.v[5], -1);
    put_int(g.v[6], std::min(min_l, min_r));
    out[self] = g;
    return std::min(min_l, min_r);
}

}  // namespace detail

// Nodes in depth-first order; empty when there are no spheres.
inline std::vector<FGeom> build_sphere_tree(std::vector<TreeSphere> spheres) {
    std::vector<FGeom> out;
    if (spheres.empty()) return out;
    out.reserve(2 * spheres.size());
    detail::build(spheres, 0, static_cast<int>(spheres.size()), out);
    return out;
}

}  // namespace rt
