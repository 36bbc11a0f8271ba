// Minimal stand-alone counterpart of the part of gmtl (imvu/gmtl, header-only) that the
// reference renderer's scene code uses: gmtl::Vec3d with element access, element-wise
// + - * /, dot, length, lengthSquared, normalize and makeNormal.
// Call sites in the reference: src/common.h:9-43, src/objects.h:20-146,
// src/main.cc:55-100,156-205, src/view_frustum.h:40-41.
//
// Semantics follow gmtl's VecOps.h: dot and lengthSquared are a running sum that starts
// from 0 and adds x, y, z products in that order; normalize divides each element by the
// length (three divisions, no reciprocal) and leaves a zero vector alone.
#pragma once

#include <cmath>

namespace gmtl {

class Vec3d {
public:
    Vec3d() : e_{0.0, 0.0, 0.0} {}
    Vec3d(double x, double y, double z) : e_{x, y, z} {}

    double& operator[](unsigned i) { return e_[i]; }
    const double& operator[](unsigned i) const { return e_[i]; }
    const double* getData() const { return e_; }

private:
    double e_[3];
};

inline Vec3d operator+(const Vec3d& a, const Vec3d& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline Vec3d operator-(const Vec3d& a, const Vec3d& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3d operator-(const Vec3d& a) { return {-a[0], -a[1], -a[2]}; }
inline Vec3d operator*(const Vec3d& a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }
inline Vec3d operator*(double s, const Vec3d& a) { return {a[0] * s, a[1] * s, a[2] * s}; }
inline Vec3d operator/(const Vec3d& a, double s) { return {a[0] / s, a[1] / s, a[2] / s}; }

inline double dot(const Vec3d& a, const Vec3d& b) {
    double acc = 0.0;
    for (unsigned i = 0; i < 3; ++i) acc += a[i] * b[i];
    return acc;
}
inline double lengthSquared(const Vec3d& a) { return dot(a, a); }
inline double length(const Vec3d& a) { return std::sqrt(lengthSquared(a)); }

inline double normalize(Vec3d& v) {
    const double len = length(v);
    if (len != 0.0) {
        for (unsigned i = 0; i < 3; ++i) v[i] /= len;
    }
    return len;
}
inline Vec3d makeNormal(const Vec3d& v) {
    Vec3d r(v);
    normalize(r);
    return r;
}

}  // namespace gmtl
