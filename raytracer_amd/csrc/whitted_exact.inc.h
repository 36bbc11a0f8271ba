// Bit-faithful evaluation of one pixel of the reference's Whitted trace.
//
// One lane per pixel, one wave per 8x8 pixel tile.  The reference's recursion
// (TraceRay, src/main.cc:19-113) becomes a per-lane depth-first walk over the ray tree
// with an explicit stack (registers, LDS, then global memory, see whitted_device.h): a hit
// node adds its own shaded colour, scaled by the
// product of the reflection/refraction coefficients on the path from the root, pushes
// its refraction child and goes on with its reflection child.  The combine of
// src/main.cc:107-112 is linear in the children's colours, so this gives the same sum
// with the additions in another order (colour only; every hit/miss/spawn decision is
// taken on exactly the doubles the reference computes).
//
// All geometry is IEEE double with no multiply-add contraction (this file must be built
// with -ffp-contract=off): add, mul, div and sqrt are correctly rounded on the device as
// on the CPU, so intersection parameters, hit points, normals and child rays are the
// same bits as in the reference, and so is every comparison made on them.
//
// Object loops run with a wave-uniform index: records are fetched with scalar loads and
// broadcast from SGPRs, the plane/sphere branch is uniform.  Every collision test is
// preceded by single-precision filters that can only answer "the reference's test certainly
// returns nullopt" (certain_miss, line_miss_from_light, box_miss below, each with its error
// bound); large scenes walk a tree of boxes instead of the whole list (find_nearest_tree,
// find_occluder_tree), with the container-order semantics restored from the indices.
#pragma once

#include "whitted_device.h"

// Everything below must be compiled without multiply-add contraction.
#pragma clang fp contract(off)

// Analysis builds only (tools/phase_budget.py): RT_PHASE_MARKERS puts a comment line into
// the assembly wherever a phase of the per-pixel walk begins, so that the static
// instructions of the kernel can be attributed to phases; RT_PHASE_PROFILE also makes the
// counting kernels count how often a wave enters each of eight phases (in place of the event
// counters).  Neither is defined in the library that ships.
#if defined(RT_PHASE_PROFILE) && !defined(RT_PHASE_MARKERS)
#define RT_PHASE_MARKERS 1
#endif
#ifdef RT_PHASE_MARKERS
#define RT_MARK(name) __asm__ volatile("; RT_PHASE " name ::: "memory")
#else
#define RT_MARK(name) ((void)0)
#endif
#ifdef RT_PHASE_PROFILE
#define RT_ENTER(counter)                                                                              \
    do {                                                                                               \
        const int me__ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      \
        if (__builtin_amdgcn_readfirstlane(me__) == me__) ++(counter);                                 \
    } while (0)
#else
#define RT_ENTER(counter) ((void)0)
#endif

namespace rt {
namespace exact {


// Scene records are only ever read; loading them through the constant address space
// lets the compiler use scalar loads whenever the index is wave-uniform.
#define RT_CONST __attribute__((address_space(4)))
typedef const RT_CONST double* cdata_t;
// The arguments of a render as they lie in the kernel-argument segment: every field is a
// scalar load, also when a launch carries the arguments of several renders and the wave
// picks one at run time.
typedef const RT_CONST KArgs CArgs;
// Code that reads a render's arguments is generic in their type AT: `const KArgs` (the
// kernel's by-value argument: the compiler loads it once and keeps it in scalar registers)
// where a launch has one render and no queue, CArgs where the wave picks a render at run time.

// Record `i` of a table of 64-byte records, read through the constant address space.
__device__ __forceinline__ DGeom load_geom(cdata_t t, int i) {
    const cdata_t r = t + (size_t)i * 8;
    DGeom g;
    g.p[0] = r[0]; g.p[1] = r[1]; g.p[2] = r[2];
    g.q[0] = r[3]; g.q[1] = r[4]; g.q[2] = r[5];
    g.r2 = r[6];
    g.transparency = ((const RT_CONST float*)(r + 7))[0];
    g.pad = 0.0f;
    return g;
}
struct LightRec {       // one DLight in scalar registers
    double pos[3];
    float ci[3];
};
__device__ __forceinline__ LightRec load_light(cdata_t t, int i) {
    const cdata_t r = t + (size_t)i * 8;
    const RT_CONST float* f = (const RT_CONST float*)(r + 3);
    LightRec l;
    l.pos[0] = r[0]; l.pos[1] = r[1]; l.pos[2] = r[2];
    l.ci[0] = f[0]; l.ci[1] = f[1]; l.ci[2] = f[2];
    return l;
}

struct V3 {
    double x, y, z;
};

__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }

// gmtl::dot: sum over x,y,z in that order.  (gmtl starts the sum from 0; 0 + x differs
// from x only in the sign of a zero result, which no later operation here can observe.)
__device__ __forceinline__ double dot(V3 a, V3 b) {
    double r = a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
__device__ __forceinline__ double len2(V3 a) { return dot(a, a); }

// ---- correctly rounded sqrt and division without the range scaling -------------------
// The compiler expands an IEEE double sqrt or division into an estimate (v_rsq / v_rcp),
// Newton steps and a final correction by the exact residual, wrapped in a rescaling
// (v_ldexp / v_div_scale / v_div_fixup) that only acts when an operand, the reciprocal or
// the result leaves the normal range.  The helpers below are the same sequences without
// the wrapping; the callers use them only where the wrapping would be the identity
// (operands and results between 2^-760 and 2^760), so they return the same correctly
// rounded bits, and fall back to the plain operators elsewhere.  What is gained: the
// three divisions of a normalisation share one refined reciprocal.
__device__ __forceinline__ double sqrt_mid(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y;
    const double h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, s0, 0.5);
    const double s1 = __builtin_fma(s0, r0, s0);
    const double h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-s1, s1, x);
    const double s2 = __builtin_fma(d0, h1, s1);
    const double d1 = __builtin_fma(-s2, s2, x);
    return __builtin_fma(d1, h1, s2);
}
__device__ __forceinline__ double rcp_refined(double s) {
    double y = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-s, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}
// n / s given y = rcp_refined(s), s > 0.  The copysign only matters for n = -0.
__device__ __forceinline__ double div_mid(double n, double s, double y) {
    const double q = n * y;
    const double r = __builtin_fma(-s, q, n);
    return __builtin_copysign(__builtin_fma(r, y, q), n);
}
// True when every active lane of the wave has `ok`: the guards below branch on that, so
// that the wave runs either the short sequences or the plain operators, never both.
__device__ __forceinline__ bool all_lanes(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0ull; }

// The range guards look at exponent fields with integer operations (a double compare costs
// as much as a double multiplication).  They are conservative: a value they turn away only
// sends the wave through the plain operators, which return the same bits.
// 2^lo <= v < 2^hi for a double v (false for negative values, NaN and infinity).
__device__ __forceinline__ bool exp_within(double v, int lo, int hi) {
    return (uint32_t)(__double2hiint(v) - ((1023 + lo) << 20)) < (uint32_t)((hi - lo) << 20);
}
// Every component at least 2^lo in magnitude.  (A zero component would be handled by the
// unscaled division too, but telling zero from a denormal costs more than the rare wave
// with a zero component loses on the plain operators.)
__device__ __forceinline__ bool mid_vec(V3 a, int lo) {
    const uint32_t ux = (uint32_t)__double2hiint(a.x) & 0x7fffffffu;
    const uint32_t uy = (uint32_t)__double2hiint(a.y) & 0x7fffffffu;
    const uint32_t uz = (uint32_t)__double2hiint(a.z) & 0x7fffffffu;
    const uint32_t m = ux < uy ? ux : uy;
    return (m < uz ? m : uz) >= (uint32_t)((1023 + lo) << 20);
}

// gmtl::normalize with the plain operators: length, then three divisions, skipped for a
// zero length.
__device__ __forceinline__ V3 unit_plain(V3 a) {
    const double len = __builtin_sqrt(len2(a));
    if (len != 0.0) {
        a.x = a.x / len;
        a.y = a.y / len;
        a.z = a.z / len;
    }
    return a;
}
// The same for 2^-760 <= l2 = len2(a) <= 2^760 and components that are zero or >= 2^-380
// (the guards only admit the latter):
// then len is in [2^-380, 2^380], |a_i| <= len (1 + 2^-52), every quotient is normal.
__device__ __forceinline__ V3 unit_mid(V3 a, double l2) {
    const double len = sqrt_mid(l2);
    const double y = rcp_refined(len);
    return mk(div_mid(a.x, len, y), div_mid(a.y, len, y), div_mid(a.z, len, y));
}
__device__ __forceinline__ V3 unit(V3 a) {
    const double l2 = len2(a);
    V3 r;
    if (all_lanes(exp_within(l2, -760, 760) & mid_vec(a, -380))) {
        RT_MARK("resume");
        r = unit_mid(a, l2);
    } else {
        RT_MARK("fallback");
        r = unit_plain(a);
    }
    RT_MARK("resume");
    return r;
}
// unit(unit(a)): a Ray's constructor normalises the direction it is given, which the
// caller has normalised already (src/common.h:24).  One guard covers both passes: with
// components zero or >= 2^-250 and l2 in [2^-500, 2^500] the first result has components
// zero or >= 2^-500 and a squared length within a few 2^-52 of 1.
__device__ __forceinline__ V3 unit2(V3 a) {
    const double l2 = len2(a);
    V3 r;
    if (all_lanes(exp_within(l2, -500, 500) & mid_vec(a, -250))) {
        RT_MARK("resume");
        const V3 u = unit_mid(a, l2);
        r = unit_mid(u, len2(u));
    } else {
        RT_MARK("fallback");
        r = unit_plain(unit_plain(a));
    }
    RT_MARK("resume");
    return r;
}

// a / s, component by component, for |a_i| <= 2 max(s, 1) (both callers guarantee it).
__device__ __forceinline__ V3 div3(V3 a, double s) {
    V3 r;
    if (all_lanes(exp_within(s, -380, 380) & mid_vec(a, -380))) {
        RT_MARK("resume");
        const double y = rcp_refined(s);
        r = mk(div_mid(a.x, s, y), div_mid(a.y, s, y), div_mid(a.z, s, y));
    } else {
        RT_MARK("fallback");
        r = mk(a.x / s, a.y / s, a.z / s);
    }
    RT_MARK("resume");
    return r;
}
// sqrt(x) for the lanes with x >= 0; lanes with `unused` set may hold anything (their result
// is not looked at) and do not keep the wave from the short sequence.
__device__ __forceinline__ double sqrt_exact(double x, bool unused = false) {
    double r;
    if (all_lanes(exp_within(x, -760, 760) | unused)) {
        RT_MARK("resume");
        r = sqrt_mid(x);
    } else {
        RT_MARK("fallback");
        r = __builtin_sqrt(x);
    }
    RT_MARK("resume");
    return r;
}
// 1.0 / sqrt(x), both operations correctly rounded (src/main.cc:88).
__device__ __forceinline__ double inv_sqrt_2r(double x) {
    double r;
    if (all_lanes(exp_within(x, -760, 760))) {
        RT_MARK("resume");
        const double sx = sqrt_mid(x);
        const double y = rcp_refined(sx);
        r = __builtin_fma(__builtin_fma(-sx, y, 1.0), y, y);
    } else {
        RT_MARK("fallback");
        r = 1.0 / __builtin_sqrt(x);
    }
    RT_MARK("resume");
    return r;
}

// ---- colour arithmetic ------------------------------------------------------------------
// Nothing computed from a colour ever decides a hit, a miss or a spawn, so the colour chain
// (shading terms, Fresnel weights, path weights, accumulation) runs in single precision with
// fused multiply-adds.  Against the reference's double chain a channel moves by a few 1e-7
// of its value, i.e. the 8-bit value differs (by one) only where the double colour times 255
// sits within ~1e-4 of an integer: a few pixels in ten thousand.

// src/color.h:77-79 with amount = 0.1f: b*amount + a*(1-amount) (the reference rounds the
// double expression to float; here it is evaluated in float).
__device__ __forceinline__ float mix_tenth(float a, float b) {
    return __builtin_fmaf(a, 1.0f - 0.1f, b * 0.1f);
}

// src/color.h:27-29: uint8_t(clamp(v,0,1)*255), truncating; a NaN channel is defined as 0
// (v_med3_f32 returns the smallest operand for a NaN input).
__device__ __forceinline__ uint32_t to_u8(float v) {
    const float c = __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
    return (uint32_t)(int)(c * 255.0f);
}


// ---- single-precision certain-miss filter --------------------------------------------
// Most collision tests of a ray are misses by a wide margin.  Those are recognised on the
// float image of the (exact, double) ray with a few float operations and a rigorous error
// bound; only the tests the filter cannot rule out are evaluated in double.  The filter
// never decides anything but "the reference's test certainly returns nullopt", so the
// results stay the reference's bit for bit.
//
// U = 2^-24 (rounded up).  Float image of the ray: |o~ - o| <= U|o|_1 =: eo, |d~ - d| <= 2U
// =: ed.  Float image of a primitive: |c~ - c| <= U|c|_1 =: cerr (stored in the record).
//   sphere (src/objects.h:124-131), L = c~ - o~:  |L - L*| <= eo + cerr + U|L|   =: eL
//     tca = L.d~:  |tca - tca*| <= eL + |L|(ed + 3U)                            =: et
//     P = L - tca d~ (perpendicular part): |P - P*| <= 2 et + 2U|L|             =: eP
//     the reference misses when tca* < 0 or d2* = |P*|^2 (up to 4*2^-53 |L|^2) > r2:
//     certain when tca < -et, or |P| > r + eP, which follows from d2 > 2 r2 + 2 eP^2;
//     the test uses 2.01 r2 + 3 eP^2, the surplus eP^2 > 1e-12 |L|^2 covering d2*'s own error.
//     et, eP are replaced by per-ray bounds etb, ePb using |L|_1 <= |o~|_1 + max|c|_1.
//   plane (src/objects.h:91-101), num = h~ - o~.n~ with h~ = float(p0.n):
//     |num - num*| <= eo + 5U|o|_1 + herr =: en,  |denom - denom*| <= ed + 4U =: eden;
//     the reference misses when t* = num*/denom* <= 0 (or |denom*| <= 1e-6): certain when
//     num and denom are certainly non-zero with opposite signs.
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float dotf(F3 a, F3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float l1f(F3 a) { return __builtin_fabsf(a.x) + __builtin_fabsf(a.y) + __builtin_fabsf(a.z); }

typedef const RT_CONST float* cfdata_t;
struct FRec {            // one FGeom record in SGPRs
    float a0, a1, a2, r2;  // centre | normal,  r2 | -1 (plane)
    float b0, b1;          // sphere: cerr, 0 ; plane: h, herr
};
__device__ __forceinline__ FRec load_rec(cfdata_t p) {
    FRec g;
    g.a0 = p[0]; g.a1 = p[1]; g.a2 = p[2]; g.r2 = p[3];
    g.b0 = p[4]; g.b1 = p[5];
    return g;
}

struct FilterRay {
    F3 o, d;
    float etb;    // bound on the error of any tca of this ray
    float far2;   // 3 * ePb^2
    float ecross; // error bound + clearance of the box test's cross products
    float ebehind;// error bound of the largest ray parameter over a box
    float en_ray; // eo + 5U|o|_1
    float eden;   // ed + 4U
    float eray;   // 1e-13 Sb^2: what the reference's own rounding can move a squared distance
};
// Everything but the direction depends on the origin only: the shadow feelers of one hit
// share it (filter_origin once per hit, set_dir once per light).
__device__ __forceinline__ void set_dir(FilterRay& r, V3 d) { r.d = f3((float)d.x, (float)d.y, (float)d.z); }
__device__ __forceinline__ FilterRay filter_origin(V3 o, float scene_p1max) {
    const float U = 6.0e-8f;
    FilterRay r;
    r.o = f3((float)o.x, (float)o.y, (float)o.z);
    r.d = f3(0.0f, 0.0f, 0.0f);
    const float o1 = l1f(r.o);
    const float eo = U * o1;
    const float ed3 = 5.0f * U;                      // ed + 3U
    const float Sb = o1 + scene_p1max;               // >= |c~ - o~|_1 for every sphere
    const float eLb = eo + U * scene_p1max + U * Sb;
    r.etb = __builtin_fmaf(Sb, ed3, eLb);
    const float ePb = __builtin_fmaf(2.0f * U, Sb, 2.0f * r.etb);
    r.far2 = 3.0f * ePb * ePb;
    r.ecross = 32.0f * U * Sb;
    r.ebehind = __builtin_fmaf(20.0f * U, Sb, r.etb);
    r.en_ray = 6.0f * eo;
    r.eden = 6.0f * U;
    r.eray = 1.0e-13f * Sb * Sb;
    return r;
}
__device__ __forceinline__ FilterRay filter_ray(V3 o, V3 d, float scene_p1max) {
    FilterRay r = filter_origin(o, scene_p1max);
    set_dir(r, d);
    return r;
}

// Light-anchored line test for shadow feelers.  A feeler's line passes through its light
// P (to within |P - ip| 4e-16, the rounding of the normalisation), so the distance of a
// sphere's centre c to that line can be taken from the light: with the per-(light, sphere)
// constant L' = c - P, D^2 = |L'|^2 - (L'.d)^2.  The host stores L' in float and
// thr = |L'|^2 - r^2 - 20 U |L'|_1^2 rounded down; the float evaluation of
// thr - (L'.d)^2 is off by at most 14 U |L'|_1^2 (L': U|L'|_1; d: 2U; dot: 3 roundings),
// so a value above eray proves D^2 > r^2 by more than the reference's own rounding of
// d2 = L.L - tca^2 (2e-15 |L|^2 <= eray): its test returns nullopt.  Four float operations
// and a compare per sphere instead of the 15 of the general filter; planes carry
// thr = -inf and always go on to the general filter.
struct LRec {
    float x, y, z, thr;
};
__device__ __forceinline__ LRec load_lrec(cfdata_t p) {
    LRec l;
    l.x = p[0]; l.y = p[1]; l.z = p[2]; l.thr = p[3];
    return l;
}
__device__ __forceinline__ bool line_miss_from_light(const LRec& l, const FilterRay& r) {
    const float t = __builtin_fmaf(l.z, r.d.z, __builtin_fmaf(l.y, r.d.y, l.x * r.d.x));
    return __builtin_fmaf(-t, t, l.thr) > r.eray;
}
// True when the reference's test of this ray against the primitive certainly returns
// nullopt.  ~15 float operations, no division, no square root.
__device__ __forceinline__ bool certain_miss_sphere(float c0, float c1, float c2, float r2, const FilterRay& r) {
    const F3 L = f3(c0, c1, c2) - r.o;
    const float tca = dotf(L, r.d);
    const F3 P = f3(__builtin_fmaf(-tca, r.d.x, L.x), __builtin_fmaf(-tca, r.d.y, L.y), __builtin_fmaf(-tca, r.d.z, L.z));
    const float d2 = dotf(P, P);
    return (tca < -r.etb) | (__builtin_fmaf(-2.01f, r2, d2) > r.far2);
}
__device__ __forceinline__ bool certain_miss(const FRec& g, const FilterRay& r) {
    if (__float_as_int(g.r2) < 0) {  // plane (wave-uniform)
        const F3 a = f3(g.a0, g.a1, g.a2);
        const float denom = dotf(r.d, a);
        const float num = g.b0 - dotf(r.o, a);
        const float en = r.en_ray + g.b1;
        // both certainly non-zero, and of opposite signs (no short-circuit: three compares)
        return (__builtin_fabsf(num) > en) & (__builtin_fabsf(denom) > r.eden) & (num * denom < 0.0f);
    }
    return certain_miss_sphere(g.a0, g.a1, g.a2, g.r2, r);
}


// Plane::TestCollision (src/objects.h:91-102) / Sphere::TestCollision
// (src/objects.h:124-138) of a broadcast primitive against this lane's ray.
// `g` is wave-uniform, so the type branch is.
template <bool SPHERE = false>  // SPHERE: the caller knows the primitive is a sphere (a leaf of the tree)
__device__ __forceinline__ bool test_collision(const DGeom& g, V3 o, V3 d, double& t) {
    // Written without early returns: the lanes that get here have passed the filter, so
    // most of them hit, and a NaN or infinity computed on the others is never used.  The
    // conditions are the reference's, negated the way its early returns negate them.
    if (!SPHERE && __double2hiint(g.r2) < 0) {
        const V3 n = mk(g.q[0], g.q[1], g.q[2]);
        const double denom = dot(d, n);
        const V3 diff = mk(g.p[0], g.p[1], g.p[2]) - o;
        const double tt = dot(diff, n) / denom;
        t = tt;
        return (__builtin_fabs(denom) > 1e-6) & (tt > 0);
    }
    const V3 L = mk(g.p[0], g.p[1], g.p[2]) - o;
    const double tca = dot(L, d);
    const double d2 = dot(L, L) - tca * tca;
    const bool ok = !(tca < 0) & !(d2 > g.r2);
    if (ok) {
        const double thc = sqrt_exact(g.r2 - d2);
        const double t0 = tca - thc;
        t = (t0 < 0) ? tca + thc : t0;
    }
    return ok;
}

// Whether TestCollision returns a value, without the value: what a shadow feeler asks of
// every object but its first occluder (src/objects.h:38-40).  A sphere needs no square
// root for that (src/objects.h:126-130 return before it), a plane needs its division.
// On a hit (pa, pb) keep what the value follows from: a plane's t and -1, or a sphere's
// tca and r2 - d2 >= 0 (collision_value() finishes the arithmetic of src/objects.h:131-137).
template <bool SPHERE = false>
__device__ __forceinline__ bool collides(const DGeom& g, V3 o, V3 d, double& pa, double& pb) {
    if (!SPHERE && __double2hiint(g.r2) < 0) {
        double t = 0.0;
        const bool hit = test_collision(g, o, d, t);
        pa = hit ? t : pa;
        pb = hit ? -1.0 : pb;
        return hit;
    }
    const V3 L = mk(g.p[0], g.p[1], g.p[2]) - o;
    const double tca = dot(L, d);
    const double d2 = dot(L, L) - tca * tca;
    const bool hit = !(tca < 0) & !(d2 > g.r2);
    pa = hit ? tca : pa;
    pb = hit ? g.r2 - d2 : pb;
    return hit;
}
__device__ __forceinline__ double collision_value(double pa, double pb) {
    const double thc = sqrt_exact(pb, __double2hiint(pb) < 0);  // a plane's -1: not used
    const double t0 = pa - thc;
    const double ts = (t0 < 0) ? pa + thc : t0;
    return (pb < 0) ? pa : ts;
}
// Transmission factor of an occluder (src/objects.h:40-46), colour arithmetic: `m` = its
// normal before normalisation (a plane's stored normal, or hit point - centre), fd = the
// feeler's direction.
__device__ __forceinline__ float transmission(V3 m, V3 fd, float transparency) {
    const float mx = (float)m.x, my = (float)m.y, mz = (float)m.z;
    const float md = __builtin_fmaf(mz, (float)fd.z, __builtin_fmaf(my, (float)fd.y, mx * (float)fd.x));
    const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx)));
    const float facing = -(md * inv);
    const float fres = mix_tenth(1.0f - facing, 0.0f);
    return (1.0f - fres) * transparency;
}
// The same for the occluder `og` the tree walk found for the feeler (ip, fdir).
__device__ __forceinline__ float occluder_factor(const DGeom* og, V3 ip, V3 fdir, double pa, double pb) {
    const double p = collision_value(pa, pb);
    const V3 hp = ip + fdir * p;
    const V3 m = (pb < 0) ? mk(og->q[0], og->q[1], og->q[2]) : (hp - mk(og->p[0], og->p[1], og->p[2]));
    return transmission(m, fdir, og->transparency);
}

// ---- sphere tree (large scenes) -------------------------------------------------------
// Interior nodes of the tree built by sphere_tree.h are boxes (float centre C, half extents
// h, rounded outwards on the host so that the box contains every member sphere).
// box_miss() returns the mask of the lanes for whose ray the reference's test certainly
// returns nullopt for EVERY sphere below the node.  With m = C - o, Sb >= |o|_1 + |C|_1 + |h|_1, U = 2^-24:
//   (A) the ray's line passes the box at a distance: for one of the axes n = d x e_k,
//       |m . n| - sum_j h_j |n_j| > ecross.  In float the left side is off by at most
//       24 U Sb (m: 2 U Sb; d: 2U; five roundings); ecross = 32 U Sb leaves a clearance of
//       8 U Sb ~ 5e-7 Sb between line and box, ten times the 4.5e-8 |L| by which the
//       reference's rounded d2 = L.L - tca^2 can misjudge the distance to a sphere; or
//   (B) the box is behind the origin: the largest ray parameter over the box,
//       m . d + sum_j h_j |d_j|, is below -ebehind (its float error is at most
//       etb + 15 U Sb): every point of every member has a negative parameter.
struct FNode {
    float c0, c1, c2;   // centre (box or sphere)
    float h0, h1, h2;   // half extents; leaf: h0 = radius^2
    int skip, tag;      // interior: next node when missed, smallest index below; leaf: tag = ~index
};
__device__ __forceinline__ FNode load_node(cfdata_t p) {
    FNode n;
    n.c0 = p[0]; n.c1 = p[1]; n.c2 = p[2];
    n.h0 = p[3]; n.h1 = p[4]; n.h2 = p[5];
    n.skip = __float_as_int(p[6]);
    n.tag = __float_as_int(p[7]);
    // all eight words are wanted here: keeps them one load (parts of the record moved into the
    // branch that uses them would be fetched by a second, dependent load)
    __asm__ volatile("" ::"s"(n.c0), "s"(n.c1), "s"(n.c2), "s"(n.h0), "s"(n.h1), "s"(n.h2), "s"(n.skip), "s"(n.tag));
    return n;
}
// (C) nearest-hit walks only: the box lies beyond what the lane has found already - its
//     smallest ray parameter, m . d - sum_j h_j |d_j| (same error bound as the largest),
//     exceeds `reach`, an upper bound of the lane's `nearest` plus that error (near_reach()):
//     every hit in the box has a larger t than the hit in hand (the reference's own rounding
//     of t, at most 4e-8 |L| at a grazing hit, is within the bound's surplus of 5 U Sb).
//     Pass reach = +inf to leave the condition out.
__device__ __forceinline__ unsigned long long box_miss(const FNode& n, const FilterRay& r, float reach) {
    const F3 m = f3(n.c0, n.c1, n.c2) - r.o;
    const F3 d = r.d;
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    // separating axes d x e_x, d x e_y, d x e_z: |m . axis| - sum_j h_j |axis_j|, four
    // operations each (the absolute values are operand modifiers)
    const float cx = __builtin_fabsf(__builtin_fmaf(m.y, d.z, -(m.z * d.y)));
    const float cy = __builtin_fabsf(__builtin_fmaf(m.z, d.x, -(m.x * d.z)));
    const float cz = __builtin_fabsf(__builtin_fmaf(m.x, d.y, -(m.y * d.x)));
    const float sx = __builtin_fmaf(-n.h2, ay, __builtin_fmaf(-n.h1, az, cx));
    const float sy = __builtin_fmaf(-n.h0, az, __builtin_fmaf(-n.h2, ax, cy));
    const float sz = __builtin_fmaf(-n.h1, ax, __builtin_fmaf(-n.h0, ay, cz));
    // largest and smallest ray parameter over the box
    float md = m.x * d.x;
    md = __builtin_fmaf(m.y, d.y, md);
    md = __builtin_fmaf(m.z, d.z, md);
    float ext = n.h0 * ax;
    ext = __builtin_fmaf(n.h1, ay, ext);
    ext = __builtin_fmaf(n.h2, az, ext);
    // one wave mask per condition: the lanes whose ray certainly misses the box
    return __builtin_amdgcn_ballot_w64(__builtin_fmaxf(__builtin_fmaxf(sx, sy), sz) > r.ecross) |
           __builtin_amdgcn_ballot_w64(md + ext < -r.ebehind) | __builtin_amdgcn_ballot_w64(md - ext > reach);
}
// Upper bound (float) of the lane's nearest hit so far, plus the error bound of a box's
// smallest ray parameter.
__device__ __forceinline__ float near_reach(double nearest, const FilterRay& r) {
    return __builtin_fmaf((float)nearest, 1.0000002f, r.ebehind);
}
// The leaves of a group node: four records from `p` on (the table has spare records at its end).
struct FLeaves {
    FNode l[4];
};
__device__ __forceinline__ FLeaves load_leaves(cfdata_t p) {
    FLeaves g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        g.l[j].c0 = p[8 * j]; g.l[j].c1 = p[8 * j + 1]; g.l[j].c2 = p[8 * j + 2]; g.l[j].h0 = p[8 * j + 3];
        g.l[j].h1 = 0.0f; g.l[j].h2 = 0.0f; g.l[j].skip = 0;
        g.l[j].tag = __float_as_int(p[8 * j + 7]);
    }
    return g;
}

// Nearest hit over all objects (src/main.cc:37-47) through the tree.  Objects are met in
// another order than the container's, so the reference's strict `t < nearest` in
// container order becomes: smaller t wins, equal t goes to the smaller index.
template <bool SPHERE>
__device__ __forceinline__ void near_exact_candidate(cdata_t geom, int idx, V3 o, V3 d, double& nearest, int& cur) {
    __asm__ volatile("" ::: "memory");
    const DGeom g = load_geom(geom, idx);
    double t;
    const bool hit = test_collision<SPHERE>(g, o, d, t);
    const bool closer = hit && (t < nearest || (t == nearest && idx < cur));
    nearest = closer ? t : nearest;
    cur = closer ? idx : cur;
}
template <bool COUNT>
__device__ __forceinline__ void near_candidate(const FRec& gf, cdata_t geom, int idx, const FilterRay& fr, V3 o, V3 d,
                                               double& nearest, int& cur) {
    if (!certain_miss(gf, fr)) near_exact_candidate<false>(geom, idx, o, d, nearest, cur);
}
template <bool COUNT>  // a leaf of the tree: always a sphere
__device__ __forceinline__ void near_candidate(const FNode& nd, cdata_t geom, int idx, const FilterRay& fr, V3 o, V3 d,
                                               double& nearest, int& cur) {
    if (!certain_miss_sphere(nd.c0, nd.c1, nd.c2, nd.h0, fr)) near_exact_candidate<true>(geom, idx, o, d, nearest, cur);
}
// The planes of a scene whose spheres are searched through the tree or a grid: a short list
// in front of that search.
template <bool COUNT, typename AT>
__device__ __forceinline__ void near_planes(AT& a, cdata_t geom, const FilterRay& fr, V3 o, V3 d, double& nearest, int& cur) {
    cfdata_t pp = (cfdata_t)(const float*)a.planes;
    for (int k = 0; k < a.n_planes; ++k, pp += 8) {
        const FRec gf = load_rec(pp);
        near_candidate<COUNT>(gf, geom, __float_as_int(pp[6]), fr, o, d, nearest, cur);
    }
}
template <bool COUNT, typename AT>
__device__ __forceinline__ void find_nearest_tree(AT& a, cdata_t geom, const FilterRay& fr, V3 o, V3 d, double& nearest,
                                                  int& cur, uint32_t& c_sph) {
    near_planes<COUNT>(a, geom, fr, o, d, nearest, cur);
    // The position in the tree is kept as a byte offset (one scalar add per step).
    const RT_CONST char* tree = (const RT_CONST char*)(const char*)a.tree;
    const uint32_t end = (uint32_t)a.n_nodes << 5;
    uint32_t off = 0;
    float reach = near_reach(nearest, fr);
    while (off < end) {
        const FNode nd = load_node((cfdata_t)(tree + off));
        RT_ENTER(c_sph);  // (analysis builds: nodes visited)
        off += 32u;
        const unsigned long long walking = __builtin_amdgcn_ballot_w64(true);
        const bool enter = (walking & ~box_miss(nd, fr, reach)) != 0ull;
        if (nd.skip < 0) {  // a group: its leaves follow, each a sphere
            const uint32_t cnt = (uint32_t)(-nd.skip);
            if (enter) {
                const FLeaves lv = load_leaves((cfdata_t)(tree + off));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((uint32_t)j < cnt) near_candidate<COUNT>(lv.l[j], geom, ~lv.l[j].tag, fr, o, d, nearest, cur);
                reach = near_reach(nearest, fr);
            }
            off += cnt << 5;
        } else {
            off = enter ? off : (uint32_t)nd.skip << 5;
        }
    }
}

// Shadow feeler through the tree (src/objects.h:31-47): the reference stops at the first
// occluder in container order, i.e. the occluder with the smallest index; `best` tracks it
// and prunes every node that only holds larger indices.  Returns the index or INT_MAX.
// `ruled_out` = the filter's answer; the rest is asked only of the lanes it leaves.
template <bool SPHERE>
__device__ __forceinline__ void feeler_exact_candidate(bool ruled_out, cdata_t geom, int idx, V3 ip, V3 fdir, int cur,
                                                       int& best, double& pa, double& pb) {
    if (!ruled_out & (idx != cur) & (idx < best)) {
        __asm__ volatile("" ::: "memory");
        const DGeom g = load_geom(geom, idx);
        double na = pa, nb = pb;
        const bool hit = collides<SPHERE>(g, ip, fdir, na, nb);
        pa = na; pb = nb;
        best = hit ? idx : best;
    }
}
__device__ __forceinline__ void feeler_candidate(const FRec& gf, cdata_t geom, int idx, const FilterRay& fr, V3 ip, V3 fdir,
                                                 int cur, int& best, double& pa, double& pb) {
    feeler_exact_candidate<false>(certain_miss(gf, fr), geom, idx, ip, fdir, cur, best, pa, pb);
}
__device__ __forceinline__ void feeler_candidate(const FNode& nd, cdata_t geom, int idx, const FilterRay& fr, V3 ip, V3 fdir,
                                                 int cur, int& best, double& pa, double& pb) {
    feeler_exact_candidate<true>(certain_miss_sphere(nd.c0, nd.c1, nd.c2, nd.h0, fr), geom, idx, ip, fdir, cur, best, pa, pb);
}
template <typename AT>
__device__ __forceinline__ void feeler_planes(AT& a, cdata_t geom, const FilterRay& fr, V3 ip, V3 fdir, int cur, int& best,
                                              double& pa, double& pb) {
    cfdata_t pp = (cfdata_t)(const float*)a.planes;
    for (int k = 0; k < a.n_planes; ++k, pp += 8) {
        const FRec gf = load_rec(pp);
        feeler_candidate(gf, geom, __float_as_int(pp[6]), fr, ip, fdir, cur, best, pa, pb);
    }
}
template <typename AT>
__device__ __forceinline__ int find_occluder_tree(AT& a, cdata_t geom, const FilterRay& fr, V3 ip, V3 fdir, int cur,
                                                  double& pa, double& pb, uint32_t& c_nodes) {
    int best = 0x7fffffff;
    feeler_planes(a, geom, fr, ip, fdir, cur, best, pa, pb);
    const RT_CONST char* tree = (const RT_CONST char*)(const char*)a.tree;
    const uint32_t end = (uint32_t)a.n_nodes << 5;
    uint32_t off = 0;
    while (off < end) {
        const FNode nd = load_node((cfdata_t)(tree + off));
        RT_ENTER(c_nodes);  // (analysis builds: nodes visited)
        off += 32u;
        // entered when some lane still looks for an index below the node's smallest and
        // its ray may touch the box
        const unsigned long long open = __builtin_amdgcn_ballot_w64(nd.tag < best);
        const bool enter = (open & ~box_miss(nd, fr, __builtin_inff())) != 0ull;
        if (nd.skip < 0) {  // a group: its leaves follow
            const uint32_t cnt = (uint32_t)(-nd.skip);
            if (enter) {
                const FLeaves lv = load_leaves((cfdata_t)(tree + off));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((uint32_t)j < cnt) feeler_candidate(lv.l[j], geom, ~lv.l[j].tag, fr, ip, fdir, cur, best, pa, pb);
            }
            off += cnt << 5;
        } else {
            off = enter ? off : (uint32_t)nd.skip << 5;
        }
    }
    return best;
}

// ---- anchor grids (anchor_grid.h) -------------------------------------------------------
// The list of the cell that the line with (float) direction d through anchor `anchor` falls
// into; `ok` is false for a direction that is not of unit length (the caller then takes the
// general path).  The cell is found with float arithmetic whose error (below 1e-6 in the
// cell coordinates) the builder has allowed for.
struct GridList {
    const GridEntry* e;
    uint32_t n;
};
template <typename AT>
__device__ __forceinline__ GridList grid_list(AT& a, int anchor, F3 d, bool& ok) {
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    const bool fx = (ax >= ay) & (ax >= az);
    const bool fy = !fx & (ay >= az);
    // face 0: k = x, (i, j) = (y, z); face 1: k = y, (x, z); face 2: k = z, (x, y)
    const float dk = fx ? d.x : (fy ? d.y : d.z);
    const float di = fx ? d.y : d.x;
    const float dj = (fx | fy) ? d.z : d.y;
    ok = __builtin_fabsf(dk) > 0.5f;  // a unit vector's largest component is at least 0.577
    const float inv = __builtin_amdgcn_rcpf(dk);
    const float half = a.grid_half;
    const float top = half + half - 1.0f;
    const float fu = __builtin_amdgcn_fmed3f(__builtin_fmaf(di * inv, half, half), 0.0f, top);
    const float fv = __builtin_amdgcn_fmed3f(__builtin_fmaf(dj * inv, half, half), 0.0f, top);
    const int face = fx ? 0 : (fy ? 1 : 2);
    const int R = a.grid_res;
    const uint32_t ci = (uint32_t)(((anchor * 3 + face) * R + (int)fv) * R + (int)fu);
    const GridCell c = a.grid_cells[ok ? ci : 0u];
    GridList gl;
    gl.e = a.grid_entries + c.start;
    gl.n = ok ? c.count : 0u;
    return gl;
}
__device__ __forceinline__ LRec entry_lrec(const GridEntry* e) {
    LRec l;
    l.x = e->l[0]; l.y = e->l[1]; l.z = e->l[2]; l.thr = e->thr;
    return l;
}
// Shadow feeler over the spheres of its cell (src/objects.h:31-47).  The list ascends in the
// container index, so the first sphere of it that the feeler hits is the first occluder among
// the spheres; `best` comes in as the index of the first occluding plane (or INT_MAX) and
// ends the search as soon as the list passes it.  Every lane walks its own list.
template <typename AT>
__device__ __forceinline__ void find_occluder_grid(AT& a, GridList gl, const FilterRay& fr, V3 ip, V3 fdir, int cur,
                                                   int& best, double& pa, double& pb) {
    const GridEntry* e = gl.e;
    for (uint32_t n = gl.n; n != 0u; --n, ++e) {
        const int idx = e->index;
        if (idx >= best) break;
        if ((idx == cur) | line_miss_from_light(entry_lrec(e), fr)) continue;
        double na = pa, nb = pb;
        const bool hit = collides<true>(a.geom[idx], ip, fdir, na, nb);
        pa = na; pb = nb;
        if (hit) { best = idx; break; }
    }
}
// Nearest hit of a primary ray over the spheres of its cell of the camera's grid
// (src/main.cc:37-47; the planes have been tested by the caller).
template <typename AT>
__device__ __forceinline__ void find_nearest_grid(AT& a, GridList gl, const FilterRay& fr, V3 o, V3 d, double& nearest,
                                                  int& cur) {
    const GridEntry* e = gl.e;
    for (uint32_t n = gl.n; n != 0u; --n, ++e) {
        if (line_miss_from_light(entry_lrec(e), fr)) continue;
        const int idx = e->index;
        double t;
        const bool hit = test_collision<true>(a.geom[idx], o, d, t);
        const bool closer = hit && (t < nearest || (t == nearest && idx < cur));
        nearest = closer ? t : nearest;
        cur = closer ? idx : cur;
    }
}

constexpr int kGroup = 4;  // the object loops take the scene four records at a time

// ---- certified single-precision shadow feelers -------------------------------------------
// A shadow feeler only asks WHETHER each object is hit, in container order, until the first one
// is (src/objects.h:35-46); the hit parameter of that first occluder only feeds a colour.  So a
// feeler can be answered in single precision wherever the answer is certain: a float test that
// proves "the reference's TestCollision returns nullopt" (certain miss) or "returns a value"
// (certain hit).  Only a lane that meets an object for which neither can be proved (a line
// that grazes a silhouette within the float error, an origin on a surface) stops and is sent
// through the double-precision search; its colour then comes from there.  Which path a lane
// takes depends on its own ray only, so a pixel does not depend on its neighbours.
//
// The direction is normalised in float: d~ = v~ rsq(v~.v~), v~ = float(P - ip) (the
// subtraction in double).  With U = 2^-24: |d~ - d| <= 6U (v~: U, the squared length: 3U,
// v_rsq_f32: 1 ulp, the products: U), the length is within 5U of 1, dist~ = (v~.v~) rsq is
// within 8U |P - ip| of the distance to the light.
//   sphere, L = c - P from the host (float, error U):  t = L~.d~ is within 10.1 U |L|_1 of L.d.
//     line test (line_miss_from_light): thr - t^2 > eray proves the miss (thr holds 24 U |L|_1^2).
//     Otherwise P~ = L~ - t d~ is the perpendicular from the line to the centre, x = r^2 - |P~|^2
//     is within `ehit` (host, see line_test_record()) of the true r^2 - D^2:
//       x < -(ehit + 32 eray)         certain miss   (eray: the reference's own rounding of d2)
//       tca~ = t + dist~ < -efront    certain miss   (tca~ is within efront = 16 U (Sb + dist~) of
//                                                     the reference's tca; it returns for tca < 0)
//       x > khit + 32 eray and tca~ > efront   certain hit; khit = ehit for an opaque sphere (its
//         factor is zero whatever the hit point), 64 ehit for a transparent one, which keeps the
//         error of thc = sqrt(x) below sqrt(ehit) / 16.  Its factor needs the normal at the hit
//         point: -dot(normal, d) = +thc / r at the near root, -thc / r at the far root, and the
//         reference takes the far root iff tca < thc (src/objects.h:132-136), i.e.
//         w = tca^2 - x < 0; |w~ - w| <= efront (2 tca~ + efront) + ehit, inside that band the lane
//         is not certain either.
//   plane (unit normal n, h = p0.n): denom~ = d~.n~ within eden = 12U, num~ = h~ - o~.n~ within
//     en = 6U |o|_1 + herr.  With |num~| > en: |denom~| > eden and opposite signs is a certain
//     miss (t < 0); |denom~| > 1e-6 (1 + 1e-5) + eden and equal signs a certain hit
//     (src/objects.h:93-100); its factor is -denom~ (the normal does not depend on the hit point).
struct FeelRay {
    F3 o;          // float image of the hit point
    F3 d;          // d~
    float dist;    // dist~
    float en_ray;  // 6U |o|_1
    float eray;    // 1e-13 Sb^2
    float efront;  // 16 U (Sb + dist~)
    bool valid;    // the squared length is in the range in which the float normalisation keeps
                   // its error bound (false also for a NaN): otherwise the lane is not certain
};
__device__ __forceinline__ FeelRay feel_ray(V3 ip, V3 to_light, float scene_p1max) {
    const float U = 6.0e-8f;
    FeelRay r;
    r.o = f3((float)ip.x, (float)ip.y, (float)ip.z);
    const F3 v = f3((float)to_light.x, (float)to_light.y, (float)to_light.z);
    const float s2 = dotf(v, v);
    const float inv = __builtin_amdgcn_rsqf(s2);
    r.d = f3(v.x * inv, v.y * inv, v.z * inv);
    r.dist = s2 * inv;
    r.valid = (s2 > 1.0e-30f) & (s2 < 1.0e30f);
    const float o1 = l1f(r.o);
    const float Sb = o1 + scene_p1max;
    r.en_ray = 6.0f * U * o1;
    r.eray = 1.0e-13f * Sb * Sb;
    r.efront = 16.0f * U * (Sb + r.dist);
    return r;
}
// (src/objects.h:42-44) for -dot(normal, feeler.dir) = facing.
__device__ __forceinline__ float transmission_facing(float facing, float transparency) {
    const float fres = mix_tenth(1.0f - facing, 0.0f);
    return (1.0f - fres) * transparency;
}
// A sphere the line test has not ruled out.  Returns true when the answer is certain; then
// `hit` tells which, and for a hit `factor` receives the occluder's transmission factor.
__device__ __forceinline__ bool classify_sphere(const LRec& l, float t, float r2, float ehit, float transparency,
                                                const FeelRay& r, bool& hit, float& factor) {
    const F3 P = f3(__builtin_fmaf(-t, r.d.x, l.x), __builtin_fmaf(-t, r.d.y, l.y), __builtin_fmaf(-t, r.d.z, l.z));
    const float x = r2 - dotf(P, P);
    const float tca = t + r.dist;
    // (32 eray: the reference's rounding, and what a line admitted by the line test with a
    // large eray - a far origin - adds to the error of x beyond ehit)
    const float e0 = __builtin_fmaf(32.0f, r.eray, ehit);
    const bool opaque = transparency == 0.0f;
    const float khit = opaque ? e0 : __builtin_fmaf(63.0f, ehit, e0);
    const bool miss = (x < -e0) | (tca < -r.efront);
    bool sure_hit = (x > khit) & (tca > r.efront);
    // which root the reference takes (only a transparent occluder's colour depends on it)
    const float w = __builtin_fmaf(tca, tca, -x);
    const float band = __builtin_fmaf(r.efront, tca + tca + r.efront, e0);
    sure_hit = sure_hit & (opaque | (__builtin_fabsf(w) > band));
    hit = sure_hit;
    if (sure_hit & !opaque) {
        const float thc_over_r = __builtin_sqrtf(x) * __builtin_amdgcn_rsqf(r2);
        factor = transmission_facing((w < 0.0f) ? -thc_over_r : thc_over_r, transparency);
    } else if (sure_hit) {
        factor = 0.0f;
    }
    return miss | sure_hit;
}
// A plane (record as in certain_miss()).  Same return convention.
__device__ __forceinline__ bool classify_plane(const FRec& g, float transparency, const FeelRay& r, bool& hit, float& factor) {
    const float U = 6.0e-8f;
    const F3 n = f3(g.a0, g.a1, g.a2);
    const float denom = dotf(r.d, n);
    const float num = g.b0 - dotf(r.o, n);
    const float en = r.en_ray + g.b1;
    const float eden = 12.0f * U;
    const bool nz = __builtin_fabsf(num) > en;
    const float sgn = num * denom;
    const bool miss = nz & (__builtin_fabsf(denom) > eden) & (sgn < 0.0f);
    const bool sure_hit = nz & (__builtin_fabsf(denom) > 1.00001e-6f + eden) & (sgn > 0.0f);
    hit = sure_hit;
    if (sure_hit) factor = transmission_facing(-denom, transparency);
    return miss | sure_hit;
}
// The float search through the light's anchor grid: planes (a short list), then the spheres of
// the lane's own cell, in ascending index.  Returns the first occluder (-1: none) for the lanes
// that stay certain; `unsure` is set for the others.
template <typename AT>
__device__ __forceinline__ int feeler_float_grid(AT& a, GridList gl, const FeelRay& fr, int cur, float& occl, bool& unsure) {
    int best = 0x7fffffff;
    unsure = false;
    cfdata_t pp = (cfdata_t)(const float*)a.planes;
    for (int k = 0; k < a.n_planes; ++k, pp += 8) {
        const int idx = __float_as_int(pp[6]);
        if ((idx != cur) & (idx < best) & !unsure) {
            bool hit;
            float factor = 0.0f;
            const bool sure = classify_plane(load_rec(pp), pp[7], fr, hit, factor);
            unsure = !sure;
            if (sure & hit) { best = idx; occl = factor; }
        }
    }
    if (!unsure) {
        const GridEntry* e = gl.e;
        for (uint32_t n = gl.n; n != 0u; --n, ++e) {
            const int idx = e->index;
            if (idx >= best) break;
            const LRec l = entry_lrec(e);
            const float t = __builtin_fmaf(l.z, fr.d.z, __builtin_fmaf(l.y, fr.d.y, l.x * fr.d.x));
            if ((idx == cur) | (__builtin_fmaf(-t, t, l.thr) > fr.eray)) continue;
            bool hit;
            float factor = 0.0f;
            const bool sure = classify_sphere(l, t, e->r2, e->ehit, e->transparency, fr, hit, factor);
            if (!sure) { unsure = true; break; }
            if (hit) { best = idx; occl = factor; break; }
        }
    }
    return (best != 0x7fffffff) ? best : -1;
}

// The object loops take the scene four records at a time: the float records of a group
// are fetched up front (scalar loads at constant offsets from one pointer), each slot
// costs a handful of scalar instructions, and the double record of a primitive is only
// fetched when some lane's filter cannot rule it out.  `rem` < 4 limits a trailing group.
struct FGroup {
    FRec r[kGroup];
};
__device__ __forceinline__ FGroup load_group(cfdata_t fp) {
    FGroup g;
#pragma unroll
    for (int j = 0; j < kGroup; ++j) g.r[j] = load_rec(fp + 8 * j);  // the table is padded to whole groups
    return g;
}

// Slot of the nearest-hit loop (src/main.cc:37-47).
template <bool COUNT>
__device__ __forceinline__ void near_slot(const FRec& gf, cdata_t gp, int j, int idx, bool valid, const FilterRay& fr,
                                          V3 o, V3 d, double& nearest, int& cur, uint32_t& c_sph, uint32_t& c_pln) {
#ifndef RT_PHASE_PROFILE
    if (COUNT && valid) { if (__float_as_int(gf.r2) < 0) ++c_pln; else ++c_sph; }
#endif
    const bool maybe = valid && !certain_miss(gf, fr);
    if (maybe) {
        __asm__ volatile("" ::: "memory");  // nothing of the double test is to run ahead of the branch
        RT_MARK("near_exact");
        RT_ENTER(c_sph);
        const DGeom g = load_geom(gp, j);
        double t;
        const bool closer = test_collision(g, o, d, t) && t < nearest;
        nearest = closer ? t : nearest;
        cur = closer ? idx : cur;
    }
    RT_MARK("near_filter");
}

// Slot of a shadow feeler's loop (src/objects.h:31-47): the first occluder in container
// order decides, the shaded object itself is skipped.  `first` = its index (-1 while none),
// `factor` = its transmission factor.
template <bool COUNT, bool LATE>
__device__ __forceinline__ void feeler_slot(const LRec& lr, cfdata_t fp, cdata_t gp, int j, int idx, bool valid,
                                            const FilterRay& fr, V3 ip, V3 fdir, int cur, int& first, float& factor,
                                            uint32_t& c_sph, uint32_t& c_pln) {
    // Whether the reference makes this test at all (no occluder met yet, not the shaded
    // object) can be asked before the line test or, LATE, only by the lanes the line test
    // leaves (for the others nothing follows from the answer).  LATE is cheaper per object
    // but sends every lane through the branch at its own object: it pays with many lights.
    const bool tested = valid && first < 0 && idx != cur;
#ifndef RT_PHASE_PROFILE
    if (COUNT && tested) { if (lr.thr == -__builtin_inff()) ++c_pln; else ++c_sph; }
#endif
    if ((LATE ? valid : tested) && !line_miss_from_light(lr, fr)) {
        __asm__ volatile("" ::: "memory");  // the rest only for the lanes the line test leaves
        RT_MARK("feeler_exact");
        RT_ENTER(c_sph);
        const FRec gf = load_rec(fp + 8 * j);
        const DGeom g = load_geom(gp, j);
        double p;
        if ((!LATE || (first < 0 && idx != cur)) && !certain_miss(gf, fr) && test_collision(g, ip, fdir, p)) {
            // src/objects.h:40-46, here with the object's record still in scalar registers
            const V3 hp = ip + fdir * p;
            const V3 m = (__double2hiint(g.r2) < 0) ? mk(g.q[0], g.q[1], g.q[2]) : (hp - mk(g.p[0], g.p[1], g.p[2]));
            factor = transmission(m, fdir, g.transparency);
            first = idx;
        }
    }
    RT_MARK("feeler_loop");
}

template <int MW>
__device__ __forceinline__ void set_bit(uint32_t (&m)[MW > 0 ? MW : 1], uint32_t node) {
#pragma unroll
    for (int w = 0; w < MW; ++w) m[w] |= ((node >> 5) == (uint32_t)w) ? (1u << (node & 31)) : 0u;
}

__device__ __forceinline__ unsigned long long wave_sum(uint32_t v) {
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// The nearest hit of a ray over all objects (src/main.cc:37-47): `cur` = its index (-1 =
// none), `nearest` = its t.
// GRID: a wave whose rays all start at the camera (`primary`; the origin is then exactly
// zero) searches the spheres through the camera's anchor grid.
template <bool COUNT, bool TREE, bool GRID, typename AT>
__device__ __forceinline__ void find_nearest(AT& a, cdata_t geom, cfdata_t fgeom, V3 o, V3 d, bool primary,
                                             double& nearest, int& cur, uint32_t& c_sph, uint32_t& c_pln) {
    const FilterRay fr = filter_ray(o, d, a.scene_p1max);
    bool by_grid = false;
    GridList gl;
    if (GRID) {
        if (all_lanes(primary)) {
            bool ok;
            gl = grid_list(a, a.n_lights, fr.d, ok);
            by_grid = all_lanes(ok & (fr.eray <= a.grid_eray_max));
        }
    }
#ifndef RT_PHASE_PROFILE
    if (COUNT && (TREE || by_grid)) { c_sph += (uint32_t)(a.n_prims - a.n_planes); c_pln += (uint32_t)a.n_planes; }
#endif
    if (GRID && by_grid) {
        near_planes<COUNT>(a, geom, fr, o, d, nearest, cur);
        find_nearest_grid(a, gl, fr, o, d, nearest, cur);
    } else if (TREE) {
        find_nearest_tree<COUNT>(a, geom, fr, o, d, nearest, cur, c_sph);
    } else {
        cfdata_t fp = fgeom;
        cdata_t gp = geom;
        int base = 0;
        for (; base + kGroup <= a.n_prims; base += kGroup, fp += 8 * kGroup, gp += 8 * kGroup) {
            const FGroup G = load_group(fp);
#pragma unroll
            for (int j = 0; j < kGroup; ++j)
                near_slot<COUNT>(G.r[j], gp, j, base + j, true, fr, o, d, nearest, cur, c_sph, c_pln);
        }
        if (base < a.n_prims) {
            const FGroup G = load_group(fp);
            const int rem = a.n_prims - base;
#pragma unroll
            for (int j = 0; j < kGroup - 1; ++j)
                near_slot<COUNT>(G.r[j], gp, j, base + j, j < rem, fr, o, d, nearest, cur, c_sph, c_pln);
        }
    }
}

// One shadow feeler (src/objects.h:31-47): the index of the first occluder in container
// order between `ip` and light `li` (never `cur`, the shaded object), or -1; `occl`
// receives its transmission factor.  `ffr` must hold filter_origin(ip).
// The collision tests the reference makes for a feeler whose first occluder is `first`
// (-1: none): objects 0..first, all of them when there is none, never `cur`.
template <typename AT>
__device__ __forceinline__ void count_feeler_tests(AT& a, int first, int cur, bool cur_plane, uint32_t& c_sph, uint32_t& c_pln) {
    const int last = (first >= 0) ? first + 1 : a.n_prims;
    const int tested = last - ((cur < last) ? 1 : 0);
    const int planes = a.planes_before[last] - ((cur_plane && cur < last) ? 1 : 0);
    c_pln += (uint32_t)planes;
    c_sph += (uint32_t)(tested - planes);
}
// The search without an anchor grid: the sphere tree, or the object list in container order.
template <bool COUNT, bool TREE, bool LATE, typename AT>
__device__ __forceinline__ int shadow_feeler_general(AT& a, cdata_t geom, cfdata_t fgeom, cfdata_t ltab, int li,
                                                     FilterRay& ffr, V3 ip, V3 fdir, int cur, bool cur_plane, float& occl,
                                                     uint32_t& c_sph, uint32_t& c_pln) {
    int first = -1;  // index of the first occluder in container order
    if (TREE) {
        // the walk meets occluders in any order: it only keeps the one with the
        // smallest index, whose factor is worked out afterwards
        double pa = 0.0, pb = 0.0;  // see collides()
        const int best = find_occluder_tree(a, geom, ffr, ip, fdir, cur, pa, pb, c_sph);
        first = (best != 0x7fffffff) ? best : -1;
        if (first >= 0) occl = occluder_factor(a.geom + first, ip, fdir, pa, pb);
#ifndef RT_PHASE_PROFILE
        if (COUNT) count_feeler_tests(a, first, cur, cur_plane, c_sph, c_pln);
#endif
    } else {
        cfdata_t lp = ltab + (size_t)li * a.ltab_stride * 4;
        cfdata_t fp = fgeom;
        cdata_t gp = geom;
        int base = 0;
        for (; base + kGroup <= a.n_prims; base += kGroup, lp += 4 * kGroup, fp += 8 * kGroup, gp += 8 * kGroup) {
            if (!__any(first < 0)) break;
            LRec lr[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) lr[j] = load_lrec(lp + 4 * j);
#pragma unroll
            for (int j = 0; j < kGroup; ++j)
                feeler_slot<COUNT, LATE>(lr[j], fp, gp, j, base + j, true, ffr, ip, fdir, cur, first, occl, c_sph,
                                   c_pln);
        }
        if (base < a.n_prims && __any(first < 0)) {
            const int rem = a.n_prims - base;
            LRec lr[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) lr[j] = load_lrec(lp + 4 * j);  // padded to whole groups
#pragma unroll
            for (int j = 0; j < kGroup - 1; ++j)
                feeler_slot<COUNT, LATE>(lr[j], fp, gp, j, base + j, j < rem, ffr, ip, fdir, cur, first, occl,
                                   c_sph, c_pln);
        }
    }
    return first;
}
// GRID: the spheres are searched through the light's anchor grid (every lane its own cell);
// lanes whose ray is outside the range the grids were built for take the general search.
template <bool COUNT, bool TREE, bool LATE, bool GRID, typename AT>
__device__ __forceinline__ int shadow_feeler(AT& a, cdata_t geom, cfdata_t fgeom, cfdata_t ltab, int li,
                                             FilterRay& ffr, V3 ip, V3 fdir, int cur, bool cur_plane, float& occl,
                                             uint32_t& c_sph, uint32_t& c_pln) {
    set_dir(ffr, fdir);
    if (!GRID) return shadow_feeler_general<COUNT, TREE, LATE>(a, geom, fgeom, ltab, li, ffr, ip, fdir, cur, cur_plane, occl, c_sph, c_pln);
    int first = -1;
    bool ok;
    RT_MARK("grid_cell");
    const GridList gl = grid_list(a, li, ffr.d, ok);
    ok = ok & (ffr.eray <= a.grid_eray_max);
    if (ok) {
        double pa = 0.0, pb = 0.0;  // see collides()
        int best = 0x7fffffff;
        RT_MARK("grid_planes");
        feeler_planes(a, geom, ffr, ip, fdir, cur, best, pa, pb);
        RT_MARK("grid_walk");
        find_occluder_grid(a, gl, ffr, ip, fdir, cur, best, pa, pb);
        RT_MARK("grid_factor");
        first = (best != 0x7fffffff) ? best : -1;
        if (first >= 0) occl = occluder_factor(a.geom + first, ip, fdir, pa, pb);
        RT_MARK("grid_tail");
#ifndef RT_PHASE_PROFILE
        if (COUNT) count_feeler_tests(a, first, cur, cur_plane, c_sph, c_pln);
#endif
    }
    if (!all_lanes(ok)) {
        if (!ok) first = shadow_feeler_general<COUNT, TREE, false>(a, geom, fgeom, ltab, li, ffr, ip, fdir, cur, cur_plane, occl, c_sph, c_pln);
    }
    return first;
}

// One shadow feeler from `ip` towards light `li` at `lpos` in a launch that uses the anchor
// grids: the certified float search first, the double search for the lanes it leaves unsure.
// Returns the first occluder in container order (-1: none), `occl` its transmission factor;
// `fd` receives the float direction (what the diffuse term uses, whichever search answered).
template <bool COUNT, bool TREE, typename AT>
__device__ __forceinline__ int feeler_by_grid(AT& a, cdata_t geom, cfdata_t fgeom, cfdata_t ltab, int li, V3 ip, V3 lpos,
                                              int cur, bool cur_plane, float& occl, F3& fd, uint32_t& c_sph, uint32_t& c_pln) {
    const V3 v = lpos - ip;
    const FeelRay fr = feel_ray(ip, v, a.scene_p1max);
    fd = fr.d;
    RT_MARK("grid_cell");
    bool ok;
    const GridList gl = grid_list(a, li, fr.d, ok);
    ok = ok & (fr.eray <= a.grid_eray_max) & fr.valid;
    RT_MARK("grid_walk");
    bool unsure;
    int first = feeler_float_grid(a, gl, fr, cur, occl, unsure);
    unsure = unsure | !ok;
#ifndef RT_PHASE_PROFILE
    if (COUNT && !unsure) count_feeler_tests(a, first, cur, cur_plane, c_sph, c_pln);
#endif
    RT_MARK("feeler_exact_search");
    if (!all_lanes(!unsure)) {
        if (unsure) {
            const V3 fdir = unit(v);  // feeler.dir (src/objects.h:38, src/common.h:27-31)
            FilterRay ffr = filter_origin(ip, a.scene_p1max);
            occl = 1.0f;
            first = shadow_feeler<COUNT, TREE, false, true>(a, geom, fgeom, ltab, li, ffr, ip, fdir, cur, cur_plane, occl, c_sph, c_pln);
        }
    }
    return first;
}

#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 64
#endif
constexpr int kRefillMin = RT_REFILL_MIN;

// Where the pixels of a wave come from when a launch has a fixed grid.  All members are
// wave-uniform.
//
// TileQueue (launches with a fixed grid): the launch's tiles - of one render, or of several
// one after the other - are numbered 0..n_tiles-1 and handed out by an atomic counter, four
// at a time (one at a time over the last quarter, which keeps the waves' finishing times
// together).  A wave takes new pixels when at least kRefillMin of its lanes are without one;
// kRefillMin is 64, i.e. a whole tile at a time once every lane has finished: lanes that start
// together stay in step, and a step costs the union of the paths its lanes take, so mixing
// pixels of different tiles and levels in a wave was measured to cost more than the idle
// lanes do (RT_REFILL_MIN 32: glass 4K depth 8 18.8 -> 23.4 ms).  What keeps lanes busy
// instead is share_rays() below (rays of the same tile handed from lane to lane).  Pixels of
// two renders never share a wave: a tile of another render waits until the lanes still
// walking have finished.
struct TileQueue {
    CArgs* renders;
    const RT_CONST int32_t* tile_end;   // running totals of the renders' tiles; null = one render
    unsigned int* queue;       // [0] next tile to hand out, [1] waves of the launch that have finished
    int n_tiles;
    int n_waves;               // waves of the launch
    int cur = 0;               // render of the walking lanes
    int tile = 0, tiles_left = 0;  // tiles in hand: tile, tile+1, ... (not yet started)
    int pos = 64;              // next pixel of the current tile, 64 = used up
    int tx = 0, ty = 0;        // the current tile
    int last = 0;              // the counter's value at the last fetch
    bool exhausted = false;
};

// Give every lane without a pixel (`have` false) the next pixel of the queue, as long as
// there are any; begin(x, k) starts a pixel's walk (and leaves `have` false for a pixel
// outside the frame, whose lane then takes another).  Called by all lanes of the wave.
// ONE_TILE: stop after the pixels of one tile have been dealt out (lanes whose pixel lies outside
// the frame stay without one), so that which pixels share a wave never depends on the order in
// which the waves of the launch reach the queue.
template <bool ONE_TILE = false, typename Begin>
__device__ __forceinline__ void take_pixels(TileQueue& f, CArgs*& ap, bool& have, Begin& begin) {
    const int me = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // New pixels are taken in batches of at least kRefillMin (lanes that start together stay
    // in step with each other: their hit blocks and shadow loops run once for all of them)
    if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(!have)) < kRefillMin) return;
    for (;;) {
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!have);
        if (idle == 0ull) break;
        if (f.pos == 64) {  // the current tile is used up: start the next one
            if (f.tiles_left == 0) {
                if (f.exhausted) break;
                const unsigned int chunk = ((long long)f.last * 4 < (long long)f.n_tiles * 3) ? 4u : 1u;
                unsigned int t = 0;
                if (me == 0) t = atomicAdd(f.queue, chunk);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= (unsigned int)f.n_tiles) { f.exhausted = true; break; }
                f.last = (int)t;
                f.tile = (int)t;
                const int rest = f.n_tiles - (int)t;
                f.tiles_left = rest < (int)chunk ? rest : (int)chunk;
            }
            int r = 0, first = 0;
            if (f.tile_end) {
                while (f.tile >= f.tile_end[r]) ++r;
                first = r ? f.tile_end[r - 1] : 0;
            }
            // pixels of two renders never share the wave
            if (r != f.cur && __builtin_amdgcn_ballot_w64(have) != 0ull) break;
            f.cur = r;
            ap = f.renders + r;
            const int tr = f.tile - first;
            f.ty = tr / ap->tiles_x;
            f.tx = tr - f.ty * ap->tiles_x;
            f.pos = 0;
            ++f.tile;
            --f.tiles_left;
        }
        const int n_idle = __builtin_popcountll(idle);
        const int m = (64 - f.pos) < n_idle ? (64 - f.pos) : n_idle;
        // rank of this lane among the idle ones
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        if (!have && rank < m) {
            const int l = f.pos + rank;
            begin(f.tx * kTile + (l & 7), f.ty * kTile + (l >> 3));
        }
        f.pos += m;
        if (ONE_TILE && __builtin_amdgcn_ballot_w64(have) != 0ull) break;
    }
}

// The walks of a wave's pixels: lane-private except for the scalar scene loads.
// MW = mask words kept per lane (0: neither hit_id nor masks are produced).
// TREE = walk the sphere tree instead of the whole object list (large scenes).
// DEEP = the stack may reach its global-memory part (max_depth > 3 + kStackLdsEntries);
//        without it that code is left out, which the register allocation needs.
// LATE = see feeler_slot() (container-order shadow loops of scenes with many lights).
// GRID = shadow feelers and primary rays find their spheres through the anchor grids.
// The step itself (whitted_walk_step.inc.h) is shared by the two functions below.

#ifdef RT_PHASE_PROFILE
// wave entries per phase: 0 iterations, 1 exact tests of the nearest-hit loop, 2 miss
// branch, 3 hit branch, 4 feelers, 5 exact tests of the feeler loops, 6 child rays built,
// 7 lane-iterations (per lane, not per wave)
#define RT_WALK_COUNTERS                                                                                      \
    uint32_t dummy = 0;                                                                                       \
    uint32_t &c_rays = dummy, &c_phits = dummy, &c_hits = dummy, &c_sph = cnt[1], &c_pln = dummy, &c_feel = dummy, \
             &c_occ = dummy, &c_refr = dummy, &c_refl = dummy, &c_ktest = dummy, &c_fex = cnt[5];
#define RT_EVENT(x) ((void)0)
#else
#define RT_WALK_COUNTERS                                                                                      \
    uint32_t &c_rays = cnt[0], &c_phits = cnt[1], &c_hits = cnt[2], &c_sph = cnt[3], &c_pln = cnt[4],       \
             &c_feel = cnt[5], &c_occ = cnt[6], &c_refr = cnt[7], &c_refl = cnt[8], &c_ktest = cnt[9];       \
    uint32_t& c_fex = c_sph;
#define RT_EVENT(x) do { if (COUNT) { x; } } while (0)
#endif

// The primary ray of pixel (x, row k of the strip): src/main.cc:135-138,
// src/view_frustum.h:37-46, src/common.h:27-31 (origin 0, direction returned).
// Frame row of strip row k (increasing in k).
template <typename AT>
__device__ __forceinline__ int frame_row(AT& a, int k) {
    // band = k / band_rows without the division (KArgs::band_magic)
    const int band = (a.band_rows == 1) ? k : (int)__umulhi((uint32_t)k, a.band_magic);
    return (a.band_first + band * a.band_step) * a.band_rows + (k - band * a.band_rows);
}
template <typename AT>
__device__ __forceinline__ V3 primary_direction(AT& a, int x, int k) {
    const int y = frame_row(a, k);
    const int xi = x - a.width / 2;
    const int yi = y - a.height / 2;
    const V3 v = mk((xi + 0.5) * a.pw, (yi + 0.5) * a.ph, 1.0) - mk(0.0, 0.0, 0.0);
    // the host has checked the range guard of unit() for every pixel of the frame
    if (a.primary_in_range) return unit_mid(v, len2(v));
    return unit(v);
}

// Store a finished pixel: src/screen.h:62-64, src/color.h:27-29.
template <int MW, typename AT>
__device__ __forceinline__ void store_pixel(AT& a, int x, int k, float cr, float cg, float cb, int32_t first_hit,
                                            const uint32_t (&hmask)[MW > 0 ? MW : 1],
                                            const uint32_t (&smask)[MW > 0 ? MW : 1]) {
    const size_t px = (size_t)k * a.width + x;
    uint8_t* dst = a.bgr + 3 * px;
    dst[0] = (uint8_t)to_u8(cb);
    dst[1] = (uint8_t)to_u8(cg);
    dst[2] = (uint8_t)to_u8(cr);
    if (MW > 0) {
        if (a.hit_id) a.hit_id[px] = first_hit;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            if (a.hit_mask) a.hit_mask[px * MW + i] = hmask[i];
            if (a.spawn_mask) a.spawn_mask[px * MW + i] = smask[i];
        }
    }
}

// ---- pixels that end at the primary ray, in single precision ---------------------------------
// A primary ray that certainly misses every object gives the background.  One that certainly
// hits exactly one object, a plane whose material has reflectivity 0 and transparency 0, ends
// at that hit as well when only the image is asked for (children under a coefficient of zero
// are left out): the pixel is the plane's own shade, src/objects.h:20-65 at depth 1.  Nothing
// computed for such a pixel decides anything but colours, PROVIDED every decision on the way is
// certain: the primary ray's tests (classify_plane; the line test from the camera's records for
// the spheres) and, per light, the first occluder of the shadow feeler (classify_plane /
// classify_sphere over all objects in container order).  Such a pixel is finished here without
// the double-precision walk.  A lane that meets a single test that is not certain leaves the
// pass without a result and is walked as usual; whether a lane qualifies depends on its own
// pixel only, so a pixel does not depend on its neighbours or on how the frame is cut.
//
// The hit point.  For the un-normalised pixel vector v (src/view_frustum.h:37-46, exact in
// double) and the plane (p0, n), h = p0.n: ip = v h / (v.n), the reference's o + d t for o = 0
// up to rounding.  The pass only takes planes met at |cos| = |v.n| / |v| > 1e-3; then v.n is
// within 5.8e-13 of itself (3 x 2^-53 |v|_1 from the sums, over |v| 1e-3) and h within
// 3.4e-16 |p0|_1, so this ip lies within 6e-13 (|ip| + |p0|_1) of the exact intersection, and
// so does the reference's own.  A feeler at distance dist >= 1e-3 (|ip|_1 + |p0|_1) from its
// light turns by less than 1.2e-9 for that, inside the slack of feel_ray()'s direction bound
// (its steps add up to 5.5 x 2^-24 = 3.28e-7 of the 3.6e-7 allowed); the shift of the origin
// itself is added to the bounds of the origin-dependent tests (en_ray, efront) as
// 1e-11 (|ip|_1 + |p0|_1).  |p0|_1 is taken from the plane's record:
// herr = 1.01 U (|h| + |p0|_1).
//
// A wave of this pass is short, and what bounds a launch made of such waves is the scalar unit:
// a CU issues one scalar instruction per cycle for its four SIMDs, as many as it issues vector
// instructions, and this code - records in scalar registers, a branch around every test that
// most lanes do not need - has more scalar instructions than vector ones.  So nothing here may
// cost scalar instructions for form's sake: the offsets into the table come from the host
// (KArgs::front_anchors ..), and the loads are left where the compiler puts them (pinning the
// values of a load in registers of their own, to issue the loads in three groups, cost 74 copies
// per wave: default 4K frame 0.0838 -> 0.0798 ms without them).
// The head of KArgs (kFrontArgDwords dwords) in scalar registers.
struct FrontArgs {
    uint32_t w[kFrontArgDwords];
    __device__ __forceinline__ cfdata_t front() const { return (cfdata_t)(const float*)(((uint64_t)w[1] << 32) | w[0]); }
    __device__ __forceinline__ uint8_t* bgr() const { return (uint8_t*)(((uint64_t)w[3] << 32) | w[2]); }
    __device__ __forceinline__ double pw() const { return __hiloint2double((int)w[5], (int)w[4]); }
    __device__ __forceinline__ double ph() const { return __hiloint2double((int)w[7], (int)w[6]); }
    __device__ __forceinline__ int n_planes() const { return (int)w[8]; }
    __device__ __forceinline__ int n_front() const { return (int)w[9]; }
    __device__ __forceinline__ int n_lights() const { return (int)w[10]; }
    __device__ __forceinline__ int width() const { return (int)w[11]; }
    __device__ __forceinline__ int height() const { return (int)w[12]; }
    __device__ __forceinline__ int rows() const { return (int)w[13]; }
    __device__ __forceinline__ int band_rows() const { return (int)w[14]; }
    __device__ __forceinline__ int band_first() const { return (int)w[15]; }
    __device__ __forceinline__ int band_step() const { return (int)w[16]; }
    __device__ __forceinline__ uint32_t band_magic() const { return w[17]; }
    __device__ __forceinline__ float scene_p1max() const { return __uint_as_float(w[18]); }
    __device__ __forceinline__ int sky_unit_shift() const { return (int)w[19]; }
    __device__ __forceinline__ float bgr_c(int i) const { return __uint_as_float(w[20 + i]); }
    __device__ __forceinline__ uint32_t row_order() const { return w[23]; }
    __device__ __forceinline__ int front_anchors() const { return (int)w[24]; }
    __device__ __forceinline__ int front_stride() const { return (int)w[25]; }
    __device__ __forceinline__ int front_camera() const { return (int)w[26]; }
    __device__ __forceinline__ int front_spheres() const { return (int)w[27]; }
};
// The arguments at `kp` (a KArgs in the constant address space).
__device__ __forceinline__ FrontArgs load_front_args(const RT_CONST uint32_t* kp) {
    FrontArgs f;
#pragma unroll
    for (int i = 0; i < kFrontArgDwords; ++i) f.w[i] = kp[i];
    return f;
}

struct FrontState {
    bool ok;       // every test so far was certain
    int cur;       // the plane that is hit, -1 = none
    bool ends;     // its material spawns no child
    double vq, hd; // v.n and p0.n of that plane
    float pb;      // >= |p0|_1
    F3 nf, kd;     // its normal and diffuse colour
    float amb;
};
struct FrontPlaneRec {  // the first half of a FrontPlane in scalar registers
    float n0, n1, n2, ends, h, herr;
    int idx;
    float transp;
};
__device__ __forceinline__ FrontPlaneRec front_plane(const uint32_t* w) {
    FrontPlaneRec r;
    r.n0 = __uint_as_float(w[0]); r.n1 = __uint_as_float(w[1]); r.n2 = __uint_as_float(w[2]);
    r.ends = __uint_as_float(w[3]); r.h = __uint_as_float(w[4]); r.herr = __uint_as_float(w[5]);
    r.idx = (int)w[6];
    r.transp = __uint_as_float(w[7]);
    return r;
}
__device__ __forceinline__ FRec as_frec(const FrontPlaneRec& r) {
    FRec g;
    g.a0 = r.n0; g.a1 = r.n1; g.a2 = r.n2; g.r2 = -1.0f; g.b0 = r.h; g.b1 = r.herr;
    return g;
}
// The primary ray against the plane record in w[0..19].
__device__ __forceinline__ void front_primary_plane(const uint32_t* w, const FeelRay& pr, V3 v, FrontState& s) {
    const FrontPlaneRec r = front_plane(w);
    bool hit;
    float factor = 0.0f;
    const bool sure = classify_plane(as_frec(r), 0.0f, pr, hit, factor);
    s.ok = s.ok & sure & !(hit & (s.cur >= 0));  // (two planes hit: which is nearer is not looked into)
    if (sure & hit) {
        s.cur = r.idx;
        s.ends = r.ends != 0.0f;
        const V3 q = mk(__hiloint2double((int)w[9], (int)w[8]), __hiloint2double((int)w[11], (int)w[10]),
                        __hiloint2double((int)w[13], (int)w[12]));
        s.vq = dot(v, q);
        s.hd = __hiloint2double((int)w[15], (int)w[14]);
        s.pb = r.herr * (1.0f / (1.01f * 6.0e-8f));
        s.nf = f3(r.n0, r.n1, r.n2);
        s.kd = f3(__uint_as_float(w[16]), __uint_as_float(w[17]), __uint_as_float(w[18]));
        s.amb = __uint_as_float(w[19]);
    }
}
// The primary ray against the sphere record in w[0..3]: the line misses the sphere, or the
// sphere lies behind the camera (src/objects.h:128-129).
__device__ __forceinline__ void front_primary_sphere(const uint32_t* w, const FeelRay& pr, bool& ok) {
    const float t = __builtin_fmaf(__uint_as_float(w[2]), pr.d.z, __builtin_fmaf(__uint_as_float(w[1]), pr.d.y, __uint_as_float(w[0]) * pr.d.x));
    ok = ok & ((__builtin_fmaf(-t, t, __uint_as_float(w[3])) > pr.eray) | (t < -pr.efront));
}
// A shadow feeler against the plane record in w[0..7] / the sphere record in w[0..7], to be
// called in container order; `cur` = the object the feeler starts on, which is skipped.
__device__ __forceinline__ void front_feeler_plane(const uint32_t* w, const FeelRay& fr, int cur, bool& good, int& best, float& occl) {
    const FrontPlaneRec r = front_plane(w);
    if (good & (r.idx != cur) & (r.idx < best)) {
        bool hit;
        float factor = 0.0f;
        good = classify_plane(as_frec(r), r.transp, fr, hit, factor);
        if (good & hit) { best = r.idx; occl = factor; }
    }
}
__device__ __forceinline__ void front_feeler_sphere(const uint32_t* w, const FeelRay& fr, int cur, bool& good, int& best, float& occl) {
    LRec l;
    l.x = __uint_as_float(w[0]); l.y = __uint_as_float(w[1]); l.z = __uint_as_float(w[2]); l.thr = __uint_as_float(w[3]);
    const int idx = (int)w[4];
    const float t = __builtin_fmaf(l.z, fr.d.z, __builtin_fmaf(l.y, fr.d.y, l.x * fr.d.x));
    if (good & (idx != cur) & (idx < best) & !(__builtin_fmaf(-t, t, l.thr) > fr.eray)) {
        bool hit;
        float factor = 0.0f;
        good = classify_sphere(l, t, __uint_as_float(w[5]), __uint_as_float(w[6]), __uint_as_float(w[7]), fr, hit, factor);
        if (good & hit) { best = idx; occl = factor; }
    }
}
template <int N>
__device__ __forceinline__ void front_load(uint32_t (&w)[N], cfdata_t p) {
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = __float_as_uint(p[i]);
}
// Returns true for the lanes whose pixel has been stored.
__device__ __forceinline__ bool primary_only_pixel(const FrontArgs& a, const int x, const int k, const bool inside) {
    const float U = 6.0e-8f;
    // the table: plane records (an even number of them), then per anchor - the lights, then the
    // camera - a header and the sphere records (an even number, padded with records that never hit)
    const cfdata_t planes = a.front();
    const int n_planes = a.n_planes();
    const int spheres2 = a.front_spheres();
    const cfdata_t anchors = planes + a.front_anchors();
    const int anchor_stride = a.front_stride();
    const cfdata_t camera = planes + a.front_camera();
    uint32_t p0[20], c0[4], c1[4];
    front_load(p0, planes);  // (a scene without planes: whatever lies there, not looked at)
    front_load(c0, camera);
    front_load(c1, camera + 8);

    // src/main.cc:135-138, src/view_frustum.h:37-46: frame row of strip row k, pixel vector
    const int band = (a.band_rows() == 1) ? k : (int)__umulhi((uint32_t)k, a.band_magic());
    const int y = (a.band_first() + band * a.band_step()) * a.band_rows() + (k - band * a.band_rows());
    const V3 v = mk((x - a.width() / 2 + 0.5) * a.pw(), (y - a.height() / 2 + 0.5) * a.ph(), 1.0);
    // the primary ray as a FeelRay whose anchor (the camera) is its origin: dist = 0
    FeelRay pr;
    pr.o = f3(0.0f, 0.0f, 0.0f);
    const F3 vf = f3((float)v.x, (float)v.y, (float)v.z);
    const float s2 = dotf(vf, vf);
    const float inv = __builtin_amdgcn_rsqf(s2);
    const float vlen = s2 * inv;
    pr.d = f3(vf.x * inv, vf.y * inv, vf.z * inv);
    pr.dist = 0.0f;
    pr.en_ray = 0.0f;
    pr.eray = 1.0e-13f * a.scene_p1max() * a.scene_p1max();
    pr.efront = 16.0f * U * a.scene_p1max();
    pr.valid = true;

    FrontState s;
    s.ok = inside & (s2 < 1.0e30f);  // (s2 >= 1: the float normalisation keeps its error bound)
    s.cur = -1;
    s.ends = false;
    s.vq = 1.0; s.hd = 0.0; s.pb = 0.0f; s.amb = 0.0f;
    s.nf = f3(0.0f, 0.0f, 0.0f); s.kd = s.nf;
    if (n_planes > 0) front_primary_plane(p0, pr, v, s);
    for (int j = 1; j < n_planes; ++j) {
        uint32_t pj[20];
        front_load(pj, planes + j * kFrontPlaneDwords);
        front_primary_plane(pj, pr, v, s);
    }
    front_primary_sphere(c0, pr, s.ok);
    front_primary_sphere(c1, pr, s.ok);
    for (int j = 2; j < spheres2; ++j) {
        uint32_t cj[4];
        front_load(cj, camera + j * 8);
        front_primary_sphere(cj, pr, s.ok);
    }
    if (!__any(s.ok)) return false;

    float cr = a.bgr_c(0), cg = a.bgr_c(1), cb = a.bgr_c(2);  // nothing hit: src/main.cc:49-53
    const bool on_plane = s.ok & (s.cur >= 0);
    // (a plane with children, or one met at a flat angle, is left to the walk)
    const bool shade = on_plane & s.ends & (__builtin_fabsf((float)s.vq) > 1.1e-3f * vlen);
    bool ok = s.ok & (shade | !on_plane);
    if (__any(shade)) {
        const V3 ip = v * (s.hd * rcp_refined(shade ? s.vq : 1.0));
        float sr = 0.0f, sg = 0.0f, sb = 0.0f;
        bool good = shade;
        cfdata_t anchor = anchors;
        for (int li = 0; li < a.n_lights(); ++li, anchor += anchor_stride) {
            uint32_t lw[9], e0[8], e1[8];
            front_load(lw, anchor);  // FrontAnchor: position (three doubles), colour
            front_load(e0, anchor + kFrontAnchorDwords);
            front_load(e1, anchor + kFrontAnchorDwords + 8);
            const V3 lpos = mk(__hiloint2double((int)lw[1], (int)lw[0]), __hiloint2double((int)lw[3], (int)lw[2]),
                               __hiloint2double((int)lw[5], (int)lw[4]));
            const F3 ci = f3(__uint_as_float(lw[6]), __uint_as_float(lw[7]), __uint_as_float(lw[8]));
            FeelRay fr = feel_ray(ip, lpos - ip, a.scene_p1max());
            const float reach = l1f(fr.o) + s.pb;
            good = good & fr.valid & (fr.dist > 1.0e-3f * reach);
            fr.en_ray = __builtin_fmaf(1.0e-11f, reach, fr.en_ray);
            fr.efront = __builtin_fmaf(1.0e-11f, reach, fr.efront);
            int best = 0x7fffffff;
            float occl = 1.0f;
            if (n_planes > 0) front_feeler_plane(p0, fr, s.cur, good, best, occl);
            for (int j = 1; j < n_planes; ++j) {
                uint32_t pj[8];
                front_load(pj, planes + j * kFrontPlaneDwords);
                front_feeler_plane(pj, fr, s.cur, good, best, occl);
            }
            front_feeler_sphere(e0, fr, s.cur, good, best, occl);
            front_feeler_sphere(e1, fr, s.cur, good, best, occl);
            for (int j = 2; j < spheres2; ++j) {
                uint32_t ej[8];
                front_load(ej, anchor + kFrontAnchorDwords + j * 8);
                front_feeler_sphere(ej, fr, s.cur, good, best, occl);
            }
            // src/objects.h:50-60, the operations of the walk's shading in the same order
            const float dk = __builtin_fabsf(dotf(s.nf, fr.d)) * occl;
            sr = __builtin_fmaxf(__builtin_fmaf(s.kd.x * ci.x, dk, sr), 0.0f);
            sb = __builtin_fmaxf(__builtin_fmaf(s.kd.z * ci.z, dk, sb), 0.0f);
            sg = __builtin_fmaxf(__builtin_fmaf(s.kd.y * ci.y, dk, sg), 0.0f);
        }
        ok = ok & (good | !on_plane);
        // own colour at depth 1 (1.0 / depth = 1) with transparency 0 and path weight 1,
        // src/main.cc:111-112
        if (on_plane) {
            cr = __builtin_fmaf(a.bgr_c(0), s.amb, sr);
            cg = __builtin_fmaf(a.bgr_c(1), s.amb, sg);
            cb = __builtin_fmaf(a.bgr_c(2), s.amb, sb);
        }
    }
    if (ok) {
#ifdef RT_FRONT_MARK  // analysis builds: pixels finished by this pass in one colour
        cr = 1.0f; cg = 0.0f; cb = 1.0f;
#endif
        // src/screen.h:62-64, src/color.h:27-29
        uint8_t* dst = a.bgr() + 3 * ((size_t)k * a.width() + x);
        dst[0] = (uint8_t)to_u8(cb);
        dst[1] = (uint8_t)to_u8(cg);
        dst[2] = (uint8_t)to_u8(cr);
    }
    return ok;
}

// One wave, one 8x8 tile: (x, k) = this lane's column and strip row; `have` is false for
// lanes without a pixel.
template <int MW, bool COUNT, bool TREE, bool DEEP, bool LATE, bool GRID, typename AT>
__device__ __forceinline__ void trace_tile(AT& a, const int x, const int k, bool have, double* stk, double* gstk,
                                           const int lane, uint32_t (&cnt)[kCounters]) {
    const cdata_t geom = (cdata_t)(const double*)a.geom;
    const cdata_t lights = (cdata_t)(const double*)a.lights;
    const cfdata_t fgeom = (cfdata_t)(const float*)a.fgeom;
    const cfdata_t ltab = (cfdata_t)(const float*)a.ltab;
    const bool inside = have;
    const F3 bgr = f3(a.bgr_c[0], a.bgr_c[1], a.bgr_c[2]);

    V3 o = mk(0.0, 0.0, 0.0);
    V3 d = primary_direction(a, x, k);
    float w = 1.0f;  // product of the reflection/refraction coefficients on the path (colour)
    uint32_t node = 1;
    int sp = 0;
    // The stack (whitted_device.h: tile launches): entries 0 and 1 in LDS, the last one in
    // the lane's private memory - the pointer is made opaque so that it stays there instead of
    // taking fourteen registers for a value that is rarely used.
    double last_entry_store[kStackFields];
    double* last_entry = last_entry_store;
    __asm__ volatile("" : "+v"(last_entry));
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    int32_t first_hit = -1;
    uint32_t hmask[MW > 0 ? MW : 1] = {0};
    uint32_t smask[MW > 0 ? MW : 1] = {0};
    RT_WALK_COUNTERS
    RT_MARK("prologue");

    double bottom[kStackFields];  // (not used by this layout)
#define RT_WALK_FINISHED have = false
#define RT_STACK_IN_LDS true
#define RT_STACK_LOW 0
    while (have) {
#include "whitted_walk_step.inc.h"
    }
#undef RT_STACK_LOW
#undef RT_STACK_IN_LDS
#undef RT_WALK_FINISHED

    RT_MARK("epilogue");
    if (inside) store_pixel<MW>(a, x, k, cr, cg, cb, first_hit, hmask, smask);
}

// ---- rays shared between the lanes of a wave ---------------------------------------------------
// The ray trees of a tile's pixels differ in size, and a wave walks until its longest one is
// done: in a deep frame most lanes end up waiting.  Where only the image is asked for nothing but
// the colour sum ties a ray to its pixel, so a lane without a ray is given the pending ray at the
// BOTTOM of a busy lane's stack (the child of the shallowest node that still waits there: the
// largest piece of work that lane can hand over), walks that subtree on its own stack and adds
// what it collects to the pixel the ray came from.  The pixel's colour is then a sum in another
// order (single precision, decides nothing); every ray is still traced exactly once with the same
// doubles.  A stack therefore has a low end as well: entries lo .. sp-1 are pending.
// `share` (kShareDwords dwords of LDS per wave): three rows of 64 colour sums, one per pixel of
// the wave, and 64 exchange words.
__device__ __forceinline__ void lds_add(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// Pair the lanes without a ray with lanes that have a pending entry, in lane order.  The taker
// announces itself in the exchange word of its rank, the giver writes its lowest entry - with the
// lane that owns the pixel in bits 16.. of the node field - into the taker's first LDS slot (free:
// the taker's stack is empty) and the taker continues with it.  Called by all lanes of the wave.
template <bool DEEP>
__device__ __forceinline__ void share_rays(bool& have, V3& o, V3& d, float& w, uint32_t& node, int& sp, int& lo, int& owner,
                                           const double (&bottom)[kStackFields], const double* last_entry, double* stk,
                                           const double* gstk, uint32_t* xchg, const int lane) {
    const bool can_give = have && sp > lo;
    const unsigned long long idle = __builtin_amdgcn_ballot_w64(!have);
    const unsigned long long givers = __builtin_amdgcn_ballot_w64(can_give);
    if (idle == 0ull || givers == 0ull) return;
    const int n_idle = __builtin_popcountll(idle), n_give = __builtin_popcountll(givers);
    const int n = n_idle < n_give ? n_idle : n_give;
    const int irank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
    const int grank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(givers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)givers, 0u));
    const bool take = !have && irank < n;
    if (take) xchg[irank] = (uint32_t)lane;
    __builtin_amdgcn_wave_barrier();
    if (can_give && grank < n) {
        const int to = (int)xchg[grank];
        double ent[kStackFields];
        if (DEEP) {
            if (lo < kStackRegEntries) {
#pragma unroll
                for (int f = 0; f < kStackFields; ++f) ent[f] = bottom[f];
            } else if (lo < kStackRegEntries + kStackLdsEntries) {
                const double* e = stk + (size_t)(lo - kStackRegEntries) * kStackFields * 64;
#pragma unroll
                for (int f = 0; f < kStackFields; ++f) ent[f] = e[f * 64];
            } else {
                const double* e = gstk + (size_t)(lo - kStackRegEntries - kStackLdsEntries) * kStackFields * 64 + lane;
#pragma unroll
                for (int f = 0; f < kStackFields; ++f) ent[f] = e[f * 64];
            }
        } else {
            if (lo < kStackLdsEntries) {
                const double* e = stk + (size_t)lo * kStackFields * 64;
#pragma unroll
                for (int f = 0; f < kStackFields; ++f) ent[f] = e[f * 64];
            } else {
#pragma unroll
                for (int f = 0; f < kStackFields; ++f) ent[f] = last_entry[f];
            }
        }
        ent[6] = __hiloint2double(__double2hiint(ent[6]) | (owner << 16), __double2loint(ent[6]));
        double* e = stk - lane + to;
#pragma unroll
        for (int f = 0; f < kStackFields; ++f) e[f * 64] = ent[f];
        ++lo;
    }
    __builtin_amdgcn_wave_barrier();
    if (take) {
        double ent[kStackFields];
#pragma unroll
        for (int f = 0; f < kStackFields; ++f) ent[f] = stk[f * 64];
        o = mk(ent[0], ent[1], ent[2]);
        d = mk(ent[3], ent[4], ent[5]);
        w = __int_as_float(__double2loint(ent[6]));
        const uint32_t tag = (uint32_t)__double2hiint(ent[6]);
        node = tag & 0xffffu;
        owner = (int)(tag >> 16);
        sp = 0;
        lo = 0;
        have = true;
    }
    __builtin_amdgcn_wave_barrier();
}

// One wave of a launch with a fixed grid: pixels from the tile queue until it is empty.
template <int MW, bool COUNT, bool TREE, bool DEEP, bool LATE, bool GRID>
__device__ __forceinline__ void trace_queue(TileQueue& feed, double* stk, double* gstk, float* share, const int lane,
                                            uint32_t (&cnt)[kCounters]) {
    constexpr bool SHARE = share_rays_in(MW, COUNT, GRID);
    // SHARE: the lane's pixel is stored when the whole wave has run out of rays (`pixel` = this
    // lane holds one that is not stored yet), and a lane's colour goes to the pixel of `owner`
    bool pixel = false;
    int owner = lane;
    int lo = 0;  // SHARE: entries lo .. sp-1 of the stack are pending
    uint32_t* xchg = (uint32_t*)(share + 3 * 64);
    CArgs* ap = feed.renders + feed.cur;
    int x = 0, k = 0;
    bool have = false;
    V3 o = mk(0.0, 0.0, 0.0), d = mk(0.0, 0.0, 1.0);
    float w = 1.0f;
    uint32_t node = 1;
    int sp = 0;
    // The stack (whitted_device.h): with the global part (DEEP) entry 0 in registers, the next
    // two in LDS, the rest in the workspace; without it as in trace_tile - entries 0 and 1 in
    // LDS, the last one in the lane's private memory.
    double bottom[kStackFields] = {0, 0, 0, 0, 0, 0, 0};
    double last_entry_store[kStackFields];
    double* last_entry = last_entry_store;
    __asm__ volatile("" : "+v"(last_entry));
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    int32_t first_hit = -1;
    uint32_t hmask[MW > 0 ? MW : 1] = {0};
    uint32_t smask[MW > 0 ? MW : 1] = {0};
    RT_WALK_COUNTERS

    // start the walk of pixel (nx, nk) in this lane
    auto begin_pixel = [&](int nx, int nk) {
        CArgs& a = *ap;
        x = nx;
        k = nk;
        have = (nx < a.width) && (nk < a.rows);
        o = mk(0.0, 0.0, 0.0);
        d = primary_direction(a, nx, nk);
        w = 1.0f;
        node = 1;
        sp = 0;
        cr = 0.0f; cg = 0.0f; cb = 0.0f;
        if (MW > 0) {
            first_hit = -1;
#pragma unroll
            for (int i = 0; i < MW; ++i) { hmask[i] = 0; smask[i] = 0; }
        }
    };

#define RT_WALK_FINISHED                                                     \
    do {                                                                     \
        if (SHARE) {                                                         \
            lds_add(share + owner, cr);                                      \
            lds_add(share + 64 + owner, cg);                                 \
            lds_add(share + 128 + owner, cb);                                \
            cr = 0.0f; cg = 0.0f; cb = 0.0f;                                 \
            sp = 0; lo = 0;                                                  \
        } else {                                                             \
            store_pixel<MW>(a, x, k, cr, cg, cb, first_hit, hmask, smask);  \
        }                                                                    \
        have = false;                                                        \
    } while (0)
#define RT_STACK_IN_LDS (!DEEP)
#define RT_STACK_LOW (SHARE ? lo : 0)
    for (;;) {
        RT_MARK("refill");
        if (SHARE) {
            if (!__any(have)) {
                // the wave has run out of rays: its pixels are complete
                if (pixel) store_pixel<MW>(*ap, x, k, share[lane], share[64 + lane], share[128 + lane], first_hit, hmask, smask);
                share[lane] = 0.0f; share[64 + lane] = 0.0f; share[128 + lane] = 0.0f;
                take_pixels<true>(feed, ap, have, begin_pixel);
                pixel = have;
                owner = lane;
                if (!__any(have)) break;
            } else {
                share_rays<DEEP>(have, o, d, w, node, sp, lo, owner, bottom, last_entry, stk, gstk, xchg, lane);
            }
        } else {
            take_pixels(feed, ap, have, begin_pixel);
            if (!__any(have)) break;
        }
        if (have) {
            CArgs& a = *ap;
            const cdata_t geom = (cdata_t)(const double*)a.geom;
            const cdata_t lights = (cdata_t)(const double*)a.lights;
            const cfdata_t fgeom = (cfdata_t)(const float*)a.fgeom;
            const cfdata_t ltab = (cfdata_t)(const float*)a.ltab;
            const F3 bgr = f3(a.bgr_c[0], a.bgr_c[1], a.bgr_c[2]);
#include "whitted_walk_step.inc.h"
        }
    }
#undef RT_STACK_LOW
#undef RT_STACK_IN_LDS
#undef RT_WALK_FINISHED
    // The last wave of the launch to finish puts the queue back to zero for the next launch
    // that uses it (launches that share a queue run one after the other), so the host never
    // has to clear it.
    if (lane == 0) {
        const unsigned int done = atomicAdd(feed.queue + 1, 1u);
        if (done == (unsigned int)feed.n_waves - 1u) {
            __hip_atomic_store(feed.queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(feed.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
#undef RT_EVENT
#undef RT_WALK_COUNTERS

}  // namespace exact
}  // namespace rt
