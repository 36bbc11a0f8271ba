/*
 * whitted_oracle.c - CPU restatement of the reference's per-pixel Whitted trace.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing shipped links, loads or calls this file: it is the
 * checker the HIP library is compared with (tests/, the smoke test) and the single-thread
 * CPU baseline that the benchmark times beside the GPU.
 *
 * PARITY UNPINNED: the reference (727 lines of Windows-only C++) has no tests, golden
 * images or fixtures, it needs <Windows.h> and the un-vendored imvu/gmtl submodule
 * (reference .gitmodules:1-3, directory empty, commit unknown), neither of which exists
 * here, so it cannot be compiled and run to produce vectors.  This file restates its
 * arithmetic operation by operation from the sources, and gmtl's Vec3d operations from
 * that library's published definitions (VecOps.h: dot and lengthSquared as a running sum
 * from 0 over i = 0,1,2; length = sqrt(lengthSquared); normalize = length, then three
 * divisions when the length is not 0; makeNormal = normalize a copy; element-wise + - * /).
 * The only reference-derived numbers it is checked against are the event counts of an
 * instrumented build of the reference recorded when the sources were surveyed
 * (tests/golden/survey_probe_counts.json).
 *
 * Every double operation below happens in the order the reference writes it.  Build with
 *   gcc -std=gnu11 -O2 -ffp-contract=off  (no -ffast-math, no -march=native)
 * so that no multiply-add is fused: the reference's only compiler (MSVC x64,
 * /fp:precise) does not contract either.
 *
 * Inputs and outputs are the structs of include/rt_hip.h, so a test hands the very same
 * bytes to this file and to the device library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_hip.h"

typedef struct { double x, y, z; } v3;

/* ---- gmtl::Vec3d subset (call sites: src/main.cc:55-100, src/objects.h:38-52,88-142,
 *      src/common.h:24,30) ------------------------------------------------------------- */
static inline v3 v3_make(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 v3_scale(v3 a, double s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_div(v3 a, double s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline double v3_dot(v3 a, v3 b) {
    double r = 0.0;
    r += a.x * b.x;
    r += a.y * b.y;
    r += a.z * b.z;
    return r;
}
static inline double v3_len2(v3 a) { return v3_dot(a, a); }
static inline double v3_len(v3 a) { return sqrt(v3_len2(a)); }
static inline v3 v3_unit(v3 a) { /* gmtl::normalize on a copy == gmtl::makeNormal */
    double len = v3_len(a);
    if (len != 0.0) { a.x /= len; a.y /= len; a.z /= len; }
    return a;
}

/* src/color.h:77-79: float mix(a, b, amount) - the double result is rounded to float. */
static inline double mix_f(double a, double b, double amount) {
    float r = (float)((b * amount) + a * (1 - amount));
    return (double)r;
}
/* src/color.h:16-18 with the Windows `max` macro: (v > 0.0) ? v : 0.0, so NaN -> 0.0. */
static inline double clamp_lo0(double v) { return (v > 0.0) ? v : 0.0; }

/* src/color.h:27-29: uint8_t(std::clamp(v,0.0,1.0) * 255), truncating.  A NaN channel
 * (undefined behaviour in the reference) is defined as 0 here and on the device. */
static inline uint8_t to_u8(double v) {
    double c = (v < 0.0) ? 0.0 : ((1.0 < v) ? 1.0 : v);
    double s = c * 255;
    if (!(s == s)) return 0;
    return (uint8_t)s;
}

typedef struct {
    const rt_prim* prims; int n_prims;
    const rt_light* lights; int n_lights;
    double bgr[3];
    int max_depth;
    int prune;            /* RT_FLAG_PRUNE */
    /* instrumentation of the current pixel */
    uint32_t hit_mask[4], spawn_mask[4];
    int32_t hit_id;
    rt_counters* cnt;
} octx;

/* Plane::TestCollision src/objects.h:91-102, Sphere::TestCollision src/objects.h:124-138.
 * Returns 1 and *t when the reference returns a value, 0 for std::nullopt. */
static int test_collision(octx* c, const rt_prim* pr, v3 o, v3 d, double* t_out) {
    if (pr->type == RT_PRIM_PLANE) {
        if (c->cnt) c->cnt->plane_tests++;
        v3 n = v3_make(pr->q[0], pr->q[1], pr->q[2]);
        double denom = v3_dot(d, n);
        if (fabs(denom) > 1e-6) {
            v3 diff = v3_sub(v3_make(pr->p[0], pr->p[1], pr->p[2]), o);
            double t = v3_dot(diff, n) / denom;
            if (t > 0) { *t_out = t; return 1; }
        }
        return 0;
    } else {
        if (c->cnt) c->cnt->sphere_tests++;
        double radius2 = pr->q[0] * pr->q[0];           /* std::pow(radius_, 2) */
        v3 L = v3_sub(v3_make(pr->p[0], pr->p[1], pr->p[2]), o);
        double tca = v3_dot(L, d);
        if (tca < 0) return 0;
        double d2 = v3_dot(L, L) - tca * tca;           /* std::pow(tca, 2) */
        if (d2 > radius2) return 0;
        double thc = sqrt(radius2 - d2);
        double t = tca - thc;
        if (t < 0) t = tca + thc;
        *t_out = t;
        return 1;
    }
}

/* Plane::calcNormal src/objects.h:104-106, Sphere::calcNormal src/objects.h:140-142. */
static v3 calc_normal(const rt_prim* pr, v3 point) {
    if (pr->type == RT_PRIM_PLANE) return v3_unit(v3_make(pr->q[0], pr->q[1], pr->q[2]));
    return v3_unit(v3_sub(point, v3_make(pr->p[0], pr->p[1], pr->p[2])));
}

/* Object::Shade src/objects.h:20-65. */
static void shade(octx* c, int self, v3 ip, v3 normal, double res[3]) {
    const rt_prim* me = &c->prims[self];
    res[0] = res[1] = res[2] = 0.0;
    for (int li = 0; li < c->n_lights; ++li) {
        const rt_light* lt = &c->lights[li];
        v3 lpos = v3_make(lt->pos[0], lt->pos[1], lt->pos[2]);
        double occlusion = 1.0;
        if (c->cnt) c->cnt->feelers++;
        for (int oi = 0; oi < c->n_prims; ++oi) {
            if (oi == self) continue;
            /* Ray feeler{position{ip}, position{light->pos}}: src/common.h:27-31 */
            v3 fdir = v3_unit(v3_sub(lpos, ip));
            double p;
            if (test_collision(c, &c->prims[oi], ip, fdir, &p)) {
                v3 no = calc_normal(&c->prims[oi], v3_add(ip, v3_scale(fdir, p)));
                double facing = -v3_dot(no, fdir);
                double fres = mix_f(pow(1 - facing, 1), 0, (double)0.1f);
                occlusion = (1 - fres) * c->prims[oi].transparency;
                if (c->cnt) c->cnt->feelers_occluded++;
                break;                                   /* first in container order */
            }
        }
        v3 ldir = v3_unit(v3_sub(lpos, ip));
        double diff = fabs(v3_dot(normal, ldir) / (v3_len(normal) * v3_len(ldir)));
        double rd = (me->kd[0] * (lt->color[0] * lt->intensity) * diff) * occlusion;
        double gd = (me->kd[1] * (lt->color[1] * lt->intensity) * diff) * occlusion;
        double bd = (me->kd[2] * (lt->color[2] * lt->intensity) * diff) * occlusion;
        res[0] = clamp_lo0(res[0] + rd);
        res[2] = clamp_lo0(res[2] + bd);
        res[1] = clamp_lo0(res[1] + gd);
    }
}

/* TraceRay src/main.cc:19-113.  `node` is the heap number of this call (root 1,
 * reflection child 2*node, refraction child 2*node+1). */
static void trace_ray(octx* c, v3 o, v3 d, int depth, uint32_t node, double out[3]) {
    if (depth >= c->max_depth) {                         /* src/main.cc:31-34 */
        out[0] = c->bgr[0]; out[1] = c->bgr[1]; out[2] = c->bgr[2];
        return;
    }
    c->spawn_mask[node >> 5] |= 1u << (node & 31);
    if (c->cnt) c->cnt->rays++;

    int cur = -1;                                        /* src/main.cc:37-47 */
    double nearest = INFINITY;
    for (int i = 0; i < c->n_prims; ++i) {
        double t;
        if (test_collision(c, &c->prims[i], o, d, &t)) {
            if (t < nearest) { nearest = t; cur = i; }
        }
    }
    if (depth == 1) c->hit_id = cur;
    if (cur < 0) {                                       /* src/main.cc:49-53 */
        out[0] = c->bgr[0]; out[1] = c->bgr[1]; out[2] = c->bgr[2];
        return;
    }
    c->hit_mask[node >> 5] |= 1u << (node & 31);
    if (c->cnt) { c->cnt->hits++; if (depth == 1) c->cnt->primary_hits++; }

    const rt_prim* obj = &c->prims[cur];
    v3 ip = v3_add(o, v3_scale(d, nearest));             /* src/main.cc:55 */
    v3 n = calc_normal(obj, ip);
    double ri = obj->refraction_index;
    if (v3_dot(n, d) > 0) {                              /* src/main.cc:62-65 */
        n = v3_neg(n);
        ri = 1.0 / ri;
    }
    double surface[3];
    shade(c, cur, ip, n, surface);                       /* src/main.cc:68 */

    v3 vt = v3_div(d, fabs(v3_dot(d, n)));               /* src/main.cc:70 */

    /* RT_FLAG_PRUNE: a child under a material coefficient of exactly zero (as the device
     * holds it, in float) is not traced; what it would return is multiplied by zero below. */
    double refl[3] = {0.0, 0.0, 0.0};
    if (!(c->prune && (float)obj->reflectivity == 0.0f)) { /* src/main.cc:74-81 */
        v3 rd = v3_unit(v3_add(vt, v3_scale(n, 2.0)));
        v3 ro = v3_add(ip, v3_scale(n, 1e-4));
        if (c->cnt && depth + 1 < c->max_depth) c->cnt->reflections++;
        trace_ray(c, ro, v3_unit(rd), depth + 1, 2 * node, refl);
    }
    double trans[3] = {0.0, 0.0, 0.0};
    {                                                    /* src/main.cc:84-97 */
        double k = (ri * ri) * v3_len2(vt) - v3_len2(v3_add(vt, n));
        if (c->cnt && !(c->prune && (float)obj->transparency == 0.0f)) c->cnt->refraction_tests++;
        if (k > 0 && !(c->prune && (float)obj->transparency == 0.0f)) {
            v3 td = v3_sub(v3_scale(v3_add(n, vt), 1.0 / sqrt(k)), n);
            td = v3_unit(td);
            v3 to = v3_sub(ip, v3_scale(n, 1e-4));
            if (c->cnt) c->cnt->refractions++;
            trace_ray(c, to, v3_unit(td), depth + 1, 2 * node + 1, trans);
        }
    }
    double facing = -v3_dot(n, d);                       /* src/main.cc:100-105 */
    double fresnel = mix_f(pow(1 - facing, 3), 1, (double)0.1f);
    double kt = (1 - fresnel) * obj->transparency;
    double kr = (1 - kt) * obj->reflectivity;
    double kd = 1 - obj->transparency;
    double inv_depth = 1.0 / depth;
    for (int ch = 0; ch < 3; ++ch) {                     /* src/main.cc:107-112 */
        double sum = refl[ch] * kr + trans[ch] * kt;
        sum = sum + surface[ch] * kd;
        double amb = ((c->bgr[ch] * obj->ambient_intensity) * kd) * inv_depth;
        out[ch] = sum + amb;
    }
}

static int check_params(const rt_params* p) {
    if (!p) return RT_ERR_ARG;
    if (p->width < 1 || p->height < 1) return RT_ERR_ARG;
    if (p->max_depth < 2 || p->max_depth > RT_MAX_DEPTH) return RT_ERR_ARG;
    if (p->band_rows < 1 || p->band_first < 0 || p->band_step < 1 || p->band_count < 0)
        return RT_ERR_ARG;
    return RT_OK;
}

int32_t oracle_mask_words(int32_t max_depth) {
    int bits = 1 << (max_depth - 1);
    return bits < 32 ? 1 : bits / 32;
}

/* Rows of the partition after clipping to the frame (same rule as rt_strip_rows). */
int32_t oracle_strip_rows(const rt_params* p) {
    if (check_params(p) != RT_OK) return RT_ERR_ARG;
    int64_t rows = 0;
    for (int b = 0; b < p->band_count; ++b) {
        int64_t y0 = ((int64_t)p->band_first + (int64_t)b * p->band_step) * p->band_rows;
        if (y0 >= p->height) break;
        int64_t y1 = y0 + p->band_rows;
        if (y1 > p->height) y1 = p->height;
        rows += y1 - y0;
    }
    return (int32_t)rows;
}

/*
 * Tracer src/main.cc:115-141 for the rows of the partition, with the ViewFrustum
 * arithmetic of src/view_frustum.h:18,28-34,37-46 and the store of src/screen.h:55-73
 * (always the `<` branch: the depth buffer is 0 and z is 1, so the pixel is overwritten).
 * Same outputs, nullability and strip order as rt_render in include/rt_hip.h.
 */
int oracle_render(const rt_prim* prims, int32_t n_prims, const rt_light* lights,
                  int32_t n_lights, const rt_params* p, uint8_t* bgr, int32_t* hit_id,
                  uint32_t* hit_mask, uint32_t* spawn_mask, rt_counters* counters) {
    if (check_params(p) != RT_OK || !prims || n_prims < 1 || n_lights < 0 || !bgr ||
        (n_lights > 0 && !lights))
        return RT_ERR_ARG;
    const int W = p->width, H = p->height;
    const int words = oracle_mask_words(p->max_depth);

    const double TO_RAD = 3.141592653589793 / 180.;      /* src/view_frustum.h:18 */
    double aspect = (double)H / (double)W;               /* src/view_frustum.h:28 */
    double phy_w = 2 * 1 * tan(0.5 * p->fov_deg * TO_RAD);
    double phy_h = aspect * phy_w;
    double pw = phy_w / W;
    double ph = phy_h / H;

    octx c;
    memset(&c, 0, sizeof c);
    c.prims = prims; c.n_prims = n_prims; c.lights = lights; c.n_lights = n_lights;
    c.bgr[0] = p->bgr[0]; c.bgr[1] = p->bgr[1]; c.bgr[2] = p->bgr[2];
    c.max_depth = p->max_depth;
    rt_counters local;
    memset(&local, 0, sizeof local);
    c.cnt = ((p->flags & RT_FLAG_COUNTERS) && counters) ? &local : NULL;
    c.prune = (p->flags & RT_FLAG_PRUNE) != 0;

    int64_t row = 0;                                     /* row index in strip order */
    for (int b = 0; b < p->band_count; ++b) {
        int64_t y0 = ((int64_t)p->band_first + (int64_t)b * p->band_step) * p->band_rows;
        if (y0 >= H) break;
        int64_t y1 = y0 + p->band_rows;
        if (y1 > H) y1 = H;
        for (int64_t y = y0; y < y1; ++y, ++row) {
            for (int x = 0; x < W; ++x) {
                int xi = x - W / 2;                      /* src/main.cc:136-137 */
                int yi = (int)y - H / 2;
                /* GetRayByPixel: Ray(position{0,0,0}, position{(x+.5)pw,(y+.5)ph,1}) */
                v3 p1 = v3_make((xi + 0.5) * pw, (yi + 0.5) * ph, 1);
                v3 org = v3_make(0, 0, 0);
                v3 dir = v3_unit(v3_sub(p1, org));
                memset(c.hit_mask, 0, sizeof c.hit_mask);
                memset(c.spawn_mask, 0, sizeof c.spawn_mask);
                c.hit_id = -1;
                double col[3];
                trace_ray(&c, org, dir, 1, 1, col);
                size_t px = (size_t)row * W + x;
                bgr[3 * px + 0] = to_u8(col[2]);
                bgr[3 * px + 1] = to_u8(col[1]);
                bgr[3 * px + 2] = to_u8(col[0]);
                if (hit_id) hit_id[px] = c.hit_id;
                if (hit_mask) memcpy(hit_mask + px * words, c.hit_mask, 4 * words);
                if (spawn_mask) memcpy(spawn_mask + px * words, c.spawn_mask, 4 * words);
            }
        }
    }
    if (c.cnt) *counters = local;
    return RT_OK;
}

/* The occluder search of Object::Shade (src/objects.h:31-47) on its own, for the device
 * test hook rt_debug_occluder: for each of n points, the index of the first object in
 * container order (other than `self`, -1 = none is skipped) that the feeler towards
 * lights[light] collides with, or -1, and the occlusion factor (1.0 when unoccluded). */
int oracle_occluder(const rt_prim* prims, int32_t n_prims, const rt_light* lights, int32_t n_lights,
                    int32_t light, int32_t self, const double* points, int32_t n, int32_t* first_out,
                    double* factor_out) {
    if (!prims || !lights || light < 0 || light >= n_lights || self < -1 || self >= n_prims || !points || n < 0 ||
        !first_out || !factor_out)
        return RT_ERR_ARG;
    octx c;
    memset(&c, 0, sizeof c);
    v3 lpos = v3_make(lights[light].pos[0], lights[light].pos[1], lights[light].pos[2]);
    for (int i = 0; i < n; ++i) {
        v3 ip = v3_make(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
        first_out[i] = -1;
        factor_out[i] = 1.0;
        for (int oi = 0; oi < n_prims; ++oi) {
            if (oi == self) continue;
            v3 fdir = v3_unit(v3_sub(lpos, ip));
            double p;
            if (test_collision(&c, &prims[oi], ip, fdir, &p)) {
                v3 no = calc_normal(&prims[oi], v3_add(ip, v3_scale(fdir, p)));
                double facing = -v3_dot(no, fdir);
                double fres = mix_f(pow(1 - facing, 1), 0, (double)0.1f);
                first_out[i] = oi;
                factor_out[i] = (1 - fres) * prims[oi].transparency;
                break;
            }
        }
    }
    return RT_OK;
}

/* The nearest-hit loop of TraceRay (src/main.cc:37-47) on its own, for the device test hook
 * rt_debug_nearest: per ray (6 doubles: origin, direction used as given) the index of the
 * object hit first (-1 = none) and its t. */
int oracle_nearest(const rt_prim* prims, int32_t n_prims, const double* rays, int32_t n, int32_t* index_out,
                   double* t_out) {
    if (!prims || n_prims < 1 || !rays || n < 0 || !index_out || !t_out) return RT_ERR_ARG;
    octx c;
    memset(&c, 0, sizeof c);
    for (int i = 0; i < n; ++i) {
        v3 o = v3_make(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
        v3 d = v3_make(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        double nearest = INFINITY;
        int cur = -1;
        for (int oi = 0; oi < n_prims; ++oi) {
            double t;
            if (test_collision(&c, &prims[oi], o, d, &t)) {
                if (t < nearest) { nearest = t; cur = oi; }
            }
        }
        index_out[i] = cur;
        t_out[i] = (cur >= 0) ? nearest : 0.0;
    }
    return RT_OK;
}

/* Known-answer helpers for the device test hook (rt_debug_intersect): the collision
 * test of prims[prim] and gmtl's normalize on n caller-supplied rays (6 doubles each). */
int oracle_intersect(const rt_prim* prims, int32_t n_prims, const double* rays, int32_t n,
                     int32_t prim, double* t_out, uint8_t* hit_out, double* unit_out) {
    if (!prims || prim < 0 || prim >= n_prims || !rays || n < 0 || !t_out || !hit_out)
        return RT_ERR_ARG;
    octx c;
    memset(&c, 0, sizeof c);
    for (int i = 0; i < n; ++i) {
        v3 o = v3_make(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
        v3 d = v3_make(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        double t = 0.0;
        hit_out[i] = (uint8_t)test_collision(&c, &prims[prim], o, d, &t);
        t_out[i] = hit_out[i] ? t : 0.0;
        if (unit_out) {
            v3 u = v3_unit(d);
            unit_out[3 * i] = u.x; unit_out[3 * i + 1] = u.y; unit_out[3 * i + 2] = u.z;
        }
    }
    return RT_OK;
}
