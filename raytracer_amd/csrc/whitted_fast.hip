// Fast pass of the Whitted tracer (gfx950, wave64) with certified decisions.
//
// The reference's arithmetic is IEEE double and its outputs depend on exact comparisons
// (hit or miss, which object is nearest, inside or outside, total reflection or not,
// first occluder of a shadow feeler).  Reproducing its doubles bit for bit costs ~300 SIMD
// cycles per vector normalisation (correctly rounded sqrt + three divisions).  This
// kernel walks the same ray tree with two cheaper kinds of arithmetic:
//
//   * the RAY CHAIN (hit point, normal, reflected / refracted direction, child origins)
//     in double with fused multiply-adds and v_rsq_f64 / v_rcp_f64 + Newton steps: the
//     same values as the reference's to ~1e-15, at a fraction of the cost;
//   * every COLLISION TEST and all of the shading in float, on the rounded ray.
//
// Each ray carries bounds eo, ed on the distance of its origin and direction from the
// reference's (a few 1e-15, growing slowly along the path).  Every comparison the
// reference makes is evaluated with a bound on the error of the compared quantity and
// accepted only when the margin exceeds the bound, so an accepted decision is the
// reference's decision.  A float test that cannot be certified is repeated in double; the
// winner of the nearest-hit loop is re-evaluated in double for an accurate hit point.
// A pixel with a comparison that not even double certifies is abandoned and, at the end
// of its tile, evaluated again with the reference's exact arithmetic (whitted_exact.inc.h).
//
// Geometry uses formulas algebraically equal to the reference's but well conditioned:
//   reflection    normalize(v' + 2n), v' = d/|d.n|        ==  d + 2c n            (c = |d.n|)
//   refraction    k = ri^2 |v'|^2 - |v'+n|^2 > 0           <=> ri^2 - 1 + c^2 > 0
//                 normalize((n+v')/sqrt(k) - n)            ==  (d + (c - s) n)/ri, s = sqrt(ri^2-1+c^2)
//   sphere        d2 = L.L - tca^2                         ==  |L - tca d|^2
//   sphere normal normalize(ip - centre)                   ==  (ip - centre)/r up to |ip-centre| - r
// (src/main.cc:70-91, src/objects.h:126-142).
//
// Error model.  U = 2^-24 and V = 4e-15 are the relative errors of one float / one fast
// double operation, both multiplied by the `guard` factor.  |o - o*| <= eo, |d - d*| <= ed
// (Euclidean, starred = reference).  |v|_1 >= |v|_2 serves as a cheap over-estimate of
// magnitudes.  Derivations are next to each use.
#include "whitted_exact.inc.h"

// Fused multiply-adds are welcome here: every result is compared with an error bound that
// already covers one rounding per operation, and fusing only removes roundings.
#pragma clang fp contract(fast)

namespace rt {

#define RT_CONST __attribute__((address_space(4)))
typedef const RT_CONST float* cfdata_t;

// ---------------------------------------------------------------- float helpers
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(F3 a, F3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 madd(F3 a, F3 b, float s) {  // a + b*s
    return f3(__builtin_fmaf(b.x, s, a.x), __builtin_fmaf(b.y, s, a.y), __builtin_fmaf(b.z, s, a.z));
}
__device__ __forceinline__ float l1(F3 a) { return __builtin_fabsf(a.x) + __builtin_fabsf(a.y) + __builtin_fabsf(a.z); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }

// ---------------------------------------------------------------- fast double helpers
struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 operator-(D3 a) { return d3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(D3 a, D3 b) { return __builtin_fma(a.z, b.z, __builtin_fma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ D3 madd(D3 a, D3 b, double s) {
    return d3(__builtin_fma(b.x, s, a.x), __builtin_fma(b.y, s, a.y), __builtin_fma(b.z, s, a.z));
}
__device__ __forceinline__ F3 to_f3(D3 a) { return f3((float)a.x, (float)a.y, (float)a.z); }
__device__ __forceinline__ float l1f(D3 a) {
    return (float)(__builtin_fabs(a.x) + __builtin_fabs(a.y) + __builtin_fabs(a.z)) * 1.0000002f;
}
// 1/sqrt(x) and 1/x to ~1e-15: hardware estimate + two Goldschmidt / Newton steps.
__device__ __forceinline__ double rsq_d(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-g, h, 0.5);
    h = __builtin_fma(h, r, h);
    return h + h;
}
__device__ __forceinline__ double rcp_d(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

// One float primitive record (see FGeom): v[3] < 0 marks a plane.
struct FRec {
    float a0, a1, a2, r2;   // centre | normal, r2 | -1
    float b0, b1, tr;       // sphere: cerr, 0 ; plane: h, herr ; transparency
};
// Wave-uniform index: scalar loads, the record lives in SGPRs.
__device__ __forceinline__ FRec load_rec(cfdata_t p) {
    FRec g;
    g.a0 = p[0]; g.a1 = p[1]; g.a2 = p[2]; g.r2 = p[3];
    g.b0 = p[4]; g.b1 = p[5]; g.tr = p[6];
    return g;
}
// Per-lane index: two 16-byte vector loads.
__device__ __forceinline__ FRec gather_rec(const FGeom* tab, int i) {
    const float* p = tab[i].v;
    FRec g;
    g.a0 = p[0]; g.a1 = p[1]; g.a2 = p[2]; g.r2 = p[3];
    g.b0 = p[4]; g.b1 = p[5]; g.tr = p[6];
    return g;
}

// kUnsure + 4*site tells the diagnostics which comparison failed (sites 0..3).
enum : int { kMiss = 0, kHit = 1, kUnsure = 2 };

// A float ray with its error radii and the per-ray constants of the certain-miss filter
// (they use scene-wide magnitudes, so they are computed once per ray).
struct RayF {
    F3 o, d;
    float eo, ed;
    float ed3;    // ed + 3U
    float etb;    // bound on the error of any tca of this ray
    float far2;   // 2.01 * (bound on the error of any perpendicular vector)^2
    float en_ray; // eo + 5U |o|_1: the ray's part of the error of a plane numerator
    float eden;   // ed + 4U: error of a plane denominator
};

__device__ __forceinline__ void prepare_ray(RayF& r, float U, float scene_p1max) {
    r.ed3 = r.ed + 3.0f * U;
    const float o1 = l1(r.o);
    const float Sb = o1 + scene_p1max;               // >= |centre - o|_1 for every sphere
    const float eLb = r.eo + U * scene_p1max + U * Sb;
    r.etb = __builtin_fmaf(Sb, r.ed3, eLb);
    const float ePb = __builtin_fmaf(2.0f * U, Sb, 2.0f * r.etb);
    r.far2 = 2.01f * ePb * ePb;
    r.en_ray = __builtin_fmaf(5.0f * U, o1, r.eo);
    r.eden = r.ed + 4.0f * U;
}

// The certain-miss filter: true when the reference certainly finds no hit.
//   sphere: behind the origin by more than any tca error, or |P| > r + eP (which follows
//           from d2 > 2 r2 + 2 eP^2), P = the part of centre - o perpendicular to d;
//   plane:  numerator and denominator certainly of opposite signs (t* < 0; if instead
//           |denom*| <= 1e-6 the reference misses as well).
// About 15 operations, no division, no square root, no branch besides the uniform type.
__device__ __forceinline__ bool certain_miss(const FRec& g, const RayF& r) {
    if (__float_as_int(g.r2) < 0) {  // plane (wave-uniform: integer test of the sign bit)
        const F3 n = f3(g.a0, g.a1, g.a2);
        const float denom = dot(r.d, n);
        const float num = g.b0 - dot(r.o, n);
        const float en = r.en_ray + g.b1;
        return (num < -en && denom > r.eden) || (num > en && denom < -r.eden);
    }
    const F3 L = f3(g.a0, g.a1, g.a2) - r.o;
    const float tca = dot(L, r.d);
    const F3 P = madd(L, r.d, -tca);
    const float d2 = dot(P, P);
    return tca < -r.etb || __builtin_fmaf(-2.01f, g.r2, d2) > r.far2;
}

// Full certified float test of a per-lane record; straight-line, the outcome is selected
// at the end.  Sphere (src/objects.h:124-138):
//   L = c - o:   |L - L*| <= eo + |c~ - c| + U|L|  =: eL            (c~ = float(c))
//   tca = L.d:   |tca - tca*| <= eL + |L| (ed + 3U)   =: et          (|d*| = 1)
//   P = L - tca d: |P - P*| <= eL + et + |tca| ed + 2U(|L| + |tca|) <= 2 et + 2U|L| =: eP
//   d2 = |P|^2:  |d2 - d2*| <= (2|P| + eP) eP + 3U d2 + U r2 + (reference's own rounding
//                of L.L - tca^2, <= 4 * 2^-53 |L|^2)                 =: e2
//   thc = sqrt(r2 - d2), m = r2 - d2 > e2:  |thc - thc*| <= e2/thc + 2U thc
// Plane (src/objects.h:91-102):
//   denom = d.n:  |denom - denom*| <= ed + 4U =: eden                 (n~ = float(n))
//   num = h - o.n: |num - num*| <= eo + 5U|o| + herr =: en            (h = float(p0.n))
//   t = num/denom: |t - t*| <= (en + |t| eden)/(|denom| - eden) + 4U|t|
__device__ __forceinline__ int test_rec(const FRec& g, const RayF& r, float U, float& t, float& te) {
    const F3 a = f3(g.a0, g.a1, g.a2);
    // sphere
    const F3 L = a - r.o;
    const float tca = dot(L, r.d);
    const F3 P = madd(L, r.d, -tca);
    const float d2 = dot(P, P);
    const float S1 = l1(L);
    const float eL = r.eo + g.b0 + U * S1;
    const float et = __builtin_fmaf(S1, r.ed3, eL);
    const float eP = __builtin_fmaf(2.0f * U, S1, 2.0f * et);
    const float e2 = __builtin_fmaf(2.0f * l1(P) + eP, eP, 6.0f * U * (d2 + g.r2)) + 1e-14f * S1 * S1;
    const float m = g.r2 - d2;
    const bool s_miss = (m < -e2) | (tca < -et);
    const bool s_through = (m > e2) & (tca > et);
    const float thc = sqrt_(__builtin_fmaxf(m, 1.0e-30f));
    const float tm = tca - thc;
    const float s_te = et + e2 * rcp(thc) * 1.01f + U * (__builtin_fabsf(tca) + 3.0f * thc);
    const bool near_root = tm > s_te, far_root = tm < -s_te;
    const float s_t = near_root ? tm : tca + thc;
    const int s_st = s_miss ? kMiss : (s_through ? ((near_root | far_root) ? kHit : kUnsure + 4) : kUnsure + 0);
    // plane
    const float denom = dot(r.d, a);
    const float num = g.b0 - dot(r.o, a);
    const float en = r.en_ray + g.b1;
    const float ad = __builtin_fabsf(denom);
    const float p_t = num * rcp(denom);
    const float at = __builtin_fabsf(p_t);
    const float p_te = __builtin_fmaf(at, r.eden, en) * rcp(__builtin_fmaxf(ad - r.eden, 1.0e-30f)) * 1.01f + 4.0f * U * at;
    const int p_tt = (p_t > p_te) ? kHit : ((p_t < -p_te) ? kMiss : kUnsure + 12);
    const int p_st = (ad + r.eden < 0.999999e-6f) ? kMiss : ((ad > 1.000001e-6f + r.eden) ? p_tt : kUnsure + 8);
    const bool pl = __float_as_int(g.r2) < 0;
    t = pl ? p_t : s_t;
    te = pl ? p_te : s_te;
    return pl ? p_st : s_st;
}

// Pass 1 of an object loop: the certain-miss filter over ALL primitives (wave-uniform
// index, record in SGPRs, prefetched one ahead) keeping, per lane, how many survived and
// the indices of the first two, in container order.  `skip` is a per-lane index treated
// as a certain miss (the plane a ray was spawned on / the object a feeler starts on).
struct Survivors {
    int cnt, c0, c1;
};
__device__ __forceinline__ Survivors filter_prims(cfdata_t geom, int n, const RayF& r, int skip, bool active) {
    Survivors s;
    s.cnt = 0;
    s.c0 = 0;
    s.c1 = 0;
    cfdata_t p = geom;
    FRec gn = load_rec(p);
    for (int i = 0; i < n; ++i) {
        const FRec g = gn;
        p += 8;
        gn = load_rec(p);  // the table has one record of padding
        const bool keep = active & (i != skip) & !certain_miss(g, r);
        s.c0 = (keep & (s.cnt == 0)) ? i : s.c0;
        s.c1 = (keep & (s.cnt == 1)) ? i : s.c1;
        s.cnt += keep ? 1 : 0;
    }
    return s;
}

// The same two tests on the double ray: for the rare float test that could not be
// certified, and for the accurate t of the nearest hit.  Same derivations with V in place
// of U; the primitive's double data is the reference's, so the |c~ - c| terms vanish.
struct RayD {
    D3 o, d;
    float eo, ed;
};
__device__ __forceinline__ int test_prim_d(const DGeom* gp, const RayD& r, float V, double& t, float& te) {
    const D3 p = d3(gp->p[0], gp->p[1], gp->p[2]);
    const double r2 = gp->r2;
    if (r2 < 0.0) {
        const D3 q = d3(gp->q[0], gp->q[1], gp->q[2]);
        const double denom = dot(r.d, q);
        const double ad = __builtin_fabs(denom);
        const float eden = r.ed + 4.0f * V;
        if (ad + (double)eden < 0.99999999e-6) return kMiss;
        if (!(ad > 1.00000001e-6 + (double)eden)) return kUnsure + 8;
        const D3 diff = p - r.o;
        const double num = dot(diff, q);
        const float en = r.eo + 5.0f * V * l1f(diff);
        const double tt = num * rcp_d(denom);
        const float at = (float)__builtin_fabs(tt) * 1.0000002f;
        const float ett = __builtin_fmaf(at, eden, en) * rcp((float)ad - eden) * 1.01f + 8.0f * V * at;
        if (tt > (double)ett) {
            t = tt;
            te = ett;
            return kHit;
        }
        if (tt < -(double)ett) return kMiss;
        return kUnsure + 12;
    }
    const D3 L = p - r.o;
    const double tca = dot(L, r.d);
    const D3 P = madd(L, r.d, -tca);
    const double d2 = dot(P, P);
    const float S1 = l1f(L);
    const float eL = r.eo + V * S1;
    const float et = __builtin_fmaf(S1, r.ed + 3.0f * V, eL);
    const float eP = __builtin_fmaf(2.0f * V, S1, 2.0f * et);
    const float P1 = l1f(P);
    const float e2 = __builtin_fmaf(2.0f * P1 + eP, eP, 6.0f * V * (float)(d2 + r2)) + 1e-14f * S1 * S1;
    const double m = r2 - d2;
    if (m < -(double)e2 || tca < -(double)et) return kMiss;
    if (!(m > (double)e2 && tca > (double)et)) return kUnsure + 0;
    const double thc = m * rsq_d(m);
    const float thf = (float)thc;
    const float eh = e2 * rcp(thf) * 1.01f + 4.0f * V * thf;
    const double tm = tca - thc;
    const float ett = et + eh + 2.0f * V * ((float)__builtin_fabs(tca) + thf);
    if (tm > (double)ett) t = tm;
    else if (tm < -(double)ett) t = tca + thc;
    else return kUnsure + 4;
    te = ett;
    return kHit;
}

template <int MW>
__device__ __forceinline__ void set_bit_f(uint32_t (&m)[MW > 0 ? MW : 1], uint32_t node) {
#pragma unroll
    for (int w = 0; w < MW; ++w) m[w] |= ((node >> 5) == (uint32_t)w) ? (1u << (node & 31)) : 0u;
}

__device__ __forceinline__ uint32_t to_u8f(float v) {
    float c = __builtin_fminf(__builtin_fmaxf(v, 0.0f), 1.0f);  // NaN -> 0
    return (uint32_t)(int)(c * 255.0f);
}

__device__ __forceinline__ double pack2(float lo, float hi) {
    return __hiloint2double(__float_as_int(hi), __float_as_int(lo));
}

// Diagnostic build only (-DRT_STAMPS): cycles per wave spent in each segment of the walk,
// summed into a.stamps[segment] (and [8+segment] = visits).  Never part of the product.
#ifdef RT_STAMPS
#define RT_STAMP(seg)                                                                  \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        unsigned long long now__;                                                      \
        __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now__)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                             \
        seg_cycles[seg] += now__ - last__;                                             \
        seg_visits[seg] += 1;                                                          \
        last__ = now__;                                                                \
    } while (0)
#else
#define RT_STAMP(seg) do { } while (0)
#endif

// Persistent waves: the grid is a fixed number of blocks per CU, every wave walks the 8x8
// pixel tiles of the strip with a stride of the number of waves in the grid.
template <int MW>
__global__ __launch_bounds__(64 * kWavesPerBlock) void whitted_fast_kernel(FArgs a) {
#ifdef RT_STAMPS
    unsigned long long seg_cycles[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long seg_visits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last__;
    __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last__)::"memory");
#endif
    extern __shared__ double lds_fstack[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int entries = (a.max_depth > 2 ? a.max_depth - 1 : 1);
    double* stk = lds_fstack + (size_t)wave * entries * kFastStackFields * 64 + lane;

    const cfdata_t geom = (cfdata_t)(const float*)a.geom;
    const cfdata_t lights = (cfdata_t)(const float*)a.lights;
    const float U = 6.0e-8f * a.guard;
    const float V = 4.0e-15f * a.guard;
    const F3 bgr = f3(a.bgr_c[0], a.bgr_c[1], a.bgr_c[2]);
    const int n_tiles = a.tiles_x * a.tiles_y;
    const int n_waves = gridDim.x * kWavesPerBlock;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.stats[1], 1ull);

    for (int tile = blockIdx.x * kWavesPerBlock + wave; tile < n_tiles; tile += n_waves) {
    const int ty = tile / a.tiles_x;
    const int tx = tile - ty * a.tiles_x;
    const int x = tx * kTile + (lane & 7);
    const int k = ty * kTile + (lane >> 3);
    const bool inside = (x < a.width) && (k < a.rows);
    bool have = inside;
    bool flagged = false;
    uint32_t reason = 0;
    int y;
    if (a.band_step == 1) {
        y = a.band_first * a.band_rows + k;              // consecutive bands: one block of rows
    } else {
        const int band = k / a.band_rows;
        y = (a.band_first + band * a.band_step) * a.band_rows + (k - band * a.band_rows);
    }

    // Primary ray (src/view_frustum.h:37-46, src/common.h:27-31) in fast double: the
    // products are the reference's, the normalisation differs by a few 1e-16.
    RayD ray;
    ray.o = d3(0.0, 0.0, 0.0);
    {
        const int xi = x - a.width / 2;
        const int yi = y - a.height / 2;
        const D3 p1 = d3((xi + 0.5) * a.pw64, (yi + 0.5) * a.ph64, 1.0);
        ray.d = p1 * rsq_d(dot(p1, p1));
    }
    ray.eo = 0.0f;
    ray.ed = 4.0f * V;
    float w = 1.0f;
    uint32_t node = 1;
    int sp = 0;
    // Plane this ray was spawned on (-1: none).  Such a ray starts b = 1e-4 off the plane on
    // the side it moves towards: for the reflection child num* = (p0 - o*).q = -b and
    // denom* = rd*.q = +c* (both signs swap together when the normal was flipped), for the
    // refraction child num* = +b and denom* = -s*/ri; the reference therefore finds t* < 0,
    // or |denom*| <= 1e-6: a miss either way, whatever the ray's error bounds are.  Holds
    // while the rounding of the reference's own hit point (2^-52 |ip|) stays far below b.
    int self_plane = -1;

    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    int32_t first_hit = -1;
    uint32_t hmask[MW > 0 ? MW : 1] = {0};
    uint32_t smask[MW > 0 ? MW : 1] = {0};

    RT_STAMP(0);
    while (have) {
        const int level = 32 - __builtin_clz(node);
        if (MW > 0) set_bit_f<MW>(smask, node);
        int why = 0;  // first comparison that could not be certified (diagnostics)
        bool bad = false;

        // the float image of the ray: rounding adds U|o| and U to the chain bounds
        RayF rf;
        rf.o = to_f3(ray.o);
        rf.d = to_f3(ray.d);
        rf.eo = ray.eo + U * l1(rf.o);
        rf.ed = ray.ed + 1.01f * U;
        prepare_ray(rf, U, a.scene_p1max);

        // ---- nearest hit (src/main.cc:37-47): the winner must beat every other candidate
        // by more than both error bounds.  Pass 1: certain-miss filter over all primitives.
        // Pass 2: the full certified test on each lane's own survivors (mostly none or one).
        int cur = -1;
        float best_t = __builtin_inff(), best_e = 0.0f, second_lo = __builtin_inff();
        const Survivors sv = filter_prims(geom, a.n_prims, rf, self_plane, true);
        auto consider = [&](int i, bool on) {
            if (!__any(on)) return;
            const FRec g = gather_rec(a.geom, on ? i : 0);
            float t, te;
            int st = test_rec(g, rf, U, t, te);
            st = on ? st : kMiss;
            if (__any(st >= kUnsure)) {
                if (st >= kUnsure) {
                    // repeat this one test in double
                    double t_d;
                    st = test_prim_d(a.geom64 + i, ray, V, t_d, te);
                    t = (float)t_d;
                    te += U * t;
                    if (st >= kUnsure) { bad = true; if (!why) why = 6 + (st >> 2); }
                }
            }
            // keep the smallest t and the lowest possible t of everything else
            const bool hit = st == kHit;
            const bool better = hit & (t < best_t);
            const float other_lo = better ? (best_t - best_e) : (hit ? (t - te) : __builtin_inff());
            second_lo = __builtin_fminf(second_lo, other_lo);
            best_t = better ? t : best_t;
            best_e = better ? te : best_e;
            cur = better ? i : cur;
        };
        consider(sv.c0, sv.cnt > 0);
        consider(sv.c1, sv.cnt > 1);
        if (__any(sv.cnt > 2)) {
            // more than two survivors in some lane: those lanes walk the rest of the list
            for (int i = 0; i < a.n_prims; ++i) {
                const FRec g = load_rec(geom + 8 * (size_t)i);
                const bool on = (sv.cnt > 2) & (i > sv.c1) & (i != self_plane) & !certain_miss(g, rf);
                consider(i, on);
            }
        }
        if (!(second_lo > best_t + best_e) && cur >= 0) { bad = true; if (!why) why = 2; }
        if (MW > 0 && level == 1) first_hit = cur;
        RT_STAMP(1);

        if (cur < 0) {
            cr = __builtin_fmaf(w, bgr.x, cr);
            cg = __builtin_fmaf(w, bgr.y, cg);
            cb = __builtin_fmaf(w, bgr.z, cb);
        } else {
            if (MW > 0) set_bit_f<MW>(hmask, node);
            // per-lane gather of the hit primitive (double) and its material
            const DGeom* gs = a.geom64 + cur;
            const FMat* ms = a.mat + cur;
            const F3 kd = f3(ms->kd[0], ms->kd[1], ms->kd[2]);
            const float m_refl = ms->reflectivity, m_tr = ms->transparency, m_amb = ms->ambient;
            const double m_ri = a.mat64[cur].refraction_index;
            const double m_inv_ri = a.mat64[cur].pad;             // 1/ri, the reference's 1.0/ri to 1e-16
            const bool is_plane = gs->r2 < 0.0;

            // accurate t of the winner, in double
            double t;
            float te;
            if (test_prim_d(gs, ray, V, t, te) != kHit) {
                bad = true;
                if (!why) why = 1;
                t = (double)best_t;
                te = best_e;
            }

            // hit point (src/main.cc:55): |ip - ip*| <= eo + t ed + te + V|ip|  =: ep
            const D3 ip = madd(ray.o, ray.d, t);
            const F3 ipf = to_f3(ip);
            const float ip1 = l1(ipf);
            const float tf = (float)t * 1.0000002f;
            const float ep = ray.eo + tf * ray.ed + te + V * (ip1 + tf);
            // normal (src/objects.h:104-106,140-142).  Sphere: the reference normalises
            // ip* - centre, whose length is r up to the rounding of its own hit point,
            // off <= 1e-15 (|L|^2/r + |L| + |ip|); (ip - centre)/r differs from that unit
            // vector by at most (ep + off)/r.
            D3 n;
            float en;
            if (is_plane) {
                n = d3(gs->q[0], gs->q[1], gs->q[2]);
                en = 4.0f * V;
            } else {
                const double inv_r = gs->q[0];   // device table: q = {1/r, r, 0} for spheres
                n = (ip - d3(gs->p[0], gs->p[1], gs->p[2])) * inv_r;
                const float reach = tf + 2.0f * (float)gs->q[1];
                const float off = 1.0e-15f * (reach * reach * (float)inv_r + reach + ip1);
                en = (1.01f * ep + off) * (float)inv_r * 1.01f + 4.0f * V;
            }
            // inside/outside (src/main.cc:62-65)
            const double cosd = dot(n, ray.d);
            const float ec = en + ray.ed + 4.0f * V;
            if (!((float)__builtin_fabs(cosd) > ec * 1.01f)) { bad = true; if (!why) why = 3; }
            const bool flip = cosd > 0.0;
            if (flip) n = -n;
            const double c = __builtin_fabs(cosd);
            const float cf = (float)c;
            const double ri = flip ? m_inv_ri : m_ri;             // refraction_index or its reciprocal
            const double inv_ri = flip ? m_ri : m_inv_ri;
            const float rif = (float)ri;

            // weights (src/main.cc:100-105); facing_ratio = c
            const float x1 = 1.0f - cf;
            const float fresnel = __builtin_fmaf(x1 * x1 * x1, 0.9f, 0.1f);       // mix(x^3, 1, .1f)
            const float kt = (1.0f - fresnel) * m_tr;
            const float kr = (1.0f - kt) * m_refl;
            const float kdm = 1.0f - m_tr;
            const float amb = m_amb * kdm * a.inv_depth[level];

            // total internal reflection test (src/main.cc:86-87): k > 0 <=> kq > 0 with
            // kq = ri^2 - 1 + c^2;  |kq - kq*| <= (2c + ec) ec + 4V(ri^2 + 1) + 1e-13
            const double kq = __builtin_fma(ri, ri, -1.0) + c * c;
            const float ek = __builtin_fmaf(2.0f * cf + ec, ec, 4.0f * V * (rif * rif + 1.0f)) + 1e-13f;
            if (!((float)__builtin_fabs(kq) > ek * 1.01f)) { bad = true; if (!why) why = 5; }
            const bool refr = kq > 0.0;
            const float wr = w * kr, wt = w * kt;

            // ---- children first (they go to the stack, so that only floats stay live
            // across the shading loops) ----
            if (level + 1 >= a.max_depth) {
                const float ws = refr ? (wr + wt) : wr;
                cr = __builtin_fmaf(ws, bgr.x, cr);
                cg = __builtin_fmaf(ws, bgr.y, cg);
                cb = __builtin_fmaf(ws, bgr.z, cb);
            } else if (!bad) {
                // child origins (src/main.cc:77-78,90-91): ip +- 1e-4 n
                const float eoc = ep + 1.0e-4f * en + V * (ip1 + 1.0e-4f);
                const int child_self = (is_plane && ip1 < 1.0e8f) ? cur : -1;
                const float self_bits = __int_as_float(0);
                (void)self_bits;
                if (refr) {
                    // refraction child (src/main.cc:88-93): td = (d + (c - s) n)/ri,
                    // s = sqrt(kq): |s - s*| <= ek/s + 2V s;
                    // |td - td*| <= (ed + ec + es + |c - s| en)/ri + 8V
                    const double s = kq * rsq_d(kq);
                    const float sf = (float)s;
                    const float es = ek * rcp(sf) * 1.01f + 2.0f * V * sf;
                    const double cs = c - s;
                    const D3 td = madd(ray.d, n, cs) * inv_ri;
                    const float etd = (ray.ed + ec + es + (float)__builtin_fabs(cs) * 1.0000002f * en) * (float)inv_ri * 1.01f + 8.0f * V;
                    const D3 to = madd(ip, n, -1.0e-4);
                    double* e = stk + (size_t)sp * kFastStackFields * 64;
                    e[0 * 64] = to.x; e[1 * 64] = to.y; e[2 * 64] = to.z;
                    e[3 * 64] = td.x; e[4 * 64] = td.y; e[5 * 64] = td.z;
                    e[6 * 64] = pack2(wt, eoc);
                    e[7 * 64] = pack2(etd, __int_as_float((int)((2u * node + 1u) | ((uint32_t)(child_self + 1) << 8))));
                    ++sp;
                }
                {
                    // reflection child (src/main.cc:75-79): rd = d + 2c n,
                    // |rd - rd*| <= ed + 2 ec + 2 c en + 8V
                    const float erd = ray.ed + 2.0f * ec + 2.0f * cf * en + 8.0f * V;
                    const D3 ro = madd(ip, n, 1.0e-4);
                    const D3 rd = madd(ray.d, n, 2.0 * c);
                    double* e = stk + (size_t)sp * kFastStackFields * 64;
                    e[0 * 64] = ro.x; e[1 * 64] = ro.y; e[2 * 64] = ro.z;
                    e[3 * 64] = rd.x; e[4 * 64] = rd.y; e[5 * 64] = rd.z;
                    e[6 * 64] = pack2(wr, eoc);
                    e[7 * 64] = pack2(erd, __int_as_float((int)((2u * node) | ((uint32_t)(child_self + 1) << 8))));
                    ++sp;
                }
            }
            RT_STAMP(2);

            // ---- Object::Shade (src/objects.h:20-65), in float from the rounded hit ----
            const F3 nf = to_f3(n);
            const float epf = ep + U * ip1;
            float sr = 0.0f, sg = 0.0f, sb = 0.0f;
            for (int li = 0; li < a.n_lights; ++li) {
                const cfdata_t lr = lights + (size_t)li * 8;
                const F3 lpos = f3(lr[0], lr[1], lr[2]);
                const float lerr = lr[3];
                const F3 lci = f3(lr[4], lr[5], lr[6]);
                // feeler: origin ip, direction normalize(lpos - ip); the un-normalised
                // vectors differ by <= epf + lerr + U|lv|, normalising doubles that at most
                const F3 lv = lpos - ipf;
                const float linv = rsq(dot(lv, lv));
                RayF fr;
                fr.o = ipf;
                fr.d = lv * linv;
                fr.eo = epf;
                fr.ed = 2.02f * (epf + lerr + U * l1(lv)) * linv + 4.0f * U;
                prepare_ray(fr, U, a.scene_p1max);
                float occl = 1.0f;
                bool open = !bad;
                // first occluder in container order (src/objects.h:35-47): filter, then the
                // full test on each lane's survivors in order until one is a certified hit
                const Survivors fs = filter_prims(geom, a.n_prims, fr, cur, open);
                auto occluder = [&](int oi, bool on) {
                    on = on & open;
                    if (!__any(on)) return;
                    const FRec g = gather_rec(a.geom, on ? oi : 0);
                    float p, pe;
                    int st = test_rec(g, fr, U, p, pe);
                    st = on ? st : kMiss;
                    if (__any(st >= kUnsure)) {
                        if (st >= kUnsure) {
                            // repeat in double with the double light; the hit point's double
                            // digits are gone by now: use the float one, its error is in ep
                            const DLight* ld = a.lights64 + li;
                            const D3 ipd = d3((double)ipf.x, (double)ipf.y, (double)ipf.z);
                            const D3 lvd = d3(ld->pos[0], ld->pos[1], ld->pos[2]) - ipd;
                            const double linvd = rsq_d(dot(lvd, lvd));
                            RayD frd;
                            frd.o = ipd;
                            frd.d = lvd * linvd;
                            frd.eo = epf;
                            frd.ed = 2.02f * (epf + V * l1f(lvd)) * (float)linvd + 4.0f * V;
                            double p_d;
                            st = test_prim_d(a.geom64 + oi, frd, V, p_d, pe);
                            p = (float)p_d;
                            if (st >= kUnsure) { bad = true; if (!why) why = 10 + (st >> 2); open = false; }
                        }
                    }
                    // occlusion factor of a hit (src/objects.h:40-46); colour only
                    const F3 hv = madd(ipf, fr.d, p) - f3(g.a0, g.a1, g.a2);
                    const F3 ns = hv * rsq(__builtin_fmaxf(dot(hv, hv), 1.0e-30f));
                    const bool pl = __float_as_int(g.r2) < 0;
                    const F3 no = f3(pl ? g.a0 : ns.x, pl ? g.a1 : ns.y, pl ? g.a2 : ns.z);
                    const float facing = -dot(no, fr.d);
                    const float fres = (1.0f - facing) * (1.0f - 0.1f);   // mix(x, 0, .1f)
                    const bool hit = st == kHit;
                    occl = hit ? (1.0f - fres) * g.tr : occl;
                    open = open & !hit;
                };
                occluder(fs.c0, fs.cnt > 0);
                occluder(fs.c1, fs.cnt > 1);
                if (__any(open & (fs.cnt > 2))) {
                    for (int oi = 0; oi < a.n_prims; ++oi) {
                        const FRec g = load_rec(geom + 8 * (size_t)oi);
                        const bool on = open & (fs.cnt > 2) & (oi > fs.c1) & (oi != cur) & !certain_miss(g, fr);
                        occluder(oi, on);
                    }
                }
                // diffuse term (src/objects.h:50-60); n and the light direction are unit
                const float diff = __builtin_fabsf(dot(nf, fr.d));
                const float f = diff * occl;
                sr = __builtin_fmaxf(__builtin_fmaf(kd.x * lci.x, f, sr), 0.0f);
                sg = __builtin_fmaxf(__builtin_fmaf(kd.y * lci.y, f, sg), 0.0f);
                sb = __builtin_fmaxf(__builtin_fmaf(kd.z * lci.z, f, sb), 0.0f);
            }
            RT_STAMP(3);
            // own colour: surface*kd + (bgr*ambient)*kd*(1/depth), src/main.cc:111-112
            cr = __builtin_fmaf(w, __builtin_fmaf(sr, kdm, bgr.x * amb), cr);
            cg = __builtin_fmaf(w, __builtin_fmaf(sg, kdm, bgr.y * amb), cg);
            cb = __builtin_fmaf(w, __builtin_fmaf(sb, kdm, bgr.z * amb), cb);
        }

        RT_STAMP(4);
        if (bad) {
            flagged = true;
            have = false;
            reason = (uint32_t)why | ((uint32_t)level << 4);
        } else if (sp > 0) {
            --sp;
            const double* e = stk + (size_t)sp * kFastStackFields * 64;
            ray.o = d3(e[0 * 64], e[1 * 64], e[2 * 64]);
            ray.d = d3(e[3 * 64], e[4 * 64], e[5 * 64]);
            const double p6 = e[6 * 64], p7 = e[7 * 64];
            w = __int_as_float(__double2loint(p6));
            ray.eo = __int_as_float(__double2hiint(p6));
            ray.ed = __int_as_float(__double2loint(p7));
            const uint32_t packed = (uint32_t)__double2hiint(p7);
            node = packed & 0xffu;
            self_plane = (int)(packed >> 8) - 1;
        } else {
            have = false;
        }
    }

    RT_STAMP(5);
    const uint32_t px = (uint32_t)k * (uint32_t)a.width + (uint32_t)x;
    if (inside && !flagged) {
        uint8_t* dst = a.bgr + 3 * (size_t)px;
        dst[0] = (uint8_t)to_u8f(cb);
        dst[1] = (uint8_t)to_u8f(cg);
        dst[2] = (uint8_t)to_u8f(cr);
        if (MW > 0) {
            if (a.hit_id) a.hit_id[px] = first_hit;
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                if (a.hit_mask) a.hit_mask[(size_t)px * MW + i] = hmask[i];
                if (a.spawn_mask) a.spawn_mask[(size_t)px * MW + i] = smask[i];
            }
        }
    }
    if (a.reason && inside) a.reason[px] = (uint8_t)reason;
    // The pixels this pass could not certify are evaluated right here with the reference's
    // exact double arithmetic (the other lanes of the wave wait); the LDS stack is free by now.
    const unsigned long long fm = __ballot(flagged);
    if (fm != 0ull) {
        if (lane == __builtin_ctzll(fm)) atomicAdd(&a.stats[0], (unsigned long long)__builtin_popcountll(fm));
        uint32_t unused_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        exact::trace_pixel<MW, false>(a.exact, x, k, flagged, stk, unused_counters);
    }
    RT_STAMP(6);
    }  // tiles
#ifdef RT_STAMPS
    if (lane == 0 && a.stamps) {
        unsigned long long* dst = a.stamps + 16ull * (blockIdx.x * kWavesPerBlock + wave);
        for (int i = 0; i < 8; ++i) {
            dst[i] = seg_cycles[i];
            dst[8 + i] = seg_visits[i];
        }
    }
#endif
}

template <int MW>
static hipError_t launch_f(const FArgs& a, hipStream_t stream) {
    const int n_tiles = a.tiles_x * a.tiles_y;
    if (n_tiles <= 0) return hipSuccess;
    int blocks = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > a.max_blocks) blocks = a.max_blocks;
    hipLaunchKernelGGL((whitted_fast_kernel<MW>), dim3(blocks), dim3(64 * kWavesPerBlock),
                       fast_stack_lds_bytes(a.max_depth), stream, a);
    return hipGetLastError();
}

hipError_t launch_whitted_fast(const FArgs& a, bool emit, hipStream_t stream) {
    switch (emit ? a.mask_words : 0) {
        case 0: return launch_f<0>(a, stream);
        case 1: return launch_f<1>(a, stream);
        case 2: return launch_f<2>(a, stream);
        case 4: return launch_f<4>(a, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace rt
