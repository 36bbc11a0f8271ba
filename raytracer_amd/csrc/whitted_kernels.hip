// Whitted trace kernels for MI355X (gfx950, wave64): every pixel evaluated with the
// reference's own double arithmetic (see whitted_exact.inc.h), one wave per 8x8 tile.
#include "whitted_exact.inc.h"

#pragma clang fp contract(off)

namespace rt {

using namespace exact;

// From this many lights on the container-order shadow loops use the late predicate
// (feeler_slot's LATE): measured on the default scene (2 lights, 2 % slower with it) and
// the glass scene (16 lights, 13 % faster).
constexpr int kLateMinLights = 4;

// Waves per SIMD the image-only kernel is compiled for (its register budget).
#ifndef RT_IMAGE_WAVES
#define RT_IMAGE_WAVES 5
#endif
// The same for the image-only kernels that use the anchor grids: every lane walks its own list
// of candidates there, which at 96 registers spills into private memory in the shadow loops
// (glass 4K depth 8: 6.5 GB of memory traffic per frame against 0.95 GB at 128 registers, for
// 1 % of the frame time).
#ifndef RT_GRID_WAVES
#define RT_GRID_WAVES 4
#endif
constexpr int image_waves(bool grid) { return grid ? RT_GRID_WAVES : RT_IMAGE_WAVES; }

template <bool COUNT, typename AT>
__device__ __forceinline__ void flush_counters(AT& a, const uint32_t (&cnt)[kCounters], int lane) {
    if (COUNT) {
        unsigned long long s[kCounters];
#pragma unroll
        for (int i = 0; i < kCounters; ++i) s[i] = wave_sum(cnt[i]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < kCounters; ++i) atomicAdd(&a.counters[i], s[i]);
        }
    }
}

// Region `w` (a wave of the launch) of the global stack workspace, lane-major like the LDS part.
// (w must be a wave-uniform value: the result then stays in scalar registers.)
template <typename AT>
__device__ __forceinline__ double* wave_gstack(AT& a, size_t w) {
    return a.gstack + w * (kDeepRegionBytes / sizeof(double));
}

// Without the deep stack the grid holds one wave per 8x8 tile of the strip; with it the grid
// is fixed and its waves take pixels from the tile queue (TileQueue).
template <int MW, bool COUNT, bool TREE, bool DEEP, bool LATE, bool GRID>
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu((MW == 0 && !COUNT) ? image_waves(GRID) : 3, (MW == 0 && !COUNT && image_waves(GRID) > 4) ? image_waves(GRID) : 4))) void whitted_kernel(KArgs a_, SkyTable) {
    extern __shared__ double lds_stack[];
    const KArgs& a = a_;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* stk = lds_stack + (size_t)wave * (DEEP ? stack_lds_entries(a.max_depth) : tile_stack_lds_entries(a.max_depth)) * kStackFields * 64 + lane;
    uint32_t cnt[kCounters] = {0};
    if (DEEP || TREE) {
        double* gstk = DEEP ? wave_gstack(a, (size_t)blockIdx.x * kWavesPerBlock + wave) : nullptr;
        // (launches that share rays between lanes: the wave's colour sums and exchange words)
        const int entries = DEEP ? stack_lds_entries(a.max_depth) : tile_stack_lds_entries(a.max_depth);
        float* share = (float*)(lds_stack + (size_t)kWavesPerBlock * entries * kStackFields * 64) + wave * kShareDwords;
        TileQueue feed;
        feed.renders = (CArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // the one argument, where it lies
        feed.tile_end = nullptr;
        feed.queue = a.tile_queue;
        feed.n_tiles = a.tiles_x * a.tiles_y;
        feed.n_waves = (int)gridDim.x * kWavesPerBlock;
        trace_queue<MW, COUNT, TREE, DEEP, LATE, GRID>(feed, stk, gstk, share, lane, cnt);
    } else {
        const int tx = blockIdx.x * kWavesPerBlock + wave;  // a block = 4 tiles side by side
        int ty = blockIdx.y;
        const int x = tx * kTile + (lane & 7);
        int k;  // row in strip order
        bool have;
        if (MW == 0 && !COUNT) {
            // A launch of this kind is mostly short waves, and what bounds those is the scalar
            // unit of the CU (one scalar instruction per cycle for four SIMDs): every scalar
            // instruction in front of the walk is paid 129,600 times per 4K frame.  So the test
            // that ends most waves comes first and reads two words: a tile outside the screen
            // bounds of the scene is background (src/main.cc:49-53).
            const RT_CONST uint32_t* kp = (const RT_CONST uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
            const RT_CONST uint32_t* sky = (const RT_CONST uint32_t*)((const RT_CONST char*)kp + sizeof(KArgs));
            // (the table is indexed by the block row: the host has laid it out in launch order)
            const uint32_t row = min((uint32_t)blockIdx.y, (uint32_t)(kSkyRows - 1));
            const uint32_t rng = sky[row >> 1] >> ((row & 1u) << 4);
            const FrontArgs fa = load_front_args(kp);
            const uint32_t col = (uint32_t)tx >> fa.sky_unit_shift();
            const bool outside = col < (rng & 0xffu) || col > ((rng >> 8) & 0xffu);
            // The rows of tiles are launched in the order the host chose: those that hold the long
            // ray trees first, so that the launch ends on short waves (KArgs::row_order)
            const uint32_t order = fa.row_order();
            const int tiles_y = (fa.rows() + kTile - 1) >> 3;  // (= gridDim.y, without the fetch)
            ty = (order & 0x40000000u) ? (int)(order & 0xffffu) - ty : (int)(order & 0xffffu) + ty;
            ty += (ty < 0) ? tiles_y : 0;
            ty -= (ty >= tiles_y) ? tiles_y : 0;
            k = ty * kTile + (lane >> 3);
            have = (x < fa.width()) && (k < fa.rows());
            if (outside) {
                if (have) {
                    uint8_t* dst = fa.bgr() + 3 * ((size_t)k * fa.width() + x);
                    dst[0] = (uint8_t)to_u8(fa.bgr_c(2));
                    dst[1] = (uint8_t)to_u8(fa.bgr_c(1));
                    dst[2] = (uint8_t)to_u8(fa.bgr_c(0));
                }
                return;
            }
            // pixels that end at the primary ray are finished in single precision (launches with
            // the container-order loops and the short stack; KArgs::front null = switched off)
            if (!TREE && !DEEP && !GRID && fa.front() != nullptr && fa.n_front() >= 0) {
                have = have & !primary_only_pixel(fa, x, k, have);
                if (!__any(have)) return;
            }
#ifdef RT_PASS_ONLY  // measurement builds: what the launch costs without the walk
            return;
#endif
        } else {
            k = ty * kTile + (lane >> 3);
            have = (x < a.width) && (k < a.rows);
        }
        trace_tile<MW, COUNT, TREE, DEEP, LATE, GRID>(a, x, k, have, stk, nullptr, lane, cnt);
    }
    flush_counters<COUNT>(a, cnt, lane);
}

template <int MW, bool COUNT, bool TREE, bool DEEP, bool LATE, bool GRID>
static hipError_t launch_ttdl(const KArgs& a, int deep_blocks, hipStream_t stream, const SkyTable& sky) {
    const int bx = (a.tiles_x + kWavesPerBlock - 1) / kWavesPerBlock;
    if (bx <= 0 || a.tiles_y <= 0) return hipSuccess;
    size_t lds = DEEP ? stack_lds_bytes(a.max_depth) : tile_stack_lds_bytes(a.max_depth);
    if ((DEEP || TREE) && share_rays_in(MW, COUNT, GRID)) lds += share_lds_bytes();
#ifdef RT_PASS_ONLY
#ifndef RT_PASS_KEEP_LDS
    if (!DEEP && !TREE && MW == 0 && !COUNT) lds = 0;
#endif
#endif
    if (DEEP || TREE) {
        if (deep_blocks < 1 || deep_blocks > (DEEP ? kDeepMaxBlocks : kQueueMaxBlocks) || !a.tile_queue || (DEEP && !a.gstack)) return hipErrorInvalidValue;
        hipLaunchKernelGGL((whitted_kernel<MW, COUNT, TREE, DEEP, LATE, GRID>), dim3(deep_blocks), dim3(64 * kWavesPerBlock), lds, stream, a, sky);
    } else {
        hipLaunchKernelGGL((whitted_kernel<MW, COUNT, TREE, DEEP, LATE, GRID>), dim3(bx, a.tiles_y), dim3(64 * kWavesPerBlock), lds, stream, a, sky);
    }
    return hipGetLastError();
}

// Several renders in one launch; the arguments of all of them travel by value in the kernel
// argument segment.  Image-only variant.  Without the deep stack blockIdx.z selects the
// render, the grid is sized for the largest one and blocks outside a smaller one return at
// once; with it the renders' tiles form one queue (tile_end = running totals).
template <bool TREE, bool DEEP, bool LATE, bool GRID>
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(image_waves(GRID), image_waves(GRID) > 4 ? image_waves(GRID) : 4)))
void whitted_batch_kernel(KBatch, int n) {
    extern __shared__ double lds_stack[];
    const RT_CONST KBatch& b = *(const RT_CONST KBatch*)__builtin_amdgcn_kernarg_segment_ptr();  // the first argument
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t cnt[kCounters] = {0};
    if (DEEP || TREE) {
        // every render's stack entries sit at the offsets of the deepest render of the launch
        int depth = 0;
        for (int i = 0; i < n; ++i) depth = max(depth, b.a[i].max_depth);
        double* stk = lds_stack + (size_t)wave * (DEEP ? stack_lds_entries(depth) : tile_stack_lds_entries(depth)) * kStackFields * 64 + lane;
        double* gstk = DEEP ? wave_gstack(b.a[0], (size_t)blockIdx.x * kWavesPerBlock + wave) : nullptr;
        const int entries = DEEP ? stack_lds_entries(depth) : tile_stack_lds_entries(depth);
        float* share = (float*)(lds_stack + (size_t)kWavesPerBlock * entries * kStackFields * 64) + wave * kShareDwords;
        TileQueue feed;
        feed.renders = b.a;
        feed.tile_end = b.tile_end;
        feed.queue = b.a[0].tile_queue;
        feed.n_tiles = b.tile_end[n - 1];
        feed.n_waves = (int)gridDim.x * kWavesPerBlock;
        trace_queue<0, false, TREE, DEEP, LATE, GRID>(feed, stk, gstk, share, lane, cnt);
    } else {
        CArgs& a = b.a[blockIdx.z];
        if ((int)blockIdx.y >= a.tiles_y || (int)(blockIdx.x * kWavesPerBlock) >= a.tiles_x) return;
        double* stk = lds_stack + (size_t)wave * tile_stack_lds_entries(a.max_depth) * kStackFields * 64 + lane;
        const int tx = blockIdx.x * kWavesPerBlock + wave;
        // (the render's rows of tiles in the order its host chose, KArgs::row_order)
        const uint32_t order = (uint32_t)a.row_order;
        int ty = (order & 0x40000000u) ? (int)(order & 0xffffu) - (int)blockIdx.y : (int)(order & 0xffffu) + (int)blockIdx.y;
        ty += (ty < 0) ? a.tiles_y : 0;
        ty -= (ty >= a.tiles_y) ? a.tiles_y : 0;
        const int x = tx * kTile + (lane & 7);
        const int k = ty * kTile + (lane >> 3);
        bool have = (x < a.width) && (k < a.rows);
        if (!GRID && a.front != nullptr && a.n_front >= 0) {
            const FrontArgs fa = load_front_args((const RT_CONST uint32_t*)&a);
            have = have & !primary_only_pixel(fa, x, k, have);
            if (!__any(have)) return;
        }
        trace_tile<0, false, TREE, DEEP, LATE, GRID>(a, x, k, have, stk, nullptr, lane, cnt);
    }
}

template <bool TREE, bool DEEP>
static void launch_batch_td(const KBatch& b, int n, bool late, bool grid, dim3 g, dim3 blk, size_t lds, hipStream_t stream) {
    if (grid) hipLaunchKernelGGL((whitted_batch_kernel<TREE, DEEP, false, true>), g, blk, lds, stream, b, n);
    else if (late && !TREE) hipLaunchKernelGGL((whitted_batch_kernel<TREE, DEEP, !TREE, false>), g, blk, lds, stream, b, n);
    else hipLaunchKernelGGL((whitted_batch_kernel<TREE, DEEP, false, false>), g, blk, lds, stream, b, n);
}

hipError_t launch_whitted_batch(const KBatch& b, int n, bool tree, bool grid, int deep_blocks, hipStream_t stream) {
    int bx = 0, by = 0, depth = 0;
    for (int i = 0; i < n; ++i) {
        const KArgs& a = b.a[i];
        bx = max(bx, (a.tiles_x + kWavesPerBlock - 1) / kWavesPerBlock);
        by = max(by, a.tiles_y);
        depth = max(depth, a.max_depth);
    }
    if (n <= 0 || bx <= 0 || by <= 0) return hipSuccess;
    const bool deep = stack_global_entries(depth) > 0;
    const bool queue = deep || tree;
    size_t lds = deep ? stack_lds_bytes(depth) : tile_stack_lds_bytes(depth);
    if (queue && share_rays_in(0, false, grid)) lds += share_lds_bytes();
    if (queue && (deep_blocks < 1 || deep_blocks > (deep ? kDeepMaxBlocks : kQueueMaxBlocks) || !b.a[0].tile_queue || (deep && !b.a[0].gstack)))
        return hipErrorInvalidValue;
    const dim3 g = queue ? dim3(deep_blocks) : dim3(bx, by, n);
    const dim3 block(64 * kWavesPerBlock);
    bool late = !tree;  // every render of the batch must have many lights
    for (int i = 0; i < n; ++i) {
        late = late && b.a[i].n_lights >= kLateMinLights;
        if (grid && !b.a[i].grid_cells) return hipErrorInvalidValue;
    }
    if (tree && deep) launch_batch_td<true, true>(b, n, late, grid, g, block, lds, stream);
    else if (tree) launch_batch_td<true, false>(b, n, late, grid, g, block, lds, stream);
    else if (deep) launch_batch_td<false, true>(b, n, late, grid, g, block, lds, stream);
    else launch_batch_td<false, false>(b, n, late, grid, g, block, lds, stream);
    return hipGetLastError();
}

// The late-predicate shadow loop exists for the image-only kernel without the tree only; a
// launch that uses the anchor grids has no use for it.
template <int MW, bool COUNT, bool TREE, bool DEEP>
static hipError_t launch_ttd(const KArgs& a, bool grid, int deep_blocks, hipStream_t stream, const SkyTable& sky) {
    if (grid) {
        if (!a.grid_cells || !a.grid_entries) return hipErrorInvalidValue;
        return launch_ttdl<MW, COUNT, TREE, DEEP, false, true>(a, deep_blocks, stream, sky);
    }
    if (MW == 0 && !COUNT && !TREE && a.n_lights >= kLateMinLights)
        return launch_ttdl<MW, COUNT, TREE, DEEP, (MW == 0 && !COUNT && !TREE), false>(a, deep_blocks, stream, sky);
    return launch_ttdl<MW, COUNT, TREE, DEEP, false, false>(a, deep_blocks, stream, sky);
}

template <int MW, bool COUNT, bool TREE>
static hipError_t launch_tt(const KArgs& a, bool grid, int deep_blocks, hipStream_t stream, const SkyTable& sky) {
    if (stack_global_entries(a.max_depth) == 0) return launch_ttd<MW, COUNT, TREE, false>(a, grid, deep_blocks, stream, sky);
    return launch_ttd<MW, COUNT, TREE, true>(a, grid, deep_blocks, stream, sky);
}

template <int MW, bool COUNT>
static hipError_t launch_t(const KArgs& a, bool tree, bool grid, int deep_blocks, hipStream_t stream, const SkyTable& sky) {
    return tree ? launch_tt<MW, COUNT, true>(a, grid, deep_blocks, stream, sky) : launch_tt<MW, COUNT, false>(a, grid, deep_blocks, stream, sky);
}

hipError_t launch_whitted(const KArgs& a, bool emit, bool count, bool tree, bool grid, int deep_blocks, hipStream_t stream,
                          const SkyTable* sky_in) {
    const int mw = emit ? a.mask_words : 0;
    SkyTable all;
    if (!sky_in)
        for (int i = 0; i < kSkyRows; ++i) all.cols[i] = 0xff00u;  // every column of every row
    const SkyTable& sky = sky_in ? *sky_in : all;
    if (count) {
        switch (mw) {
            case 0: return launch_t<0, true>(a, tree, grid, deep_blocks, stream, sky);
            case 1: return launch_t<1, true>(a, tree, grid, deep_blocks, stream, sky);
            case 2: return launch_t<2, true>(a, tree, grid, deep_blocks, stream, sky);
            case 4: return launch_t<4, true>(a, tree, grid, deep_blocks, stream, sky);
        }
    } else {
        switch (mw) {
            case 0: return launch_t<0, false>(a, tree, grid, deep_blocks, stream, sky);
            case 1: return launch_t<1, false>(a, tree, grid, deep_blocks, stream, sky);
            case 2: return launch_t<2, false>(a, tree, grid, deep_blocks, stream, sky);
            case 4: return launch_t<4, false>(a, tree, grid, deep_blocks, stream, sky);
        }
    }
    return hipErrorInvalidValue;
}

// rt_debug_intersect: one lane per ray, primitive broadcast; the test is made the way the
// render kernel makes it (certain-miss filter, then the double test).
__global__ void debug_intersect_kernel(const DGeom* geom_, const FGeom* fgeom_, float p1max, int prim,
                                       const double* rays, int n, double* t_out, uint8_t* hit_out,
                                       double* unit_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DGeom g = load_geom((cdata_t)(const double*)geom_, prim);
    V3 o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    V3 d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    double t = 0.0;
    const FRec gf = load_rec((cfdata_t)(const float*)fgeom_ + 8 * prim);
    const FilterRay fr = filter_ray(o, d, p1max);
    // the filter presumes a unit direction, as every ray of the render kernel has
    const bool unit_dir = __builtin_fabs(len2(d) - 1.0) < 1e-12;
    bool hit = !(unit_dir && certain_miss(gf, fr)) && test_collision(g, o, d, t);
    t_out[i] = hit ? t : 0.0;
    hit_out[i] = hit ? 1 : 0;
    if (unit_out) {
        V3 u = unit(d);
        unit_out[3 * i] = u.x;
        unit_out[3 * i + 1] = u.y;
        unit_out[3 * i + 2] = u.z;
    }
}

// rt_debug_occluder: one lane per point, the feeler search exactly as the render kernel runs it.
template <bool TREE, bool GRID>
__global__ void debug_occluder_kernel(KArgs a, int light, int self, const double* points, int n, int32_t* first_out,
                                      double* factor_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cdata_t geom = (cdata_t)(const double*)a.geom;
    const cdata_t lights = (cdata_t)(const double*)a.lights;
    const V3 ip = mk(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    const LightRec L = load_light(lights, light);
    float occl = 1.0f;
    uint32_t c0 = 0, c1 = 0;
    int first;
    if (GRID) {
        F3 fd;
        first = feeler_by_grid<false, TREE>(a, geom, (cfdata_t)(const float*)a.fgeom, (cfdata_t)(const float*)a.ltab, light, ip,
                                            mk(L.pos[0], L.pos[1], L.pos[2]), self, false, occl, fd, c0, c1);
    } else {
        const V3 fdir = unit(mk(L.pos[0], L.pos[1], L.pos[2]) - ip);
        FilterRay ffr = filter_origin(ip, a.scene_p1max);
        first = shadow_feeler<false, TREE, false, false>(a, geom, (cfdata_t)(const float*)a.fgeom, (cfdata_t)(const float*)a.ltab,
                                                         light, ffr, ip, fdir, self, false, occl, c0, c1);
    }
    first_out[i] = first;
    factor_out[i] = occl;
}

// rt_debug_nearest: one lane per ray, the nearest-hit search exactly as the render kernel runs it.
template <bool TREE, bool GRID>
__global__ void debug_nearest_kernel(KArgs a, const double* rays, int n, int32_t* index_out, double* t_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    const V3 d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    double nearest = __builtin_inf();
    int cur = -1;
    uint32_t c0 = 0, c1 = 0;
    // rays that start at the camera take the camera's grid when every ray of the wave does
    const bool primary = (o.x == 0.0) & (o.y == 0.0) & (o.z == 0.0);
    find_nearest<false, TREE, GRID>(a, (cdata_t)(const double*)a.geom, (cfdata_t)(const float*)a.fgeom, o, d, primary, nearest,
                                    cur, c0, c1);
    index_out[i] = cur;
    t_out[i] = (cur >= 0) ? nearest : 0.0;
}

// rt_debug_nearest / rt_debug_occluder with RT_DEBUG_RECORDS: the single-precision searches of
// primary_only_pixel() on their own, one lane per ray / point, over the same records with the
// same per-record tests.  -2 = not certain at that precision (the render kernel then walks the
// pixel in double).
__global__ void debug_records_nearest_kernel(KArgs a, const double* rays, int n, int32_t* index_out, double* t_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float U = 6.0e-8f;
    const cfdata_t planes = (cfdata_t)(const float*)a.front;
    const int n_planes = a.n_planes, n_spheres = a.n_front;
    const int planes2 = (n_planes + 1) & ~1, spheres2 = n_spheres > 0 ? ((n_spheres + 1) & ~1) : 2;
    const cfdata_t camera = planes + planes2 * kFrontPlaneDwords + (size_t)a.n_lights * (kFrontAnchorDwords + spheres2 * 8) + kFrontAnchorDwords;
    const V3 d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);  // a unit vector from the origin
    FeelRay pr;
    pr.o = f3(0.0f, 0.0f, 0.0f);
    pr.d = f3((float)d.x, (float)d.y, (float)d.z);
    pr.dist = 0.0f;
    pr.en_ray = 0.0f;
    pr.eray = 1.0e-13f * a.scene_p1max * a.scene_p1max;
    pr.efront = 16.0f * U * a.scene_p1max;
    pr.valid = true;
    FrontState s;
    s.ok = true; s.cur = -1; s.ends = false; s.vq = 1.0; s.hd = 0.0; s.pb = 0.0f; s.amb = 0.0f;
    s.nf = f3(0.0f, 0.0f, 0.0f); s.kd = s.nf;
    for (int j = 0; j < n_planes; ++j) {
        uint32_t pj[20];
        front_load(pj, planes + j * kFrontPlaneDwords);
        front_primary_plane(pj, pr, d, s);
    }
    for (int j = 0; j < spheres2; ++j) {
        uint32_t cj[4];
        front_load(cj, camera + j * 8);
        front_primary_sphere(cj, pr, s.ok);
    }
    index_out[i] = s.ok ? s.cur : -2;
    t_out[i] = (s.ok && s.cur >= 0) ? s.hd / s.vq : 0.0;  // (p0.n) / (d.n): the hit parameter up to rounding
}
__global__ void debug_records_occluder_kernel(KArgs a, int light, int self, const double* points, int n, int32_t* first_out,
                                              double* factor_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cfdata_t planes = (cfdata_t)(const float*)a.front;
    const int n_planes = a.n_planes, n_spheres = a.n_front;
    const int planes2 = (n_planes + 1) & ~1, spheres2 = n_spheres > 0 ? ((n_spheres + 1) & ~1) : 2;
    const cfdata_t anchor = planes + planes2 * kFrontPlaneDwords + (size_t)light * (kFrontAnchorDwords + spheres2 * 8);
    const RT_CONST double* lp = (const RT_CONST double*)anchor;
    const V3 ip = mk(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    const FeelRay fr = feel_ray(ip, mk(lp[0], lp[1], lp[2]) - ip, a.scene_p1max);
    bool good = fr.valid;
    int best = 0x7fffffff;
    float occl = 1.0f;
    for (int j = 0; j < n_planes; ++j) {
        uint32_t pj[8];
        front_load(pj, planes + j * kFrontPlaneDwords);
        front_feeler_plane(pj, fr, self, good, best, occl);
    }
    for (int j = 0; j < spheres2; ++j) {
        uint32_t ej[8];
        front_load(ej, anchor + kFrontAnchorDwords + j * 8);
        front_feeler_sphere(ej, fr, self, good, best, occl);
    }
    first_out[i] = good ? ((best != 0x7fffffff) ? best : -1) : -2;
    factor_out[i] = (good && best != 0x7fffffff) ? occl : 1.0;
}
hipError_t launch_debug_records(const KArgs& a, bool occluder, int light, int self, const double* in, int n, int32_t* index_out,
                                double* value_out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (!a.front) return hipErrorInvalidValue;
    const dim3 g((n + 255) / 256), block(256);
    if (occluder) hipLaunchKernelGGL(debug_records_occluder_kernel, g, block, 0, stream, a, light, self, in, n, index_out, value_out);
    else hipLaunchKernelGGL(debug_records_nearest_kernel, g, block, 0, stream, a, in, n, index_out, value_out);
    return hipGetLastError();
}

hipError_t launch_debug_nearest(const KArgs& a, bool tree, bool grid, const double* rays, int n, int32_t* index_out,
                                double* t_out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (grid && !a.grid_cells) return hipErrorInvalidValue;
    const dim3 g((n + 255) / 256), block(256);
    if (tree && grid) hipLaunchKernelGGL((debug_nearest_kernel<true, true>), g, block, 0, stream, a, rays, n, index_out, t_out);
    else if (tree) hipLaunchKernelGGL((debug_nearest_kernel<true, false>), g, block, 0, stream, a, rays, n, index_out, t_out);
    else if (grid) hipLaunchKernelGGL((debug_nearest_kernel<false, true>), g, block, 0, stream, a, rays, n, index_out, t_out);
    else hipLaunchKernelGGL((debug_nearest_kernel<false, false>), g, block, 0, stream, a, rays, n, index_out, t_out);
    return hipGetLastError();
}

hipError_t launch_debug_occluder(const KArgs& a, bool tree, bool grid, int light, int self, const double* points, int n,
                                 int32_t* first_out, double* factor_out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (grid && !a.grid_cells) return hipErrorInvalidValue;
    const dim3 g((n + 255) / 256), block(256);
    if (tree && grid)
        hipLaunchKernelGGL((debug_occluder_kernel<true, true>), g, block, 0, stream, a, light, self, points, n, first_out, factor_out);
    else if (tree)
        hipLaunchKernelGGL((debug_occluder_kernel<true, false>), g, block, 0, stream, a, light, self, points, n, first_out, factor_out);
    else if (grid)
        hipLaunchKernelGGL((debug_occluder_kernel<false, true>), g, block, 0, stream, a, light, self, points, n, first_out, factor_out);
    else
        hipLaunchKernelGGL((debug_occluder_kernel<false, false>), g, block, 0, stream, a, light, self, points, n, first_out, factor_out);
    return hipGetLastError();
}

hipError_t launch_debug_intersect(const DGeom* geom, const FGeom* fgeom, float scene_p1max, int prim,
                                  const double* rays, int n, double* t_out, uint8_t* hit_out, double* unit_out,
                                  hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(debug_intersect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, geom, fgeom, scene_p1max, prim, rays, n,
                       t_out, hit_out, unit_out);
    return hipGetLastError();
}

}  // namespace rt
