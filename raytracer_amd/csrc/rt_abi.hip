// C ABI of the tracer (include/rt_hip.h): context, scene upload, render, test hooks.
// Host code only; the kernels live in whitted_kernels.hip.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <array>

#include "anchor_grid.h"
#include "sky_table.h"
#include "sphere_tree.h"
#include "whitted_device.h"

#pragma clang fp contract(off)

namespace {

thread_local std::string g_create_error = "";

struct DevBuf {
    void* ptr = nullptr;
    size_t cap = 0;
};

}  // namespace

struct rt_ctx {
    int device = 0;
    int n_cus = 256;
    std::string err;
    rt::DGeom* d_geom = nullptr;
    rt::DMat* d_mat = nullptr;
    rt::DLight* d_lights = nullptr;
    int32_t n_prims = 0, n_lights = 0;
    bool have_scene = false;
    // float image of the geometry for the certain-miss filter
    rt::FGeom* f_geom = nullptr;
    float scene_p1max = 0.0f;
    // sphere tree + the planes' records (used when the scene has many spheres)
    rt::FGeom* f_tree = nullptr;
    rt::FGeom* f_planes = nullptr;
    int32_t* d_planes_before = nullptr;
    int32_t n_nodes = 0, n_planes = 0;
    // anchor grids (anchor_grid.h): one per light and one for the camera
    rt::GridCell* d_grid_cells = nullptr;
    rt::GridEntry* d_grid_entries = nullptr;
    int32_t grid_res = 0;
    float grid_eray_max = 0.0f;
    float* d_ltab = nullptr;   // light-anchored line-test constants, n_lights x ltab_stride x 4
    float* d_front = nullptr;  // KArgs::front; null = scene too large for that pass
    int32_t n_front = 0;
    int32_t ltab_stride = 0;
    // grow-only device staging for rt_render's host outputs
    DevBuf bgr, hit_id, hit_mask, spawn_mask, counters, dbg_in, dbg_t, dbg_hit, dbg_unit;
    // workspace of deep-stack launches (max_depth > 5: tile queue + the global part of the
    // per-lane stacks): one per stream the context has made such a launch on, since
    // launches on one stream run one after the other but launches on different streams may
    // overlap.  Its size depends on the device only (workspace_bytes), never on the frame;
    // rt_trim() frees them all.
    std::map<void*, void*> workspaces;
    std::map<void*, void*> queues;   // streams that only ever needed the tile queue (256 bytes): slots
                                     // of `queue_pool` (allocated by rt_create), own allocations beyond
    void* queue_pool = nullptr;
    int queue_pool_used = 0;
    size_t workspace_bytes = 0;
    // host copy of the scene and the screen bounds worked out from it (sky_table.h), kept per
    // partition the context has rendered lately
    std::vector<rt_prim> host_prims;
    struct SkyEntry {
        rt_params p;
        rt::SkyTable t;        // in launch order (indexed by block row)
        int32_t row_order = 0; // KArgs::row_order
    };
    std::vector<SkyEntry> sky_cache;
    size_t sky_next = 0;
    hipStream_t stream = nullptr;
    // rt_render / rt_render_node: second stream and events that let the copy of one part of
    // a frame run while the next part is rendered
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> part_done;
    hipEvent_t copies_done = nullptr;
    // rt_render into ordinary (pageable) host memory: page-locked buffer the parts are copied
    // into by DMA, and the events that tell when a part has arrived there
    void* host_stage = nullptr;
    size_t host_stage_bytes = 0;
    std::vector<hipEvent_t> part_arrived;
};

namespace {

int fail(rt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(rt_ctx* ctx, hipError_t e, const char* what) {
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    int code = (e == hipErrorOutOfMemory) ? RT_ERR_NOMEM : RT_ERR_HIP;
    return fail(ctx, code, msg);
}

// Makes `device` current for the lifetime of the object and puts the caller's device back
// afterwards: the entry points must not change the calling thread's current device.
struct DeviceGuard {
    int prev = -1;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device) {
        status = hipGetDevice(&prev);
        if (status != hipSuccess) { prev = -1; return; }
        if (prev != device) status = hipSetDevice(device);
        else prev = -1;  // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define RT_HIP_TRY(ctx, call)                                 \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return hip_fail(ctx, e__, #call); \
    } while (0)

int reserve(rt_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return RT_OK;
    if (b.ptr) RT_HIP_TRY(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.cap = 0;
    RT_HIP_TRY(ctx, hipMalloc(&b.ptr, bytes));
    b.cap = bytes;
    return RT_OK;
}

int check_params(const rt_params* p) {
    if (!p) return RT_ERR_ARG;
    if (p->width < 1 || p->height < 1 || p->width > 65536 || p->height > 65536) return RT_ERR_ARG;
    if (p->max_depth < 2 || p->max_depth > RT_MAX_DEPTH) return RT_ERR_ARG;
    if (p->band_rows < 1 || p->band_first < 0 || p->band_step < 1 || p->band_count < 0) return RT_ERR_ARG;
    if (!(p->fov_deg > 0.0 && p->fov_deg < 180.0)) return RT_ERR_ARG;
    return RT_OK;
}

// Tile queues handed out without an allocation: the first kQueuePoolSlots streams a context
// makes a queue-only launch on (sphere-tree scenes at max_depth <= 5) take a slot of a pool
// that rt_create allocated, so such renders can be enqueued where hipMalloc is not allowed.
constexpr int kQueuePoolSlots = 64;

// The workspace of `stream` for launches with a fixed grid: tile queue + (deep: the stack
// regions).  Allocated at the first such launch on that stream; a launch that only needs the
// queue takes the header of the full workspace when the stream has one, else 256 bytes.
int stream_workspace(rt_ctx* ctx, void* stream, bool deep, void** out) {
    auto full = ctx->workspaces.find(stream);
    if (full != ctx->workspaces.end() && full->second) {
        *out = full->second;
        return RT_OK;
    }
    if (deep) {
        void* w = nullptr;
        RT_HIP_TRY(ctx, hipMalloc(&w, ctx->workspace_bytes));
        RT_HIP_TRY(ctx, hipMemsetAsync(w, 0, rt::kWorkspaceHeader, static_cast<hipStream_t>(stream)));
        ctx->workspaces[stream] = w;
        *out = w;
        return RT_OK;
    }
    void*& q = ctx->queues[stream];
    if (!q) {
        if (ctx->queue_pool && ctx->queue_pool_used < kQueuePoolSlots)
            q = static_cast<char*>(ctx->queue_pool) + rt::kWorkspaceHeader * ctx->queue_pool_used++;
        else {
            RT_HIP_TRY(ctx, hipMalloc(&q, rt::kWorkspaceHeader));
            RT_HIP_TRY(ctx, hipMemsetAsync(q, 0, rt::kWorkspaceHeader, static_cast<hipStream_t>(stream)));
        }
    }
    *out = q;
    return RT_OK;
}

// Point the arguments of a launch with a fixed grid at workspace `ws` and work out the grid
// size.  The tile queue at the head of the workspace is zero whenever no launch is using
// it: zeroed when the workspace is made, and put back to zero by the last wave of every launch.
int attach_workspace(rt_ctx* ctx, rt::KArgs* args, int n, long long tiles, int max_depth, bool grid, void* ws,
                     int* deep_blocks) {
    const bool deep = rt::stack_global_entries(max_depth) > 0;
    *deep_blocks = rt::deep_grid_blocks(ctx->n_cus, tiles, deep, grid);
    for (int i = 0; i < n; ++i) {
        args[i].tile_queue = static_cast<unsigned int*>(ws);
        args[i].gstack = deep ? reinterpret_cast<double*>(static_cast<char*>(ws) + rt::kWorkspaceHeader) : nullptr;
    }
    return RT_OK;
}

// Fill the kernel arguments for the partition in p.  The camera arithmetic is the
// ViewFrustum constructor's (src/view_frustum.h:18,28-34), evaluated on the host in
// double exactly as written there.

// The tree pays once a ray has more spheres to test than a walk visits nodes.
constexpr int kTreeMinSpheres = 40;
bool use_tree(const rt_ctx* ctx, const rt_params* p) {
    if (p->flags & RT_FLAG_LINEAR) return false;
    if (p->flags & RT_FLAG_TREE) return ctx->n_nodes > 0;
    return ctx->n_prims - ctx->n_planes >= kTreeMinSpheres;
}

// The anchor grids pay from a handful of spheres on (each feeler then tests the spheres of one
// cell instead of all of them).
constexpr int kGridMinSpheres = 8;
// Scenes with at most this many spheres and planes get the table of the pass that finishes
// pixels ending at the primary ray (it tests every object of the scene; launches with more
// spheres than kGridMinSpheres do not run it anyway unless the caller switches the grids off).
constexpr size_t kFrontMaxSpheres = 16, kFrontMaxPlanes = 8;
bool use_grid(const rt_ctx* ctx, const rt_params* p) {
    if (!ctx->d_grid_cells || (p->flags & RT_FLAG_NOGRID)) return false;
    if (p->flags & RT_FLAG_GRID) return true;
    return ctx->n_prims - ctx->n_planes >= kGridMinSpheres;
}

int make_args(rt_ctx* ctx, const rt_params* p, rt::KArgs& a) {
    if (check_params(p) != RT_OK) return fail(ctx, RT_ERR_ARG, "rt_params: value out of range");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    std::memset(&a, 0, sizeof a);
    a.geom = ctx->d_geom;
    a.mat = ctx->d_mat;
    a.lights = ctx->d_lights;
    a.fgeom = ctx->f_geom;
    a.tree = ctx->f_tree;
    a.planes = ctx->f_planes;
    a.planes_before = ctx->d_planes_before;
    a.n_nodes = ctx->n_nodes;
    a.n_planes = ctx->n_planes;
    a.grid_cells = ctx->d_grid_cells;
    a.grid_entries = ctx->d_grid_entries;
    a.grid_res = ctx->grid_res;
    a.grid_half = 0.5f * static_cast<float>(ctx->grid_res);
    a.grid_eray_max = ctx->grid_eray_max;
    a.ltab = ctx->d_ltab;
    a.ltab_stride = ctx->ltab_stride;
    a.front = (p->flags & RT_FLAG_NOFRONT) ? nullptr : ctx->d_front;
    a.n_front = ctx->n_front;
    {   // where the blocks of `front` lie (the layout rt_upload_scene wrote), in dwords
        const int planes2 = (ctx->n_planes + 1) & ~1;
        a.front_spheres = ctx->n_front > 0 ? ((ctx->n_front + 1) & ~1) : 2;
        a.front_anchors = planes2 * rt::kFrontPlaneDwords;
        a.front_stride = rt::kFrontAnchorDwords + a.front_spheres * 8;
        a.front_camera = a.front_anchors + ctx->n_lights * a.front_stride + rt::kFrontAnchorDwords;
    }
    a.scene_p1max = ctx->scene_p1max;
    a.n_prims = ctx->n_prims;
    a.n_lights = ctx->n_lights;
    a.width = p->width;
    a.height = p->height;
    a.max_depth = p->max_depth;
    a.band_rows = p->band_rows;
    a.band_first = p->band_first;
    a.band_step = p->band_step;
    a.rows = rt_strip_rows(p);
    a.tiles_x = (a.width + rt::kTile - 1) / rt::kTile;
    a.tiles_y = (a.rows + rt::kTile - 1) / rt::kTile;
    a.mask_words = rt_mask_words(p->max_depth);

    const double TO_RAD = 3.141592653589793 / 180.;
    const double aspect = static_cast<double>(p->height) / static_cast<double>(p->width);
    const double phy_width = 2 * 1 * std::tan(0.5 * p->fov_deg * TO_RAD);
    const double phy_height = aspect * phy_width;
    a.pw = phy_width / p->width;
    a.ph = phy_height / p->height;
    for (int i = 0; i < 3; ++i) a.bgr_c[i] = static_cast<float>(p->bgr[i]);
    for (int dpt = 1; dpt <= RT_MAX_DEPTH; ++dpt) a.inv_depth[dpt] = static_cast<float>(1.0 / dpt);  // src/main.cc:112
    // row k of the strip lies in band k / band_rows: exact as a multiply-high for k < 2^16
    // (frames are at most 65536 rows; a band of 65536 rows holds them all)
    a.band_magic = (p->band_rows >= 65536) ? 0u : static_cast<uint32_t>(((1ull << 32) + p->band_rows - 1) / p->band_rows);
    if (p->band_rows == 1) a.band_magic = 0xffffffffu;  // unused: band = k
    // the primary ray's p1 = ((xi + .5) pw, (yi + .5) ph, 1): inside the range in which the
    // short normalisation is exact when the smallest and largest components are
    const double cmin = 0.5 * (a.pw < a.ph ? a.pw : a.ph);
    const double cmax = (0.5 * p->width + 1.0) * a.pw + (0.5 * p->height + 1.0) * a.ph;
    a.primary_in_range = (cmin >= 0x1p-370 && cmax <= 0x1p370) ? 1 : 0;
    a.prune = (p->flags & RT_FLAG_PRUNE) ? 1 : 0;
    a.sky_unit_shift = rt::sky_unit_shift(p->width);
    return RT_OK;
}

}  // namespace

extern "C" {

int32_t rt_mask_words(int32_t max_depth) {
    if (max_depth < 1 || max_depth > RT_MAX_DEPTH) return RT_ERR_ARG;
    const int bits = 1 << (max_depth - 1);
    return bits < 32 ? 1 : bits / 32;
}

int32_t rt_strip_rows(const rt_params* p) {
    if (check_params(p) != RT_OK) return RT_ERR_ARG;
    int64_t rows = 0;
    for (int b = 0; b < p->band_count; ++b) {
        const int64_t y0 = (static_cast<int64_t>(p->band_first) + static_cast<int64_t>(b) * p->band_step) * p->band_rows;
        if (y0 >= p->height) break;
        int64_t y1 = y0 + p->band_rows;
        if (y1 > p->height) y1 = p->height;
        rows += y1 - y0;
    }
    return static_cast<int32_t>(rows);
}

const char* rt_version(void) { return "0.1"; }

int rt_create(int32_t device_id, rt_ctx** out) {
    if (!out) return fail(nullptr, RT_ERR_ARG, "rt_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, RT_ERR_HIP, std::string("rt_create: no HIP device: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= count) return fail(nullptr, RT_ERR_ARG, "rt_create: device id out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RT_ERR_HIP, std::string("rt_create: device is ") + prop.gcnArchName +
                                             ", this library holds gfx950 code only");
    rt_ctx* ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return fail(nullptr, RT_ERR_NOMEM, "rt_create: out of host memory");
    ctx->device = device_id;
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->workspace_bytes = rt::stack_workspace_bytes(RT_MAX_DEPTH, rt::deep_grid_blocks(ctx->n_cus, 1LL << 40));
    DeviceGuard guard(device_id);
    e = guard.status;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->copies_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&ctx->queue_pool, rt::kWorkspaceHeader * kQueuePoolSlots);
    if (e == hipSuccess) e = hipMemset(ctx->queue_pool, 0, rt::kWorkspaceHeader * kQueuePoolSlots);
    if (e != hipSuccess) {
        rt_destroy(ctx);
        return hip_fail(nullptr, e, "rt_create: device set-up");
    }
    *out = ctx;
    return RT_OK;
}

static void free_workspaces(rt_ctx* ctx) {
    for (auto& kv : ctx->workspaces)
        if (kv.second) (void)hipFree(kv.second);
    ctx->workspaces.clear();
    const char* pool = static_cast<const char*>(ctx->queue_pool);
    for (auto& kv : ctx->queues) {
        const char* q = static_cast<const char*>(kv.second);
        const bool pooled = pool && q >= pool && q < pool + rt::kWorkspaceHeader * kQueuePoolSlots;
        if (q && !pooled) (void)hipFree(kv.second);
    }
    ctx->queues.clear();
    ctx->queue_pool_used = 0;
}

void rt_destroy(rt_ctx* ctx) {
    if (!ctx) return;
    {
        DeviceGuard guard(ctx->device);
        DevBuf* bufs[] = {&ctx->bgr, &ctx->hit_id, &ctx->hit_mask, &ctx->spawn_mask, &ctx->counters,
                          &ctx->dbg_in, &ctx->dbg_t, &ctx->dbg_hit, &ctx->dbg_unit};
        for (DevBuf* b : bufs)
            if (b->ptr) (void)hipFree(b->ptr);
        if (ctx->d_geom) (void)hipFree(ctx->d_geom);
        if (ctx->d_mat) (void)hipFree(ctx->d_mat);
        if (ctx->d_lights) (void)hipFree(ctx->d_lights);
        if (ctx->f_geom) (void)hipFree(ctx->f_geom);
        if (ctx->f_tree) (void)hipFree(ctx->f_tree);
        if (ctx->f_planes) (void)hipFree(ctx->f_planes);
        if (ctx->d_planes_before) (void)hipFree(ctx->d_planes_before);
        if (ctx->d_ltab) (void)hipFree(ctx->d_ltab);
        if (ctx->d_front) (void)hipFree(ctx->d_front);
        if (ctx->d_grid_cells) (void)hipFree(ctx->d_grid_cells);
        if (ctx->d_grid_entries) (void)hipFree(ctx->d_grid_entries);
        free_workspaces(ctx);
        if (ctx->queue_pool) (void)hipFree(ctx->queue_pool);
        for (hipEvent_t ev : ctx->part_done) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : ctx->part_arrived) (void)hipEventDestroy(ev);
        if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
        if (ctx->copies_done) (void)hipEventDestroy(ctx->copies_done);
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

int rt_trim(rt_ctx* ctx) {
    if (!ctx) return RT_ERR_ARG;
    DeviceGuard guard(ctx->device);
    RT_HIP_TRY(ctx, guard.status);
    RT_HIP_TRY(ctx, hipDeviceSynchronize());  // a launch may still be using a workspace
    free_workspaces(ctx);
    return RT_OK;
}

int rt_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(nullptr, RT_ERR_ARG, "rt_host_alloc: bad argument");
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipHostMalloc");
    return RT_OK;
}

void rt_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_upload_scene(rt_ctx* ctx, const rt_prim* prims, int32_t n_prims, const rt_light* lights, int32_t n_lights) {
    if (!ctx) return RT_ERR_ARG;
    if (!prims || n_prims < 1 || n_prims > RT_MAX_PRIMS) return fail(ctx, RT_ERR_ARG, "rt_upload_scene: n_prims");
    if (n_lights < 0 || n_lights > RT_MAX_LIGHTS || (n_lights > 0 && !lights))
        return fail(ctx, RT_ERR_ARG, "rt_upload_scene: n_lights");
    std::vector<rt::DGeom> geom(n_prims);
    std::vector<rt::DMat> mat(n_prims);
    std::vector<rt::DLight> lt(n_lights > 0 ? n_lights : 1);
    for (int i = 0; i < n_prims; ++i) {
        const rt_prim& s = prims[i];
        if (s.type != RT_PRIM_PLANE && s.type != RT_PRIM_SPHERE)
            return fail(ctx, RT_ERR_ARG, "rt_upload_scene: unknown primitive type");
        if (s.reserved != 0) return fail(ctx, RT_ERR_ARG, "rt_upload_scene: rt_prim.reserved must be 0");
        rt::DGeom& g = geom[i];
        rt::DMat& m = mat[i];
        std::memset(&g, 0, sizeof g);
        std::memset(&m, 0, sizeof m);
        for (int c = 0; c < 3; ++c) {
            g.p[c] = s.p[c];
            g.q[c] = s.q[c];
            m.kd[c] = static_cast<float>(s.kd[c]);
        }
        // std::pow(radius_, 2) (src/objects.h:125) is radius_*radius_ correctly rounded.
        g.r2 = (s.type == RT_PRIM_PLANE) ? -1.0 : s.q[0] * s.q[0];
        // The reference takes any numbers; here positions, normals and radii must be finite
        // (the tree builder and the float filters are not defined on NaN/inf input).  A radius
        // may be zero or negative: the reference only ever uses its square.
        for (int c = 0; c < 3; ++c)
            if (!std::isfinite(s.p[c]) || !std::isfinite(s.q[c]))
                return fail(ctx, RT_ERR_ARG, "rt_upload_scene: non-finite position, normal or radius");
        // A plane's normal is the unit vector its constructor stores (src/objects.h:88): the
        // error bounds of the single-precision filters are derived for |normal| = 1.  The
        // zero vector (what gmtl's normalisation leaves of a zero normal) is accepted too.
        if (s.type == RT_PRIM_PLANE) {
            const double n2 = s.q[0] * s.q[0] + s.q[1] * s.q[1] + s.q[2] * s.q[2];
            if (n2 != 0.0 && std::fabs(n2 - 1.0) > 1e-12)
                return fail(ctx, RT_ERR_ARG, "rt_upload_scene: plane normal is not of unit length (pass normal_ as "
                                             "stored by the Plane constructor)");
        }
        g.transparency = static_cast<float>(s.transparency);
        m.reflectivity = static_cast<float>(s.reflectivity);
        m.transparency = static_cast<float>(s.transparency);
        m.refraction_index = s.refraction_index;
        m.ambient_intensity = static_cast<float>(s.ambient_intensity);
        m.inv_refraction_index = 1.0 / s.refraction_index;  // src/main.cc:64, the same division done once
        if (s.type == RT_PRIM_PLANE) {
            // gmtl::makeNormal(normal_): length as a running sum from 0, three divisions, skipped
            // for a zero length - the arithmetic of Plane::calcNormal, done here once
            double l2 = 0.0;
            l2 += s.q[0] * s.q[0];
            l2 += s.q[1] * s.q[1];
            l2 += s.q[2] * s.q[2];
            const double len = std::sqrt(l2);
            for (int c = 0; c < 3; ++c) m.plane_normal[c] = (len != 0.0) ? s.q[c] / len : s.q[c];
        }
    }
    for (int i = 0; i < n_lights; ++i) {
        for (int c = 0; c < 3; ++c)
            if (!std::isfinite(lights[i].pos[c]))
                return fail(ctx, RT_ERR_ARG, "rt_upload_scene: non-finite light position");
        std::memset(&lt[i], 0, sizeof lt[i]);
        for (int c = 0; c < 3; ++c) {
            lt[i].pos[c] = lights[i].pos[c];
            lt[i].ci[c] = static_cast<float>(lights[i].color[c] * lights[i].intensity);  // src/objects.h:54-56
        }
    }
    // float image of the geometry with the rounding error of each record (U = 2^-24
    // rounded up); see the filter's derivation in whitted_exact.inc.h
    const float kU = 6.0e-8f;
    std::vector<rt::FGeom> fgeom((static_cast<size_t>(n_prims) + 3) / 4 * 4);
    float p1max = 0.0f;
    for (int i = 0; i < n_prims; ++i) {
        const rt_prim& s = prims[i];
        rt::FGeom& g = fgeom[i];
        std::memset(&g, 0, sizeof g);
        double p1 = 0.0;
        for (int c = 0; c < 3; ++c) p1 += std::fabs(s.p[c]);
        if (s.type == RT_PRIM_PLANE) {
            // h = origin . normal; herr covers the rounding of h to float and the last
            // subtraction of the filter (|h| <= |origin|_1)
            const double hd = s.p[0] * s.q[0] + s.p[1] * s.q[1] + s.p[2] * s.q[2];
            for (int c = 0; c < 3; ++c) g.v[c] = static_cast<float>(s.q[c]);
            g.v[3] = -1.0f;
            g.v[4] = static_cast<float>(hd);
            g.v[5] = kU * (static_cast<float>(std::fabs(hd)) + static_cast<float>(p1)) * 1.01f;
        } else {
            for (int c = 0; c < 3; ++c) g.v[c] = static_cast<float>(s.p[c]);
            g.v[3] = static_cast<float>(s.q[0] * s.q[0]);
            g.v[4] = static_cast<float>(p1) * kU;
            if (static_cast<float>(p1) * 1.0001f > p1max) p1max = static_cast<float>(p1) * 1.0001f;
        }
    }
    for (size_t i = n_prims; i < fgeom.size(); ++i) fgeom[i] = fgeom[n_prims - 1];  // pad to whole groups of 4
    // sphere tree, the planes' records with their original index, plane prefix counts
    std::vector<rt::TreeSphere> tsph;
    std::vector<rt::FGeom> fplanes;
    std::vector<int32_t> planes_before(static_cast<size_t>(n_prims) + 1, 0);
    for (int i = 0; i < n_prims; ++i) {
        const rt_prim& s = prims[i];
        planes_before[i + 1] = planes_before[i] + (s.type == RT_PRIM_PLANE ? 1 : 0);
        if (s.type == RT_PRIM_PLANE) {
            rt::FGeom g = fgeom[i];
            std::memcpy(&g.v[6], &i, sizeof(int32_t));
            g.v[7] = static_cast<float>(s.transparency);
            fplanes.push_back(g);
        } else {
            tsph.push_back(rt::TreeSphere{{s.p[0], s.p[1], s.p[2]}, std::fabs(s.q[0]), i});
        }
    }
    // per (light, object) constants of the light-anchored line test (whitted_exact.inc.h)
    const size_t lstride = fgeom.size();
    std::vector<float> ltab(static_cast<size_t>(n_lights > 0 ? n_lights : 1) * lstride * 4, 0.0f);
    for (int li = 0; li < n_lights; ++li) {
        double l1 = 0.0;
        for (int c = 0; c < 3; ++c) l1 += std::fabs(lights[li].pos[c]);
        if (static_cast<float>(l1) * 1.0001f > p1max) p1max = static_cast<float>(l1) * 1.0001f;
        for (size_t i = 0; i < lstride; ++i) {
            float* rec = &ltab[(static_cast<size_t>(li) * lstride + i) * 4];
            rec[3] = -INFINITY;  // planes and padding: never ruled out here
            if (i >= static_cast<size_t>(n_prims) || prims[i].type != RT_PRIM_SPHERE) continue;
            const double L[3] = {prims[i].p[0] - lights[li].pos[0], prims[i].p[1] - lights[li].pos[1],
                                 prims[i].p[2] - lights[li].pos[2]};
            rt::line_test_record(L, prims[i].q[0] * prims[i].q[0], rec);
        }
    }
    double tree_extent1 = 0.0;
    int tree_nodes = 0;
    std::vector<rt::FGeom> tree = rt::build_sphere_tree(tsph, &tree_extent1, &tree_nodes);
    // the filter's per-ray error bound must cover the boxes too
    if (static_cast<float>(tree_extent1) * 1.0001f > p1max) p1max = static_cast<float>(tree_extent1) * 1.0001f;
    // anchor grids: the lights, then the camera (the origin, src/view_frustum.h:40).  Built for
    // rays whose origin lies within 255 scene extents of the scene.
    rt::AnchorGrids grids;
    if (!tsph.empty()) {
        std::vector<rt::GridSphere> gs;
        gs.reserve(tsph.size());
        for (const rt::TreeSphere& t : tsph)
            gs.push_back(rt::GridSphere{{t.c[0], t.c[1], t.c[2]}, prims[t.index].q[0] * prims[t.index].q[0], t.index,
                                        static_cast<float>(prims[t.index].transparency)});
        std::vector<std::array<double, 3>> anchors;
        for (int li = 0; li < n_lights; ++li) anchors.push_back({lights[li].pos[0], lights[li].pos[1], lights[li].pos[2]});
        anchors.push_back({0.0, 0.0, 0.0});
        int res = rt::anchor_grid_resolution(gs.size(), anchors.size());
        if (const char* env = std::getenv("RT_GRID_RES")) {  // measurement aid
            const int v = std::atoi(env);
            if (v >= 1 && v <= 1024) res = v;
        }
        rt::build_anchor_grids(gs, anchors, 256.0 * static_cast<double>(p1max), res, size_t(8) << 20, &grids);
    }
    // the table of the pass that finishes pixels ending at the primary ray (FrontPlane /
    // FrontAnchor in whitted_device.h): small scenes only, the pass tests every object
    std::vector<float> front;
    const bool want_front = tsph.size() <= kFrontMaxSpheres && fplanes.size() <= kFrontMaxPlanes;
    if (want_front) {
        auto put = [&front](const void* rec, size_t bytes) {
            const size_t at = front.size();
            front.resize(at + bytes / sizeof(float));
            std::memcpy(&front[at], rec, bytes);
        };
        size_t n_fp = 0;
        for (int i = 0; i < n_prims; ++i) {
            if (prims[i].type != RT_PRIM_PLANE) continue;
            const rt_prim& s = prims[i];
            rt::FrontPlane fp;
            std::memset(&fp, 0, sizeof fp);
            const double hd = s.p[0] * s.q[0] + s.p[1] * s.q[1] + s.p[2] * s.q[2];
            for (int c = 0; c < 3; ++c) {
                fp.n[c] = static_cast<float>(mat[i].plane_normal[c]);
                fp.q[c] = s.q[c];
                fp.kd[c] = mat[i].kd[c];
            }
            fp.ends = (mat[i].reflectivity == 0.0f && mat[i].transparency == 0.0f) ? 1.0f : 0.0f;
            fp.h = fgeom[i].v[4];
            fp.herr = fgeom[i].v[5];
            fp.hd = hd;
            fp.index = i;
            fp.transparency = static_cast<float>(s.transparency);
            fp.ambient = mat[i].ambient_intensity;
            put(&fp, sizeof fp);
            ++n_fp;
        }
        if (n_fp & 1) {
            rt::FrontPlane fp;
            std::memset(&fp, 0, sizeof fp);
            fp.index = 0x7fffffff;
            put(&fp, sizeof fp);
        }
        for (int ai = 0; ai <= n_lights; ++ai) {
            rt::FrontAnchor fa;
            std::memset(&fa, 0, sizeof fa);
            if (ai < n_lights)
                for (int c = 0; c < 3; ++c) {
                    fa.pos[c] = lt[ai].pos[c];
                    fa.ci[c] = lt[ai].ci[c];
                }
            put(&fa, sizeof fa);
            size_t n_e = 0;
            for (const rt::TreeSphere& t : tsph) {  // container order
                const double L[3] = {t.c[0] - fa.pos[0], t.c[1] - fa.pos[1], t.c[2] - fa.pos[2]};
                rt::GridEntry e;
                std::memset(&e, 0, sizeof e);
                float rec[4], cold[2];
                rt::line_test_record(L, prims[t.index].q[0] * prims[t.index].q[0], rec, cold);
                e.l[0] = rec[0]; e.l[1] = rec[1]; e.l[2] = rec[2];
                e.thr = rec[3];
                e.index = t.index;
                e.r2 = cold[0];
                e.ehit = cold[1];
                e.transparency = static_cast<float>(prims[t.index].transparency);
                put(&e, sizeof e);
                ++n_e;
            }
            for (; n_e < 2 || (n_e & 1); ++n_e) {  // entries that never hit
                rt::GridEntry e;
                std::memset(&e, 0, sizeof e);
                e.thr = INFINITY;
                e.index = 0x7fffffff;
                put(&e, sizeof e);
            }
        }
    }
    DeviceGuard guard(ctx->device);
    RT_HIP_TRY(ctx, guard.status);
    ctx->have_scene = false;
    if (ctx->d_front) RT_HIP_TRY(ctx, hipFree(ctx->d_front));
    ctx->d_front = nullptr;
    ctx->n_front = 0;
    if (want_front) {
        RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_front), front.size() * sizeof(float)));
        RT_HIP_TRY(ctx, hipMemcpy(ctx->d_front, front.data(), front.size() * sizeof(float), hipMemcpyHostToDevice));
        ctx->n_front = static_cast<int32_t>(tsph.size());
    }
    if (ctx->d_grid_cells) RT_HIP_TRY(ctx, hipFree(ctx->d_grid_cells));
    ctx->d_grid_cells = nullptr;
    if (ctx->d_grid_entries) RT_HIP_TRY(ctx, hipFree(ctx->d_grid_entries));
    ctx->d_grid_entries = nullptr;
    ctx->grid_res = 0;
    if (grids.res > 0) {
        const size_t cb = grids.cells.size() * sizeof(rt::GridCell);
        const size_t eb = (grids.entries.size() + 1) * sizeof(rt::GridEntry);  // never empty
        RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_grid_cells), cb));
        RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_grid_entries), eb));
        RT_HIP_TRY(ctx, hipMemcpy(ctx->d_grid_cells, grids.cells.data(), cb, hipMemcpyHostToDevice));
        if (!grids.entries.empty())
            RT_HIP_TRY(ctx, hipMemcpy(ctx->d_grid_entries, grids.entries.data(), eb - sizeof(rt::GridEntry), hipMemcpyHostToDevice));
        ctx->grid_res = grids.res;
        ctx->grid_eray_max = grids.eray_max;
    }
    if (ctx->f_geom) RT_HIP_TRY(ctx, hipFree(ctx->f_geom));
    ctx->f_geom = nullptr;
    if (ctx->f_tree) RT_HIP_TRY(ctx, hipFree(ctx->f_tree));
    if (ctx->f_planes) RT_HIP_TRY(ctx, hipFree(ctx->f_planes));
    if (ctx->d_planes_before) RT_HIP_TRY(ctx, hipFree(ctx->d_planes_before));
    if (ctx->d_ltab) RT_HIP_TRY(ctx, hipFree(ctx->d_ltab));
    ctx->f_tree = nullptr; ctx->f_planes = nullptr; ctx->d_planes_before = nullptr; ctx->d_ltab = nullptr;
    ctx->n_nodes = 0; ctx->n_planes = 0;
    RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_ltab), ltab.size() * sizeof(float)));
    RT_HIP_TRY(ctx, hipMemcpy(ctx->d_ltab, ltab.data(), ltab.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->ltab_stride = static_cast<int32_t>(lstride);
    if (!tree.empty()) {
        RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->f_tree), tree.size() * sizeof(rt::FGeom)));
        RT_HIP_TRY(ctx, hipMemcpy(ctx->f_tree, tree.data(), tree.size() * sizeof(rt::FGeom), hipMemcpyHostToDevice));
    }
    if (!fplanes.empty()) {
        RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->f_planes), fplanes.size() * sizeof(rt::FGeom)));
        RT_HIP_TRY(ctx, hipMemcpy(ctx->f_planes, fplanes.data(), fplanes.size() * sizeof(rt::FGeom), hipMemcpyHostToDevice));
    }
    RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_planes_before), planes_before.size() * sizeof(int32_t)));
    RT_HIP_TRY(ctx, hipMemcpy(ctx->d_planes_before, planes_before.data(), planes_before.size() * sizeof(int32_t),
                              hipMemcpyHostToDevice));
    ctx->n_nodes = tree_nodes;
    ctx->n_planes = static_cast<int32_t>(fplanes.size());
    RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->f_geom), fgeom.size() * sizeof(rt::FGeom)));
    RT_HIP_TRY(ctx, hipMemcpy(ctx->f_geom, fgeom.data(), fgeom.size() * sizeof(rt::FGeom), hipMemcpyHostToDevice));
    ctx->scene_p1max = p1max;
    if (ctx->d_geom) RT_HIP_TRY(ctx, hipFree(ctx->d_geom));
    if (ctx->d_mat) RT_HIP_TRY(ctx, hipFree(ctx->d_mat));
    if (ctx->d_lights) RT_HIP_TRY(ctx, hipFree(ctx->d_lights));
    ctx->d_geom = nullptr; ctx->d_mat = nullptr; ctx->d_lights = nullptr;
    RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_geom), geom.size() * sizeof(rt::DGeom)));
    RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_mat), mat.size() * sizeof(rt::DMat)));
    RT_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_lights), lt.size() * sizeof(rt::DLight)));
    RT_HIP_TRY(ctx, hipMemcpy(ctx->d_geom, geom.data(), geom.size() * sizeof(rt::DGeom), hipMemcpyHostToDevice));
    RT_HIP_TRY(ctx, hipMemcpy(ctx->d_mat, mat.data(), mat.size() * sizeof(rt::DMat), hipMemcpyHostToDevice));
    RT_HIP_TRY(ctx, hipMemcpy(ctx->d_lights, lt.data(), lt.size() * sizeof(rt::DLight), hipMemcpyHostToDevice));
    ctx->n_prims = n_prims;
    ctx->n_lights = n_lights;
    ctx->host_prims.assign(prims, prims + n_prims);
    ctx->sky_cache.clear();
    ctx->have_scene = true;
    return RT_OK;
}

// Screen bounds of the scene for the partition in `p` (image-only launches with one wave per
// tile of scenes small enough that working them out costs less than they save); null = none.
constexpr int kSkyMaxPrims = 64;
constexpr size_t kSkyCacheEntries = 16;
static const rt_ctx::SkyEntry* sky_bounds(rt_ctx* ctx, const rt_params* p, const rt::KArgs& a) {
    if (ctx->n_prims > kSkyMaxPrims || ctx->host_prims.empty()) return nullptr;
    for (const rt_ctx::SkyEntry& e : ctx->sky_cache)
        if (e.p.width == p->width && e.p.height == p->height && e.p.band_rows == p->band_rows &&
            e.p.band_first == p->band_first && e.p.band_step == p->band_step && e.p.band_count == p->band_count &&
            e.p.fov_deg == p->fov_deg)
            return &e;
    rt_ctx::SkyEntry e;
    e.p = *p;
    rt::SkyTable by_row;
    uint32_t sphere_tiles[rt::kSkyRows];
    rt::build_sky_table(ctx->host_prims.data(), ctx->n_prims, *p, a.rows, a.pw, a.ph, &by_row, sphere_tiles);
    // the order of the launch's block rows, and the table laid out in that order
    e.row_order = (a.tiles_y <= rt::kSkyRows) ? rt::tile_row_order(sphere_tiles, a.tiles_y) : 0;
    if (const char* env = std::getenv("RT_ROW_ORDER")) e.row_order = std::atoi(env);  // measurement aid
    e.t = by_row;
    if (a.tiles_y <= rt::kSkyRows)
        for (int b = 0; b < a.tiles_y; ++b) e.t.cols[b] = by_row.cols[rt::tile_row_of_block(e.row_order, b, a.tiles_y)];
    if (ctx->sky_cache.size() < kSkyCacheEntries) {
        ctx->sky_cache.push_back(e);
        return &ctx->sky_cache.back();
    }
    ctx->sky_next = (ctx->sky_next + 1) % kSkyCacheEntries;
    ctx->sky_cache[ctx->sky_next] = e;
    return &ctx->sky_cache[ctx->sky_next];
}

// Enqueue one render with device outputs on `stream`; the deep-stack workspace is `ws` when
// given (the built-in one) and this stream's own otherwise.  The caller has made the
// context's device current.
static int enqueue_render(rt_ctx* ctx, const rt_params* p, hipStream_t stream, void* ws, uint8_t* d_bgr,
                          int32_t* d_hit_id, uint32_t* d_hit_mask, uint32_t* d_spawn_mask, rt_counters* d_counters) {
    rt::KArgs a;
    int rc = make_args(ctx, p, a);
    if (rc != RT_OK) return rc;
    const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
    if (count && !d_counters) return fail(ctx, RT_ERR_ARG, "RT_FLAG_COUNTERS set but counters is null");
    a.bgr = d_bgr;
    a.hit_id = d_hit_id;
    a.hit_mask = d_hit_mask;
    a.spawn_mask = d_spawn_mask;
    a.counters = reinterpret_cast<unsigned long long*>(d_counters);
    const bool emit = d_hit_id || d_hit_mask || d_spawn_mask;
    int deep_blocks = 0;
    const bool tree = use_tree(ctx, p);
    if (a.tiles_x > 0 && a.tiles_y > 0 && (tree || rt::stack_global_entries(a.max_depth) > 0)) {
        if (!ws && (rc = stream_workspace(ctx, stream, rt::stack_global_entries(a.max_depth) > 0, &ws)) != RT_OK) return rc;
        if ((rc = attach_workspace(ctx, &a, 1, static_cast<long long>(a.tiles_x) * a.tiles_y, a.max_depth, use_grid(ctx, p),
                                   ws, &deep_blocks)) != RT_OK)
            return rc;
    }
    const rt::SkyTable* sky = nullptr;
    if (!emit && !count && !tree && rt::stack_global_entries(a.max_depth) == 0 && !(p->flags & RT_FLAG_NOSKY)) {
        if (const rt_ctx::SkyEntry* e = sky_bounds(ctx, p, a)) {
            sky = &e->t;
            a.row_order = e->row_order;
        }
    }
    RT_HIP_TRY(ctx, rt::launch_whitted(a, emit, count, tree, use_grid(ctx, p), deep_blocks, stream, sky));
    return RT_OK;
}

int rt_render_device(rt_ctx* ctx, const rt_params* p, void* stream, uint8_t* d_bgr, int32_t* d_hit_id,
                     uint32_t* d_hit_mask, uint32_t* d_spawn_mask, rt_counters* d_counters) {
    if (!ctx) return RT_ERR_ARG;
    if (!d_bgr) return fail(ctx, RT_ERR_ARG, "rt_render_device: d_bgr is null");
    DeviceGuard guard(ctx->device);
    RT_HIP_TRY(ctx, guard.status);
    return enqueue_render(ctx, p, static_cast<hipStream_t>(stream), nullptr, d_bgr, d_hit_id, d_hit_mask, d_spawn_mask,
                          d_counters);
}

int rt_render_device_static(rt_ctx* ctx, const rt_params* p, void* stream, uint8_t* d_bgr, int32_t* d_hit_id,
                            uint32_t* d_hit_mask, uint32_t* d_spawn_mask, rt_counters* d_counters) {
    if (!ctx) return RT_ERR_ARG;
    if (!d_bgr) return fail(ctx, RT_ERR_ARG, "rt_render_device_static: d_bgr is null");
    // the built-in workspace holds the stacks of kDeepMaxBlocks blocks at RT_MAX_DEPTH;
    // a depth it has no room for is refused here (and by the parameter check for every
    // entry point while RT_MAX_DEPTH is the library's limit)
    if (p && rt::stack_global_entries(p->max_depth) * rt::kStackFields * 64 * sizeof(double) > rt::kDeepRegionBytes)
        return fail(ctx, RT_ERR_ARG, "rt_render_device_static: max_depth needs more stack than the built-in "
                                     "workspace holds (limit: max_depth 8)");
    DeviceGuard guard(ctx->device);
    RT_HIP_TRY(ctx, guard.status);
    // the built-in workspace's address on this device (looked up at the first such call)
    void* ws = nullptr;
    RT_HIP_TRY(ctx, rt::static_workspace(ctx->device, &ws));
    return enqueue_render(ctx, p, static_cast<hipStream_t>(stream), ws, d_bgr, d_hit_id, d_hit_mask, d_spawn_mask,
                          d_counters);
}

int rt_render_device_batch(rt_ctx* const* ctxs, const rt_params* params, int32_t n, void* stream,
                           uint8_t* const* d_bgr) {
    if (!ctxs || n < 1 || !ctxs[0]) return RT_ERR_ARG;
    rt_ctx* first = ctxs[0];
    if (!params || !d_bgr) return fail(first, RT_ERR_ARG, "rt_render_device_batch: null argument");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !d_bgr[i]) return fail(first, RT_ERR_ARG, "rt_render_device_batch: null context or output");
        if (ctxs[i]->device != first->device)
            return fail(first, RT_ERR_ARG, "rt_render_device_batch: contexts on different devices");
        if (params[i].flags & RT_FLAG_COUNTERS)
            return fail(first, RT_ERR_ARG, "rt_render_device_batch: RT_FLAG_COUNTERS is not supported");
        if (use_tree(ctxs[i], &params[i]) != use_tree(first, &params[0]))
            return fail(first, RT_ERR_ARG, "rt_render_device_batch: scenes with and without a sphere tree in one batch");
    }
    const bool tree = use_tree(first, &params[0]);
    bool grid = true;  // only when every render of the batch has its grids
    for (int i = 0; i < n; ++i) grid = grid && use_grid(ctxs[i], &params[i]);
    DeviceGuard guard(first->device);
    RT_HIP_TRY(first, guard.status);
    for (int base = 0; base < n; base += rt::kMaxBatch) {
        const int m = (n - base < rt::kMaxBatch) ? n - base : rt::kMaxBatch;
        rt::KBatch b;
        std::memset(&b, 0, sizeof b);
        int depth = 0, live = 0;
        long long tiles = 0;
        for (int i = 0; i < m; ++i) {
            rt::KArgs a;
            int rc = make_args(ctxs[base + i], &params[base + i], a);
            if (rc != RT_OK) {
                if (ctxs[base + i] != first) first->err = ctxs[base + i]->err;
                return rc;
            }
            if (a.tiles_x <= 0 || a.tiles_y <= 0) continue;  // an empty partition
            a.bgr = d_bgr[base + i];
            // launches with one wave per tile: the rows with the long ray trees first
            if (!tree && !(params[base + i].flags & RT_FLAG_NOSKY))
                if (const rt_ctx::SkyEntry* e = sky_bounds(ctxs[base + i], &params[base + i], a)) a.row_order = e->row_order;
            tiles += static_cast<long long>(a.tiles_x) * a.tiles_y;
            if (tiles > INT32_MAX) return fail(first, RT_ERR_ARG, "rt_render_device_batch: too many tiles in one launch");
            b.a[live] = a;
            b.tile_end[live] = static_cast<int32_t>(tiles);
            depth = a.max_depth > depth ? a.max_depth : depth;
            ++live;
        }
        if (live == 0) continue;
        // the launch uses the first context's workspace of this stream
        int deep_blocks = 0;
        if (tree || rt::stack_global_entries(depth) > 0) {
            void* ws = nullptr;
            int rcg = stream_workspace(first, stream, rt::stack_global_entries(depth) > 0, &ws);
            if (rcg == RT_OK)
                rcg = attach_workspace(first, b.a, live, tiles, depth, grid, ws, &deep_blocks);
            if (rcg != RT_OK) return rcg;
        }
        RT_HIP_TRY(first, rt::launch_whitted_batch(b, live, tree, grid, deep_blocks, static_cast<hipStream_t>(stream)));
    }
    return RT_OK;
}

}  // extern "C"

namespace {

// A partition cut into consecutive parts (whole bands each), so that the copy of one part
// to the host can run while the next part is rendered.  Part i holds strip rows
// [row0, row0 + rows) of the partition it was cut from.
struct Part {
    rt_params p;
    size_t row0, rows;
};
constexpr int kMaxParts = 8;
constexpr size_t kMinPartPixels = 128 * 1024;  // smaller parts cost more in launches than they hide

// reband: a partition of one band may be cut into smaller bands (only when the caller stores
// the strip packed).
std::vector<Part> split_partition(const rt_params& p, bool reband) {
    std::vector<Part> parts;
    const size_t rows = static_cast<size_t>(rt_strip_rows(&p));
    const size_t width = static_cast<size_t>(p.width);
    int want = static_cast<int>(rows * width / kMinPartPixels);
    if (want > kMaxParts) want = kMaxParts;
    // bands that really hold rows
    int nb = 0;
    for (int b = 0; b < p.band_count; ++b) {
        if ((static_cast<int64_t>(p.band_first) + static_cast<int64_t>(b) * p.band_step) * p.band_rows >= p.height) break;
        ++nb;
    }
    if (want >= 2 && nb >= 2) {
        // groups of consecutive bands
        const int n = want < nb ? want : nb;
        size_t row0 = 0;
        for (int j = 0; j < n; ++j) {
            const int b0 = static_cast<int>(static_cast<int64_t>(nb) * j / n);
            const int b1 = static_cast<int>(static_cast<int64_t>(nb) * (j + 1) / n);
            Part q;
            q.p = p;
            q.p.band_first = p.band_first + b0 * p.band_step;
            q.p.band_count = b1 - b0;
            q.row0 = row0;
            q.rows = static_cast<size_t>(rt_strip_rows(&q.p));
            row0 += q.rows;
            parts.push_back(q);
        }
        return parts;
    }
    if (reband && want >= 2 && nb == 1 && p.band_first == 0) {
        // one band from the bottom row on (the whole frame): cut it into bands of whole tile
        // rows; pixels do not depend on how the rows are grouped
        size_t per = (rows + want - 1) / want;
        per = (per + rt::kTile - 1) / rt::kTile * rt::kTile;
        size_t row0 = 0;
        for (int j = 0; row0 < rows; ++j) {
            Part q;
            q.p = p;
            q.p.band_rows = static_cast<int32_t>(per);
            q.p.band_first = j;
            q.p.band_step = 1;
            q.p.band_count = 1;
            q.row0 = row0;
            q.rows = (rows - row0 < per) ? rows - row0 : per;
            // the last part is clipped by the frame; a band that ends inside the frame
            // (rows < height) is clipped here
            if (static_cast<size_t>(rt_strip_rows(&q.p)) != q.rows) { parts.clear(); break; }
            row0 += q.rows;
            parts.push_back(q);
        }
        if (!parts.empty()) return parts;
    }
    Part whole;
    whole.p = p;
    whole.row0 = 0;
    whole.rows = rows;
    parts.push_back(whole);
    return parts;
}

int part_event(rt_ctx* ctx, size_t i, hipEvent_t* ev) {
    while (ctx->part_done.size() <= i) {
        hipEvent_t e = nullptr;
        RT_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->part_done.push_back(e);
    }
    *ev = ctx->part_done[i];
    return RT_OK;
}

// Image-only render of partition `p` into the context's staging buffer, part by part, each
// part copied to the host as soon as it is done (on the context's second stream).  Strip row
// r of the partition goes to host + dst_row(r) bytes, where bands are `band_pitch` bytes
// apart on the host (band_pitch = bytes of one band: the strip is stored packed).  Nothing
// is waited for: the caller synchronises ctx->copy_stream.
int enqueue_render_to_host(rt_ctx* ctx, const rt_params& p, uint8_t* host, size_t band_pitch, rt_counters* d_counters) {
    const size_t width = static_cast<size_t>(p.width);
    const size_t rows = static_cast<size_t>(rt_strip_rows(&p));
    if (rows == 0) return RT_OK;
    int rc;
    if ((rc = reserve(ctx, ctx->bgr, rows * width * 3 + 4)) != RT_OK) return rc;
    uint8_t* d_bgr = static_cast<uint8_t*>(ctx->bgr.ptr);
    const size_t band_bytes = static_cast<size_t>(p.band_rows) * width * 3;
    const bool packed = band_pitch == band_bytes;
    const std::vector<Part> parts = split_partition(p, packed);
    for (size_t i = 0; i < parts.size(); ++i) {
        const Part& q = parts[i];
        uint8_t* d_part = d_bgr + q.row0 * width * 3;
        if ((rc = enqueue_render(ctx, &q.p, ctx->stream, nullptr, d_part, nullptr, nullptr, nullptr, d_counters)) != RT_OK)
            return rc;
        hipEvent_t ev;
        if ((rc = part_event(ctx, i, &ev)) != RT_OK) return rc;
        RT_HIP_TRY(ctx, hipEventRecord(ev, ctx->stream));
        RT_HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ev, 0));
        if (packed) {
            RT_HIP_TRY(ctx, hipMemcpyAsync(host + q.row0 * width * 3, d_part, q.rows * width * 3, hipMemcpyDeviceToHost,
                                           ctx->copy_stream));
        } else {
            // whole bands of the part with one strided copy, a clipped last band on its own
            const size_t band0 = q.row0 / static_cast<size_t>(p.band_rows);  // parts start on band boundaries
            const size_t full = q.rows / static_cast<size_t>(p.band_rows);
            const size_t tail = q.rows - full * static_cast<size_t>(p.band_rows);
            if (full)
                RT_HIP_TRY(ctx, hipMemcpy2DAsync(host + band0 * band_pitch, band_pitch, d_part, band_bytes, band_bytes, full,
                                                 hipMemcpyDeviceToHost, ctx->copy_stream));
            if (tail)
                RT_HIP_TRY(ctx, hipMemcpyAsync(host + (band0 + full) * band_pitch, d_part + full * band_bytes,
                                               tail * width * 3, hipMemcpyDeviceToHost, ctx->copy_stream));
        }
    }
    return RT_OK;
}

// ---- rt_render into ordinary host memory --------------------------------------------------
// A device-to-host copy into memory that is not page-locked is staged by the HIP runtime and
// runs at about a third of the PCIe rate (4K frame: 1.7 ms against 0.5 ms).  rt_render does the
// staging itself: every part of the frame goes by DMA into a page-locked buffer of the context
// while the next part is rendered, and as parts arrive a few threads copy them into the caller's
// memory, which the memory system does faster than the link delivers.
class HostCopyPool {
  public:
    static HostCopyPool& get() {
        static HostCopyPool pool;
        return pool;
    }
    // dst[0, bytes) = src[0, bytes), cut into chunks that the workers and the caller share;
    // returns when all of it is copied.
    void copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
        constexpr size_t kChunk = size_t(256) << 10;
        if (bytes <= 2 * kChunk || workers_.empty()) {
            std::memcpy(dst, src, bytes);
            return;
        }
        Job job;
        {
            std::lock_guard<std::mutex> lock(m_);
            for (size_t off = 0; off < bytes; off += kChunk) {
                tasks_.push_back(Task{dst + off, src + off, bytes - off < kChunk ? bytes - off : kChunk, &job});
                ++job.left;
            }
        }
        cv_.notify_all();
        // the caller copies too, then waits for what the workers still hold
        for (;;) {
            Task t;
            {
                std::lock_guard<std::mutex> lock(m_);
                if (tasks_.empty()) break;
                t = tasks_.front();
                tasks_.pop_front();
            }
            run(t);
        }
        std::unique_lock<std::mutex> lock(m_);
        job.done.wait(lock, [&job] { return job.left == 0; });
    }

  private:
    struct Job {
        size_t left = 0;  // guarded by m_
        std::condition_variable done;
    };
    struct Task {
        uint8_t* dst;
        const uint8_t* src;
        size_t bytes;
        Job* job;
    };
    HostCopyPool() {
        unsigned n = std::thread::hardware_concurrency();
        n = n >= 8 ? 3 : (n >= 4 ? 2 : (n >= 2 ? 1 : 0));  // the caller is one more
        for (unsigned i = 0; i < n; ++i) workers_.emplace_back([this] { work(); });
    }
    ~HostCopyPool() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    void run(const Task& t) {
        std::memcpy(t.dst, t.src, t.bytes);
        std::lock_guard<std::mutex> lock(m_);
        if (--t.job->left == 0) t.job->done.notify_all();
    }
    void work() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [this] { return stop_ || !tasks_.empty(); });
                if (tasks_.empty()) return;  // stop_
                t = tasks_.front();
                tasks_.pop_front();
            }
            run(t);
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Task> tasks_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};

// Whether `p` is host memory the HIP runtime knows nothing about (neither allocated by it nor
// registered with it): a copy into it would be staged by the runtime.
bool is_ordinary_host_memory(const void* p) {
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof attr);
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // "invalid value" is how older runtimes answer for such a pointer
        return true;
    }
    return attr.type == hipMemoryTypeUnregistered;
}

constexpr size_t kStageMinBytes = size_t(2) << 20;  // smaller frames are not worth the threads

// The image-only render of a packed partition into ordinary host memory (see above).  Returns
// with the frame complete in `host`.
int render_to_ordinary_memory(rt_ctx* ctx, const rt_params& p, uint8_t* host, rt_counters* d_counters) {
    const size_t width = static_cast<size_t>(p.width);
    const size_t rows = static_cast<size_t>(rt_strip_rows(&p));
    const size_t bytes = rows * width * 3;
    int rc;
    if ((rc = reserve(ctx, ctx->bgr, bytes + 4)) != RT_OK) return rc;
    if (ctx->host_stage_bytes < bytes) {
        if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
        ctx->host_stage = nullptr;
        ctx->host_stage_bytes = 0;
        RT_HIP_TRY(ctx, hipHostMalloc(&ctx->host_stage, bytes, hipHostMallocDefault));
        ctx->host_stage_bytes = bytes;
    }
    uint8_t* d_bgr = static_cast<uint8_t*>(ctx->bgr.ptr);
    uint8_t* stage = static_cast<uint8_t*>(ctx->host_stage);
    const std::vector<Part> parts = split_partition(p, true);
    while (ctx->part_arrived.size() < parts.size()) {
        hipEvent_t e = nullptr;
        RT_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->part_arrived.push_back(e);
    }
    size_t enqueued = 0;
    rc = RT_OK;
    for (; enqueued < parts.size(); ++enqueued) {
        const Part& q = parts[enqueued];
        const size_t off = q.row0 * width * 3;
        if ((rc = enqueue_render(ctx, &q.p, ctx->stream, nullptr, d_bgr + off, nullptr, nullptr, nullptr, d_counters)) != RT_OK) break;
        hipEvent_t ev;
        if ((rc = part_event(ctx, enqueued, &ev)) != RT_OK) break;
        hipError_t e = hipEventRecord(ev, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, ev, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(stage + off, d_bgr + off, q.rows * width * 3, hipMemcpyDeviceToHost, ctx->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->part_arrived[enqueued], ctx->copy_stream);
        if (e != hipSuccess) { rc = hip_fail(ctx, e, "rt_render: copy of a part"); break; }
    }
    if (rc != RT_OK) {
        // (nothing writes into the caller's memory but this thread; the streams are drained so
        // that the staging buffer is free for the next call)
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->copy_stream);
        return rc;
    }
    for (size_t i = 0; i < parts.size(); ++i) {
        RT_HIP_TRY(ctx, hipEventSynchronize(ctx->part_arrived[i]));
        const size_t off = parts[i].row0 * width * 3;
        HostCopyPool::get().copy(host + off, stage + off, parts[i].rows * width * 3);
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_render(rt_ctx* ctx, const rt_params* p, uint8_t* bgr, int32_t* hit_id, uint32_t* hit_mask,
              uint32_t* spawn_mask, rt_counters* counters) {
    if (!ctx) return RT_ERR_ARG;
    if (!bgr) return fail(ctx, RT_ERR_ARG, "rt_render: bgr is null");
    if (check_params(p) != RT_OK) return fail(ctx, RT_ERR_ARG, "rt_params: value out of range");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
    if (count && !counters) return fail(ctx, RT_ERR_ARG, "RT_FLAG_COUNTERS set but counters is null");
    const size_t px = static_cast<size_t>(rt_strip_rows(p)) * p->width;
    const size_t words = rt_mask_words(p->max_depth);
    DeviceGuard guard(ctx->device);
    RT_HIP_TRY(ctx, guard.status);
    int rc;
    if (count) {
        if ((rc = reserve(ctx, ctx->counters, sizeof(rt_counters))) != RT_OK) return rc;
        RT_HIP_TRY(ctx, hipMemsetAsync(ctx->counters.ptr, 0, sizeof(rt_counters), ctx->stream));
    }
    rt_counters* d_counters = count ? static_cast<rt_counters*>(ctx->counters.ptr) : nullptr;
    if (!hit_id && !hit_mask && !spawn_mask) {
        // the image alone: rendered in parts, each copied while the next one is rendered
        const size_t band_bytes = static_cast<size_t>(p->band_rows) * p->width * 3;
        if (px * 3 >= kStageMinBytes && is_ordinary_host_memory(bgr)) {
            if ((rc = render_to_ordinary_memory(ctx, *p, bgr, d_counters)) != RT_OK) return rc;
            if (count) {
                RT_HIP_TRY(ctx, hipMemcpyAsync(counters, ctx->counters.ptr, sizeof(rt_counters), hipMemcpyDeviceToHost, ctx->stream));
                RT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            }
            return RT_OK;
        }
        if ((rc = enqueue_render_to_host(ctx, *p, bgr, band_bytes, d_counters)) != RT_OK) {
            // copies of earlier parts may still be writing into the caller's buffer
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ctx->copy_stream);
            return rc;
        }
        if (count) {
            RT_HIP_TRY(ctx, hipMemcpyAsync(counters, ctx->counters.ptr, sizeof(rt_counters), hipMemcpyDeviceToHost, ctx->stream));
            RT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        RT_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
        return RT_OK;
    }
    if ((rc = reserve(ctx, ctx->bgr, px * 3 + 4)) != RT_OK) return rc;
    if (hit_id && (rc = reserve(ctx, ctx->hit_id, px * 4 + 4)) != RT_OK) return rc;
    if (hit_mask && (rc = reserve(ctx, ctx->hit_mask, px * words * 4 + 4)) != RT_OK) return rc;
    if (spawn_mask && (rc = reserve(ctx, ctx->spawn_mask, px * words * 4 + 4)) != RT_OK) return rc;
    rc = enqueue_render(ctx, p, ctx->stream, nullptr, static_cast<uint8_t*>(ctx->bgr.ptr),
                        hit_id ? static_cast<int32_t*>(ctx->hit_id.ptr) : nullptr,
                        hit_mask ? static_cast<uint32_t*>(ctx->hit_mask.ptr) : nullptr,
                        spawn_mask ? static_cast<uint32_t*>(ctx->spawn_mask.ptr) : nullptr, d_counters);
    if (rc != RT_OK) return rc;
    if (px) {
        RT_HIP_TRY(ctx, hipMemcpyAsync(bgr, ctx->bgr.ptr, px * 3, hipMemcpyDeviceToHost, ctx->stream));
        if (hit_id) RT_HIP_TRY(ctx, hipMemcpyAsync(hit_id, ctx->hit_id.ptr, px * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (hit_mask)
            RT_HIP_TRY(ctx, hipMemcpyAsync(hit_mask, ctx->hit_mask.ptr, px * words * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (spawn_mask)
            RT_HIP_TRY(ctx, hipMemcpyAsync(spawn_mask, ctx->spawn_mask.ptr, px * words * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (count)
        RT_HIP_TRY(ctx, hipMemcpyAsync(counters, ctx->counters.ptr, sizeof(rt_counters), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_render_node(rt_ctx* const* ctxs, int32_t n, const rt_params* p, int32_t band_rows, uint8_t* bgr) {
    if (!ctxs || n < 1 || !ctxs[0]) return RT_ERR_ARG;
    rt_ctx* first = ctxs[0];
    if (!bgr) return fail(first, RT_ERR_ARG, "rt_render_node: bgr is null");
    if (check_params(p) != RT_OK) return fail(first, RT_ERR_ARG, "rt_params: value out of range");
    if (band_rows < 1) return fail(first, RT_ERR_ARG, "rt_render_node: band_rows must be positive");
    if (p->flags & RT_FLAG_COUNTERS) return fail(first, RT_ERR_ARG, "rt_render_node: RT_FLAG_COUNTERS is not supported");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(first, RT_ERR_ARG, "rt_render_node: null context");
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(first, RT_ERR_ARG, "rt_render_node: a context is listed twice");
        if (!ctxs[i]->have_scene) {
            ctxs[i]->err = "no scene uploaded";
            return fail(first, RT_ERR_STATE, "rt_render_node: a context has no scene");
        }
    }
    const int nb = (p->height + band_rows - 1) / band_rows;
    const size_t band_bytes = static_cast<size_t>(band_rows) * p->width * 3;
    int rc = RT_OK;
    int started = 0, failed = -1;
    for (int i = 0; i < n && rc == RT_OK; ++i) {
        rt_params q = *p;
        q.band_rows = band_rows;
        q.band_first = i;
        q.band_step = n;
        q.band_count = (i < nb) ? (nb - i + n - 1) / n : 0;
        DeviceGuard guard(ctxs[i]->device);
        if (guard.status != hipSuccess) { rc = hip_fail(ctxs[i], guard.status, "hipSetDevice"); failed = i; break; }
        // band b of the frame starts b * band_bytes into the caller's buffer: this context's
        // bands are n * band_bytes apart, the first one i * band_bytes in
        rc = enqueue_render_to_host(ctxs[i], q, bgr + static_cast<size_t>(i) * band_bytes,
                                    static_cast<size_t>(n) * band_bytes, nullptr);
        if (rc != RT_OK) failed = i;
        started = i + 1;
    }
    // wait for whatever was started, also after a failure: the caller's buffer is in use
    for (int i = 0; i < started; ++i) {
        DeviceGuard guard(ctxs[i]->device);
        const hipError_t e = (guard.status == hipSuccess) ? hipStreamSynchronize(ctxs[i]->copy_stream) : guard.status;
        if (e != hipSuccess && rc == RT_OK) { rc = hip_fail(ctxs[i], e, "hipStreamSynchronize"); failed = i; }
    }
    if (failed > 0) first->err = ctxs[failed]->err;  // errors are reported on ctxs[0]
    return rc;
}

// rt_params that only carry the object-loop mode (the debug hooks read no camera fields).
static rt_params mode_params(uint32_t flags) {
    rt_params p;
    std::memset(&p, 0, sizeof p);
    p.width = p.height = 8; p.max_depth = 2; p.band_rows = 8; p.band_step = 1; p.band_count = 1;
    p.fov_deg = 60.0;
    p.flags = flags & (RT_FLAG_LINEAR | RT_FLAG_TREE | RT_FLAG_NOGRID | RT_FLAG_GRID);
    return p;
}

int rt_debug_nearest(rt_ctx* ctx, const double* rays, int32_t n, uint32_t flags, int32_t* index_out, double* t_out) {
    if (!ctx) return RT_ERR_ARG;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (!rays || n < 0 || !index_out || !t_out) return fail(ctx, RT_ERR_ARG, "rt_debug_nearest: bad argument");
    if (n == 0) return RT_OK;
    const rt_params p = mode_params(flags);
    rt::KArgs a;
    int rc = make_args(ctx, &p, a);
    if (rc != RT_OK) return rc;
    DeviceGuard guard(ctx->device);
    RT_HIP_TRY(ctx, guard.status);
    const size_t N = static_cast<size_t>(n);
    if ((rc = reserve(ctx, ctx->dbg_in, N * 48)) != RT_OK) return rc;
    if ((rc = reserve(ctx, ctx->dbg_t, N * 8)) != RT_OK) return rc;
    if ((rc = reserve(ctx, ctx->dbg_unit, N * 4)) != RT_OK) return rc;
    RT_HIP_TRY(ctx, hipMemcpyAsync(ctx->dbg_in.ptr, rays, N * 48, hipMemcpyHostToDevice, ctx->stream));
    if (flags & RT_DEBUG_RECORDS) {
        if (!a.front) return fail(ctx, RT_ERR_STATE, "rt_debug_nearest: the scene has no records for the single-precision pass");
        RT_HIP_TRY(ctx, rt::launch_debug_records(a, false, 0, -1, static_cast<const double*>(ctx->dbg_in.ptr), n,
                                                 static_cast<int32_t*>(ctx->dbg_unit.ptr), static_cast<double*>(ctx->dbg_t.ptr),
                                                 ctx->stream));
    } else
    RT_HIP_TRY(ctx, rt::launch_debug_nearest(a, use_tree(ctx, &p), use_grid(ctx, &p), static_cast<const double*>(ctx->dbg_in.ptr), n,
                                             static_cast<int32_t*>(ctx->dbg_unit.ptr),
                                             static_cast<double*>(ctx->dbg_t.ptr), ctx->stream));
    RT_HIP_TRY(ctx, hipMemcpyAsync(index_out, ctx->dbg_unit.ptr, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP_TRY(ctx, hipMemcpyAsync(t_out, ctx->dbg_t.ptr, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_debug_occluder(rt_ctx* ctx, const double* points, int32_t n, int32_t light, int32_t self, uint32_t flags,
                      int32_t* first_out, double* factor_out) {
    if (!ctx) return RT_ERR_ARG;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (!points || n < 0 || !first_out || !factor_out || light < 0 || light >= ctx->n_lights || self < -1 ||
        self >= ctx->n_prims)
        return fail(ctx, RT_ERR_ARG, "rt_debug_occluder: bad argument");
    if (n == 0) return RT_OK;
    const rt_params p = mode_params(flags);
    rt::KArgs a;
    int rc = make_args(ctx, &p, a);
    if (rc != RT_OK) return rc;
    DeviceGuard guard(ctx->device);
    RT_HIP_TRY(ctx, guard.status);
    const size_t N = static_cast<size_t>(n);
    if ((rc = reserve(ctx, ctx->dbg_in, N * 24)) != RT_OK) return rc;
    if ((rc = reserve(ctx, ctx->dbg_t, N * 8)) != RT_OK) return rc;
    if ((rc = reserve(ctx, ctx->dbg_unit, N * 4)) != RT_OK) return rc;
    RT_HIP_TRY(ctx, hipMemcpyAsync(ctx->dbg_in.ptr, points, N * 24, hipMemcpyHostToDevice, ctx->stream));
    if (flags & RT_DEBUG_RECORDS) {
        if (!a.front) return fail(ctx, RT_ERR_STATE, "rt_debug_occluder: the scene has no records for the single-precision pass");
        RT_HIP_TRY(ctx, rt::launch_debug_records(a, true, light, self, static_cast<const double*>(ctx->dbg_in.ptr), n,
                                                 static_cast<int32_t*>(ctx->dbg_unit.ptr), static_cast<double*>(ctx->dbg_t.ptr),
                                                 ctx->stream));
    } else
    RT_HIP_TRY(ctx, rt::launch_debug_occluder(a, use_tree(ctx, &p), use_grid(ctx, &p), light, self,
                                              static_cast<const double*>(ctx->dbg_in.ptr), n,
                                              static_cast<int32_t*>(ctx->dbg_unit.ptr),
                                              static_cast<double*>(ctx->dbg_t.ptr), ctx->stream));
    RT_HIP_TRY(ctx, hipMemcpyAsync(first_out, ctx->dbg_unit.ptr, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP_TRY(ctx, hipMemcpyAsync(factor_out, ctx->dbg_t.ptr, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_debug_intersect(rt_ctx* ctx, const double* rays, int32_t n, int32_t prim, double* t_out, uint8_t* hit_out,
                       double* unit_out) {
    if (!ctx) return RT_ERR_ARG;
    if (!ctx->have_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (!rays || n < 0 || !t_out || !hit_out || prim < 0 || prim >= ctx->n_prims)
        return fail(ctx, RT_ERR_ARG, "rt_debug_intersect: bad argument");
    if (n == 0) return RT_OK;
    DeviceGuard guard(ctx->device);
    RT_HIP_TRY(ctx, guard.status);
    int rc;
    const size_t N = static_cast<size_t>(n);
    if ((rc = reserve(ctx, ctx->dbg_in, N * 48)) != RT_OK) return rc;
    if ((rc = reserve(ctx, ctx->dbg_t, N * 8)) != RT_OK) return rc;
    if ((rc = reserve(ctx, ctx->dbg_hit, N)) != RT_OK) return rc;
    if (unit_out && (rc = reserve(ctx, ctx->dbg_unit, N * 24)) != RT_OK) return rc;
    RT_HIP_TRY(ctx, hipMemcpyAsync(ctx->dbg_in.ptr, rays, N * 48, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP_TRY(ctx, rt::launch_debug_intersect(ctx->d_geom, ctx->f_geom, ctx->scene_p1max, prim, static_cast<const double*>(ctx->dbg_in.ptr), n,
                                               static_cast<double*>(ctx->dbg_t.ptr),
                                               static_cast<uint8_t*>(ctx->dbg_hit.ptr),
                                               unit_out ? static_cast<double*>(ctx->dbg_unit.ptr) : nullptr,
                                               ctx->stream));
    RT_HIP_TRY(ctx, hipMemcpyAsync(t_out, ctx->dbg_t.ptr, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP_TRY(ctx, hipMemcpyAsync(hit_out, ctx->dbg_hit.ptr, N, hipMemcpyDeviceToHost, ctx->stream));
    if (unit_out)
        RT_HIP_TRY(ctx, hipMemcpyAsync(unit_out, ctx->dbg_unit.ptr, N * 24, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"
