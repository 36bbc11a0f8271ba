// Device-side data layout and launch interface of the Whitted kernels (gfx950 only).
//
// Scene records are 64 bytes so that one record is one s_load_dwordx16: inside the
// object loops the record index is wave-uniform, the record lives in SGPRs and every
// lane tests its own ray against the broadcast primitive.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rt_hip.h"

namespace rt {

// Geometry of one collider + the one material value a shadow feeler needs.
// r2 < 0 marks a plane (q = unit normal as stored by the Plane constructor);
// otherwise a sphere with r2 = radius*radius (std::pow(radius_,2), src/objects.h:125) and
// q = {radius, 0, 0} (not read by the kernels).
struct DGeom {
    double p[3];
    double q[3];
    double r2;
    float transparency;  // colour only: what an occluder lets through (src/objects.h:44)
    float pad;
};

// Material of one collider, gathered per lane once per hit.  The refraction index decides
// whether a refraction ray exists (src/main.cc:86-87) and stays double; everything that only
// scales colours is single precision.
struct DMat {
    double refraction_index;
    double inv_refraction_index;  // 1.0 / refraction_index (src/main.cc:64)
    float kd[3];
    float reflectivity;
    float transparency;
    float ambient_intensity;
    double plane_normal[3];  // planes: calcNormal's result, makeNormal(normal_) (src/objects.h:104-106),
                             // the same for every hit and therefore worked out once at upload
};

// One point light; ci = light_color * double(intensity) (src/objects.h:54-56), rounded to
// single precision (colour only).
struct DLight {
    double pos[3];
    float ci[3];
    float pad[7];
};

static_assert(sizeof(DGeom) == 64 && sizeof(DMat) == 64 && sizeof(DLight) == 64, "record size");

// Float image of the geometry for the certain-miss filter (whitted_exact.inc.h): 32-byte
// record, one s_load_dwordx8 at a wave-uniform index:
//   sphere  {centre.xyz, r2        | cerr = 2^-24 |centre|_1 (rounded up), 0, 0, 0}
//   plane   {normal.xyz, -1 (mark) | h = float(origin.normal), herr,       0, 0}
// with herr bounding |h - origin.normal| and the float rounding of the normal applied to
// the origin.  The table is padded to a multiple of four records (the kernels read it in
// groups of four).
struct FGeom {
    float v[8];
};
static_assert(sizeof(FGeom) == 32, "record size");

// Anchor grids (anchor_grid.h): for every light, and for the camera, the lines through that
// point are sorted into the cells of three cube faces; a cell lists the spheres some line of
// the cell may touch, in container order.  A shadow feeler (its line passes through its
// light) and a primary ray (through the camera) then test the few spheres of one cell
// instead of walking the scene.
struct GridCell {
    uint32_t start, count;   // entries [start, start + count) of the entry table
};
struct GridEntry {
    float l[3];              // sphere centre - anchor
    float thr;               // threshold of the light-anchored line test (line_miss_from_light)
    int32_t index;           // the sphere's position in the caller's object list
    float r2, ehit;          // classify_sphere(): radius^2 and the error bound of the hit test
    float transparency;      // what the sphere lets through as an occluder (src/objects.h:44)
};
static_assert(sizeof(GridCell) == 8 && sizeof(GridEntry) == 32, "record size");

// The table of the pass that finishes pixels ending at the primary ray (primary_only_pixel() in
// whitted_exact.inc.h; small scenes only - the pass tests every object):
//   FrontPlane x (n_planes rounded up to even)
//   per anchor - the lights in order, then the camera: FrontAnchor, then one GridEntry per sphere
//   of the scene in container order, rounded up to an even number (at least two) with entries that
//   never hit (thr = +inf, index = INT_MAX).
struct FrontPlane {
    float n[3];              // the plane's normal as the shading uses it: float(calcNormal's result)
    float ends;              // 1 = reflectivity 0 and transparency 0: a hit spawns no child that is walked
    float h, herr;           // as in FGeom: float(origin . normal), and its error bound
    int32_t index;           // position in the caller's object list
    float transparency;
    double q[3];             // normal_ as TestCollision uses it (src/objects.h:93)
    double hd;               // origin . normal_
    float kd[3];
    float ambient;
    float pad[4];
};
struct FrontAnchor {
    double pos[3];           // the light (camera: zero)
    float ci[3];             // DLight::ci
    float pad[7];
};
constexpr int kFrontPlaneDwords = 24, kFrontAnchorDwords = 16;
static_assert(sizeof(FrontPlane) == 4 * kFrontPlaneDwords && sizeof(FrontAnchor) == 4 * kFrontAnchorDwords, "record size");

// Per-lane stack entry: origin xyz, direction xyz, and one field that holds the path weight
// (float) and the node number.
constexpr int kCounters = 10;  // fields of rt_counters
static_assert(sizeof(rt_counters) == kCounters * sizeof(uint64_t), "rt_counters");
constexpr int kStackFields = 7;
#ifndef RT_WAVES_PER_BLOCK
#define RT_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = RT_WAVES_PER_BLOCK;  // tiles (waves) per block
constexpr int kTile = 8;  // a wave renders an 8x8 pixel tile

struct KArgs {
    // ---- the first kFrontArgDwords dwords: everything primary_only_pixel() reads, fetched by
    // its waves in one go (FrontArgs in whitted_exact.inc.h mirrors this block)
    const float* front;        // small scenes: the table primary_only_pixel() reads (FrontPlane,
                               // FrontAnchor below); null = that pass is off
    uint8_t* bgr;
    double pw, ph;             // pixel pitch (src/view_frustum.h:33-34)
    int32_t n_planes, n_front; // planes of the scene; spheres of the scene (records per anchor in
                               // `front`, before rounding up)
    int32_t n_lights, width;
    int32_t height;
    int32_t rows;              // rows of the strip after clipping
    int32_t band_rows, band_first, band_step;
    uint32_t band_magic;       // ceil(2^32 / band_rows): k / band_rows = mulhi(k, band_magic), k < 2^16
    float scene_p1max;         // >= |centre|_1 of every sphere (filter bound)
    int32_t sky_unit_shift;    // see SkyTable
    float bgr_c[3];            // background colour
    int32_t row_order;         // launches with one wave per tile: block row b renders tile row
                               // (first + b) mod tiles_y, or (first - b) mod tiles_y when bit 30 is set;
                               // first = the low 16 bits (see tile_row_order() in sky_table.h)
    int32_t front_anchors;     // layout of `front` in dwords, worked out once on the host (a wave
    int32_t front_stride;      // would spend a dozen scalar instructions on it): where the
    int32_t front_camera;      // lights' blocks start, their pitch, the camera's sphere records,
    int32_t front_spheres;     // sphere records per anchor (n_front rounded up to even, at least 2)
    // ---- the rest
    int32_t max_depth;
    const DGeom* geom;
    const DMat* mat;
    const DLight* lights;
    const FGeom* fgeom;        // n_prims records rounded up to a multiple of 4
    const FGeom* tree;         // sphere tree (sphere_tree.h), n_nodes records; null = not used
    const FGeom* planes;       // the planes' filter records, original index in v[6] (bits)
    const int32_t* planes_before;  // [n_prims + 1]: number of planes with a smaller index
    double* gstack;            // global part of the stack: one region per wave of the launch
                               // (stack_workspace_bytes), null if unused
    unsigned int* tile_queue;  // launches with a fixed grid: [0] the next tile to hand out, [1] waves
                               // that have finished (both zero between launches), null otherwise
    const GridCell* grid_cells;    // anchor grids: [anchor][face][v][u], anchors = the lights, then the
    const GridEntry* grid_entries; // camera; null = not built
    const float* ltab;         // per light, ltab_stride records {c - P (xyz), threshold}: see
                               // line_miss_from_light(); planes carry threshold -inf
    int32_t* hit_id;
    uint32_t* hit_mask;
    uint32_t* spawn_mask;
    unsigned long long* counters;  // kCounters x u64, same order as rt_counters
    int32_t n_prims;
    int32_t n_nodes;
    int32_t ltab_stride;       // records per light: n_prims rounded up to a multiple of 4
    int32_t tiles_x, tiles_y;  // 8x8 tiles covering width x rows
    int32_t mask_words;
    float inv_depth[RT_MAX_DEPTH + 1];  // 1.0/depth for depth 1..max_depth-1
    int32_t primary_in_range;  // every primary ray's p1 passes the range guard of unit()
    int32_t grid_res;          // cells per face edge
    float grid_half;           // grid_res / 2
    float grid_eray_max;       // the grids hold for rays whose FilterRay::eray is at most this
    int32_t prune;             // ids, masks and counters describe the walk without the children of
                               // weight zero (the image-only kernels always leave those out)
};
constexpr int kFrontArgDwords = 28;
static_assert(offsetof(KArgs, front) == 0 && offsetof(KArgs, bgr) == 8 && offsetof(KArgs, pw) == 16 &&
                  offsetof(KArgs, n_planes) == 32 && offsetof(KArgs, band_magic) == 68 && offsetof(KArgs, bgr_c) == 80 &&
                  offsetof(KArgs, row_order) == 92 && offsetof(KArgs, front_anchors) == 96 &&
                  offsetof(KArgs, max_depth) == 4 * kFrontArgDwords,
              "FrontArgs reads the head of KArgs by position");

// Screen bounds of the scene for launches with one wave per tile (sky_table.h): tile row ty reads
// cols[min(ty, kSkyRows - 1)] = lo | hi << 8, the range of tile columns - in units of
// 2^KArgs::sky_unit_shift tiles - outside of which no primary ray of that row hits anything.
// Travels in the kernel-argument segment behind the KArgs; one independent scalar load per wave.
constexpr int kSkyRows = 1024;
struct SkyTable {
    uint16_t cols[kSkyRows];
};
static_assert(sizeof(KArgs) % 8 == 0 && alignof(KArgs) == 8 && alignof(SkyTable) <= 8,
              "the kernels find their arguments by offset in the kernel-argument segment: KArgs at 0, SkyTable behind it");

// The walk keeps at most one pending refraction child per tree level 2..max_depth-1, 3.5 KB
// per wave and entry.  Launches with a fixed grid (the tile queue): entry 0 (pushed by the
// primary hit) lives in registers, the next
// kStackLdsEntries in LDS (what bounds the number of resident waves: 7 KB per wave keeps
// five waves per SIMD), deeper ones in a global-memory workspace, written and read back by
// the same wave while it is still in the L2.  A launch that needs the workspace (max_depth
// > 5) is a fixed grid of at most kDeepMaxBlocks blocks whose waves take 8x8 tiles from a
// queue (one atomic add per tile) until it is empty, so the workspace holds one region per
// wave of that grid whatever the frame size: 52.5 MiB for the full grid at max_depth 8.
constexpr int kStackLdsEntries = 2;
#ifndef RT_STACK_REG_ENTRIES
#define RT_STACK_REG_ENTRIES 1
#endif
constexpr int kStackRegEntries = RT_STACK_REG_ENTRIES;  // deep-stack launches: entries kept in registers
__host__ __device__ inline int stack_entries(int max_depth) { return max_depth > 2 ? max_depth - 2 : 0; }
__host__ __device__ inline int stack_lds_entries(int max_depth) {
    const int e = stack_entries(max_depth) - kStackRegEntries;
    return e < 0 ? 0 : (e > kStackLdsEntries ? kStackLdsEntries : e);
}
__host__ __device__ inline int stack_global_entries(int max_depth) {
    if (stack_entries(max_depth) <= kStackLdsEntries + 1) return 0;  // fits the layout of one-tile launches
    return stack_entries(max_depth) - kStackRegEntries - kStackLdsEntries;
}
// Bytes of dynamic LDS one block needs for a given max_depth.
inline size_t stack_lds_bytes(int max_depth) {
    return (size_t)kWavesPerBlock * stack_lds_entries(max_depth) * kStackFields * 64 * sizeof(double);
}
// Launches with one wave per tile (max_depth <= 3 + kStackLdsEntries, container-order loops)
// keep no entry in registers: entries 0..kStackLdsEntries-1 in LDS, the last one - reached
// only by a refraction child at the deepest level that has children - in the lane's private
// memory.
__host__ __device__ inline int tile_stack_lds_entries(int max_depth) {
    const int e = stack_entries(max_depth);
    return e > kStackLdsEntries ? kStackLdsEntries : e;
}
inline size_t tile_stack_lds_bytes(int max_depth) {
    return (size_t)kWavesPerBlock * tile_stack_lds_entries(max_depth) * kStackFields * 64 * sizeof(double);
}
// Queue launches that use the anchor grids and produce only the image let the idle lanes of a wave
// take pending rays off the stacks of the busy ones (share_rays() in whitted_exact.inc.h).  The
// wave then needs a place where several lanes can add to one pixel's colour and a word per lane
// to pair givers with takers: kShareDwords dwords of LDS per wave, behind the stack entries of
// the block's waves.
#ifndef RT_SHARE
#define RT_SHARE 1
#endif
constexpr int kShareDwords = 256;  // 3 x 64 colour sums (float), 64 exchange words
__host__ __device__ constexpr bool share_rays_in(int mask_words, bool count, bool grid) {
    return RT_SHARE != 0 && mask_words == 0 && !count && grid;
}
inline size_t share_lds_bytes() { return (size_t)kWavesPerBlock * kShareDwords * sizeof(uint32_t); }
// Deep-stack launches: grid size and workspace.  The workspace starts with a 256-byte
// header whose first word is the tile queue; the waves' regions follow.
constexpr int kDeepMaxBlocks = 1280;      // 256 CUs x 5 resident blocks of 4 waves
constexpr int kDeepBlocksPerCu = 5;
constexpr size_t kWorkspaceHeader = 256;
constexpr size_t kDeepRegionBytes =       // per wave, at RT_MAX_DEPTH
    (size_t)(RT_MAX_DEPTH - 2 - kStackRegEntries - kStackLdsEntries) * kStackFields * 64 * sizeof(double);
constexpr size_t kWorkspaceMaxBytes = kWorkspaceHeader + (size_t)kDeepMaxBlocks * kWavesPerBlock * kDeepRegionBytes;
// Launches with a queue but without the global stack (sphere-tree scenes at max_depth <= 5)
// fit five waves per SIMD: five blocks per CU.
constexpr int kQueueBlocksPerCu = 5;
constexpr int kQueueMaxBlocks = 256 * kQueueBlocksPerCu;
// Launches that use the anchor grids are compiled for four waves per SIMD (128 registers: the
// per-lane list walk does not fit 96 without spilling): four blocks per CU.
constexpr int kGridBlocksPerCu = 4;
inline int deep_grid_blocks(int n_cus, long long tiles, bool deep = true, bool grid = false) {
    long long g = (long long)n_cus * (grid ? kGridBlocksPerCu : (deep ? kDeepBlocksPerCu : kQueueBlocksPerCu));
    const long long cap = deep ? kDeepMaxBlocks : kQueueMaxBlocks;
    if (g > cap) g = cap;
    const long long need = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > need) g = need;
    return (int)(g < 1 ? 1 : g);
}
// Bytes of workspace a deep launch of `blocks` blocks needs (0 when the depth needs none).
inline size_t stack_workspace_bytes(int max_depth, int blocks) {
    if (stack_global_entries(max_depth) == 0) return 0;
    return kWorkspaceHeader + (size_t)blocks * kWavesPerBlock * kDeepRegionBytes;
}

// Arguments of up to kMaxBatch renders done by one launch (kernel argument segment: 4 KB).
constexpr int kMaxBatch = 8;
struct KBatch {
    KArgs a[kMaxBatch];
    int32_t tile_end[kMaxBatch];   // deep-stack launches: running total of the renders' tiles
};
static_assert(sizeof(KBatch) <= 3072, "kernel argument segment");
// deep_blocks: grid of a deep-stack launch (deep_grid_blocks), ignored otherwise.
hipError_t launch_whitted_batch(const KBatch& b, int n, bool tree, bool grid, int deep_blocks, hipStream_t stream);

// The library's built-in workspace (static_workspace.hip: a file-scope device array of
// kWorkspaceMaxBytes): its address on `device`, which must be current; looked up once per device.
hipError_t static_workspace(int device, void** ptr);

// Enqueue the render of one strip on `stream`.  emit = write hit_id/masks (pointers in
// `a` that are null are skipped), count = add event counts to a.counters.
// grid = use the anchor grids (a.grid_cells must be set).
// deep_blocks: grid of a deep-stack launch (deep_grid_blocks), ignored otherwise.
// sky = screen bounds for an image-only launch with one wave per tile, null = none.
hipError_t launch_whitted(const KArgs& a, bool emit, bool count, bool tree, bool grid, int deep_blocks, hipStream_t stream,
                          const SkyTable* sky = nullptr);


// Enqueue the nearest-hit hook (rt_debug_nearest); only the scene fields of `a` are read.
// The record-based single-precision searches of primary_only_pixel() on their own (test hook):
// occluder = the feeler search from n points towards `light`, else the primary search along n rays
// from the origin; index_out -2 = not certain.
hipError_t launch_debug_records(const KArgs& a, bool occluder, int light, int self, const double* in, int n, int32_t* index_out,
                                double* value_out, hipStream_t stream);
hipError_t launch_debug_nearest(const KArgs& a, bool tree, bool grid, const double* rays, int n, int32_t* index_out, double* t_out,
                                hipStream_t stream);

// Enqueue the feeler-search hook (rt_debug_occluder); only the scene fields of `a` are read.
hipError_t launch_debug_occluder(const KArgs& a, bool tree, bool grid, int light, int self, const double* points, int n,
                                 int32_t* first_out, double* factor_out, hipStream_t stream);

// Enqueue the collision-test hook (rt_debug_intersect).
hipError_t launch_debug_intersect(const DGeom* geom, const FGeom* fgeom, float scene_p1max, int prim, const double* rays, int n,
                                  double* t_out, uint8_t* hit_out, double* unit_out,
                                  hipStream_t stream);

}  // namespace rt
