/*
 * rt_hip.h - C ABI of the MI355X (gfx950) Whitted tracer.
 *
 * The library replaces ONE call of the reference renderer: the free function
 *   void Tracer(const Color& bgr, Screen& scr, objs, lights)      (src/main.cc:115-141)
 * together with everything it reaches per pixel (TraceRay src/main.cc:19-113,
 * Object::Shade src/objects.h:20-65, Plane/Sphere::TestCollision/calcNormal
 * src/objects.h:91-106,124-142, ViewFrustum src/view_frustum.h:20-47,
 * Screen::StorePixel src/screen.h:55-73, Color::red/green/blue src/color.h:27-29).
 *
 * The reference has no FFI of its own; these entry points are what a binding for that
 * call would need.  Everything crossing the boundary is a plain pointer, a size or a POD
 * struct of doubles/int32.  All host buffers are owned by the caller, all device memory
 * by the library.  Calls block unless stated otherwise.  A context belongs to one
 * device and must not be used from two threads at once.
 *
 * Every function returning int returns RT_OK (0) or a negative RT_ERR_* code; the text
 * of the last failure is available from rt_last_error().
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------ */
#define RT_OK 0
#define RT_ERR_ARG (-1)      /* bad argument (null pointer, size out of range, ...)      */
#define RT_ERR_HIP (-2)      /* a HIP runtime call failed (no device, launch failure...) */
#define RT_ERR_STATE (-3)    /* call order wrong (render before upload, ...)             */
#define RT_ERR_NOMEM (-4)    /* host or device allocation failed                          */

/* ---- limits ------------------------------------------------------------------------ */
#define RT_MAX_DEPTH 8       /* largest max_depth: node masks are 2^(max_depth-1) bits   */
#define RT_MAX_PRIMS 65536
#define RT_MAX_LIGHTS 256

#define RT_PRIM_PLANE 0
#define RT_PRIM_SPHERE 1

/*
 * One collider, flattened from the reference's Plane / Sphere objects
 * (src/objects.h:82-111, 114-147) plus the DiffuseMaterial (src/color.h:66-74) and the
 * ambient Light (src/common.h:47-54) every Object carries (src/objects.h:77-78).
 * Order in the array = order in the reference's `colliders` vector; it decides ties in
 * the nearest-hit loop (src/main.cc:39-47) and which occluder a shadow feeler sees first
 * (src/objects.h:35-47).
 */
typedef struct rt_prim {
    int32_t type;             /* RT_PRIM_PLANE | RT_PRIM_SPHERE                           */
    int32_t reserved;         /* must be 0 (RT_ERR_ARG otherwise)                        */
    double p[3];              /* plane: origin_ ; sphere: center_                        */
    double q[3];              /* plane: normal_ AS STORED, i.e. after the constructor's
                                 makeNormal (src/objects.h:88): of unit length to 1e-12
                                 (or all zero), RT_ERR_ARG otherwise; sphere: {radius_, 0, 0},
                                 radius_ already widened from the constructor's float
                                 (src/objects.h:116,146)                                 */
    double kd[3];             /* material().kd r,g,b (ka and ks are never read)          */
    double reflectivity;      /* src/color.h:71                                          */
    double transparency;      /* src/color.h:72                                          */
    double refraction_index;  /* src/color.h:73                                          */
    double ambient_intensity; /* double(GetAmbient().intensity), a float in the reference
                                 (src/common.h:53, used at src/main.cc:112)              */
} rt_prim;

/* One PointLight (src/common.h:57-64). intensity = double(float intensity). */
typedef struct rt_light {
    double pos[3];
    double color[3];          /* light_color r,g,b                                       */
    double intensity;
} rt_light;

/* rt_params.flags */
#define RT_FLAG_COUNTERS 1u   /* also count events into rt_counters (slower kernel)      */
#define RT_FLAG_LINEAR 2u     /* test every object in container order, as the reference
                                 does (src/main.cc:37-47), whatever the scene size       */
#define RT_FLAG_TREE 4u       /* walk the sphere tree even for a small scene.  Without
                                 either flag the tree is used from 40 spheres on.  The
                                 results are identical bit for bit in both modes.        */
#define RT_FLAG_PRUNE 8u      /* hit_id, masks and counters describe the ray tree that is
                                 actually walked for the image: children whose path weight
                                 is exactly zero (a material with reflectivity 0, or
                                 transparency 0) are left out, as they always are when only
                                 the image is asked for - the reference traces them and
                                 multiplies what they return by zero (src/main.cc:107-112),
                                 so the image is the same.  Without the flag these outputs
                                 describe the reference's full ray tree.                  */

#define RT_FLAG_NOGRID 16u    /* do not use the anchor grids: per light, and for the camera,
                                 the lines through that point are sorted into cells that
                                 list the spheres they may touch, so that a shadow feeler
                                 (src/objects.h:35-46) and a primary ray (src/main.cc:37-47)
                                 test the few spheres of one cell.  Used from 8 spheres on;
                                 the results are identical bit for bit with and without.  */
#define RT_FLAG_GRID 32u      /* use the anchor grids whatever the scene size            */
#define RT_FLAG_NOSKY 64u     /* trace every pixel.  Without it an image-only render of a
                                 small scene first works out (on the host, once per
                                 partition) which 8x8 tiles no primary ray can hit anything
                                 in, and those tiles get the background colour
                                 (src/main.cc:49-53) untraced; the image is the same.     */
#define RT_DEBUG_RECORDS 256u  /* rt_debug_nearest / rt_debug_occluder only: run the search the
                                 way the single-precision pass below runs it (small scenes:
                                 every object's record, tested in single precision); an index
                                 of -2 = that search is not certain for this ray, which the
                                 render kernels then walk in double precision               */
#define RT_FLAG_NOFRONT 128u  /* walk every pixel in double precision.  Without it an
                                 image-only render of a small scene (depth <= 5) finishes the
                                 pixels whose primary ray ends its ray tree - it misses
                                 everything, or hits one plane with reflectivity 0 and
                                 transparency 0 - in single precision when every test on the
                                 way is certain at that precision: same hits and occluders,
                                 colours within the tolerance of the colour chain.        */

/*
 * Per-render parameters: the values the reference hard-wires in Tracer()/main()
 * (frame size src/main.cc:215, FOV 60 src/main.cc:126, MAX_DEPTH src/main.cc:17,
 * background src/main.cc:221) made runtime, plus the row partition used to split one
 * frame across GPUs.
 *
 * Row partition: the frame is cut into bands of `band_rows` consecutive image rows,
 * band b covering rows [b*band_rows, (b+1)*band_rows).  A render produces bands
 * band_first, band_first+band_step, ... (band_count of them), packed one after the
 * other in the output buffers ("strip order").  A band that sticks out of the frame is
 * clipped; rt_strip_rows() gives the number of rows a partition really holds.
 * The whole frame is band_rows=height, band_first=0, band_step=1, band_count=1.
 */
typedef struct rt_params {
    int32_t width, height;    /* frame size in pixels                                    */
    int32_t max_depth;        /* reference MAX_DEPTH: a ray at depth >= max_depth returns
                                 the background untraced; 2..RT_MAX_DEPTH                */
    int32_t band_rows, band_first, band_step, band_count;
    uint32_t flags;
    double fov_deg;           /* horizontal field of view in degrees (reference: 60)     */
    double bgr[3];            /* background colour r,g,b (reference: .58,.76,.92)        */
} rt_params;

/*
 * Event counts over one render (only filled when RT_FLAG_COUNTERS is set).
 * rays = TraceRay invocations that pass the depth stop (src/main.cc:31-34), i.e. the sum
 * of the spawn-mask popcounts; it is the ray count behind every Mrays/s figure.
 */
typedef struct rt_counters {
    uint64_t rays;            /* traced rays, primary + secondary                        */
    uint64_t primary_hits;    /* depth-1 rays that found an object                       */
    uint64_t hits;            /* rays that found an object, all depths                   */
    uint64_t sphere_tests;    /* Sphere::TestCollision calls, tracing + feelers          */
    uint64_t plane_tests;     /* Plane::TestCollision calls, tracing + feelers           */
    uint64_t feelers;         /* (hit, light) pairs shaded                               */
    uint64_t feelers_occluded;/* feelers that met an occluder                            */
    uint64_t refractions;     /* refraction children spawned (k > 0, src/main.cc:87)     */
    uint64_t reflections;     /* reflection children traced: TraceRay calls for them that
                                 pass the depth stop (src/main.cc:80, 31-34)             */
    uint64_t refraction_tests;/* hits at which k is evaluated (src/main.cc:86): all of
                                 them, or under RT_FLAG_PRUNE those on an object whose
                                 transparency is not zero                                */
} rt_counters;

typedef struct rt_ctx rt_ctx;

/* Number of 32-bit words of one pixel's hit/spawn mask: max(1, 2^(max_depth-1) / 32).   */
int32_t rt_mask_words(int32_t max_depth);

/* Rows held by the partition in `p` (after clipping to the frame), or RT_ERR_ARG.       */
int32_t rt_strip_rows(const rt_params* p);

/* Library version, "major.minor".                                                       */
const char* rt_version(void);

/* A context is not thread-safe: calls on one rt_ctx must not overlap in host time (use one
 * context per host thread; the work they enqueue may overlap on the device).             */

/* Create a context on HIP device `device_id`.  Fails with RT_ERR_HIP when no gfx950
 * device (or no device at all) is usable: there is no CPU fallback.                     */
int rt_create(int32_t device_id, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);

/* No entry point changes the calling thread's current HIP device: each one makes the
 * context's device current for the duration of the call and puts the caller's back.       */

/*
 * Device memory a context holds besides the scene (with its derived tables: float images,
 * sphere tree, anchor grids - about 5 MB for 1024 spheres) and the staging buffers of
 * rt_render (which grow to the largest frame rendered): renders with max_depth > 5 keep part
 * of their ray stacks in a workspace of 52.5 MiB (256 CUs), one per stream such a render was
 * enqueued on, allocated and cleared at the first such render on that stream.  Renders at
 * max_depth <= 5 allocate nothing, whatever the scene: those of scenes that use the sphere
 * tree (40 spheres or more) need a 256-byte tile queue per stream, which comes out of a pool
 * of 64 that rt_create allocates (a 65th stream allocates one).  No render enqueues a
 * memset: a launch leaves its queue at zero.  rt_trim() waits for the device to go idle and
 * frees the workspaces (a later render allocates afresh); a caller that creates streams
 * freely calls it now and then.  No counterpart in the reference.
 */
int rt_trim(rt_ctx* ctx);

/*
 * Page-locked host memory for frames (hipHostMalloc, usable with every device): rt_render
 * and rt_render_node copy into such a buffer at the full PCIe rate by DMA.  (Into ordinary
 * memory rt_render's image-only form stages the frame through a page-locked buffer of its own
 * and a few copy threads, about 20 % slower than the direct copy; the other forms leave the
 * staging to the HIP runtime, at about a third of the rate.)  Errors are reported by rt_last_error(NULL).  No counterpart in the
 * reference, whose Screen allocates its buffers with new[] (src/screen.h:24-33).
 */
int rt_host_alloc(size_t bytes, void** out);
void rt_host_free(void* p);

/* Last error text of `ctx` (or of rt_create when ctx is NULL).  Never NULL.             */
const char* rt_last_error(const rt_ctx* ctx);

/*
 * Copy the scene to the device.  Replaces the `objs` / `lights` arguments of Tracer()
 * (src/main.cc:118-119).  1 <= n_prims <= RT_MAX_PRIMS, 0 <= n_lights <= RT_MAX_LIGHTS.
 * Positions, normals, radii and light positions must be finite (RT_ERR_ARG otherwise; the
 * reference checks nothing); a radius may be zero or negative, only its square is used.
 */
int rt_upload_scene(rt_ctx* ctx, const rt_prim* prims, int32_t n_prims,
                    const rt_light* lights, int32_t n_lights);

/*
 * Render the partition described by `p` into HOST buffers, strip order, row 0 of a band
 * first, x ascending; image row 0 is the BOTTOM scanline as in the reference's
 * framebuffer (src/screen.h:62-64, bottom-up DIB src/screen.h:102-112).
 *   bgr        rows*width*3 bytes, B,G,R per pixel (src/screen.h:62-64)       required
 *              Every hit, miss and spawn decision is the reference's; the colour
 *              arithmetic (which decides nothing) runs in single precision where the
 *              reference's runs in double: a channel is within 1 of the reference's
 *              byte, and equal to it on more than 99.9 % of the pixels of a frame.
 *              The same scene, camera and partition always give the same bytes; a
 *              render that asks only for `bgr` may differ from one that also asks for
 *              ids, masks or counters by 1 in a channel of a few pixels in a million
 *              (it adds a pixel's colour terms up in another order)
 *   hit_id     rows*width int32: index of the object the primary ray hit
 *              (current_obj at src/main.cc:44, depth 1), -1 on a miss         nullable
 *   hit_mask   rows*width*rt_mask_words(max_depth) uint32. Nodes of the ray tree are
 *              heap-numbered: root 1, reflection child of i = 2i (src/main.cc:80),
 *              refraction child = 2i+1 (src/main.cc:94). Bit i (word i/32, bit i%32)
 *              is set iff node i was traced and found an object               nullable
 *   spawn_mask same shape: bit i set iff TraceRay was entered for node i with
 *              depth < max_depth                                              nullable
 *   counters   filled when RT_FLAG_COUNTERS is set                            nullable
 * Replaces Tracer() + Screen::StorePixel for the rows of the partition.
 * The copies run at the full PCIe rate when the host buffers are page-locked
 * (rt_host_alloc, hipHostMalloc, hipHostRegister), at about a third of it otherwise.  With
 * only `bgr` asked for, the partition is rendered in up to 8 parts and each part is copied
 * while the next one is rendered; when `bgr` is ordinary memory (neither allocated by nor
 * registered with the HIP runtime) the parts go by DMA into a page-locked buffer of the
 * context and from there into `bgr` with a few threads of the library (4K frame: 0.67 ms
 * against 0.56 ms into page-locked memory and 1.7 ms with the runtime's own staging).
 */
int rt_render(rt_ctx* ctx, const rt_params* p, uint8_t* bgr, int32_t* hit_id,
              uint32_t* hit_mask, uint32_t* spawn_mask, rt_counters* counters);

/*
 * Same render, but every output pointer is DEVICE memory on the context's device and
 * the work is only enqueued on `stream` (a hipStream_t, NULL = the default stream); the
 * call returns without waiting.  d_counters (device, nullable) must be zeroed by the
 * caller; counts are ADDED to it.  This is the form used under a collective: the strip
 * is rendered straight into the buffer RCCL sends.
 */
int rt_render_device(rt_ctx* ctx, const rt_params* p, void* stream, uint8_t* d_bgr,
                     int32_t* d_hit_id, uint32_t* d_hit_mask, uint32_t* d_spawn_mask,
                     rt_counters* d_counters);

/*
 * rt_render_device for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): same arguments, same
 * results, but a render that needs a workspace (max_depth > 5, or a scene with the sphere
 * tree) keeps it in an array built into the library (52.5 MiB + 256 bytes per GPU this entry
 * point is used on, held until the process ends; the runtime loads the array's code object
 * at the first call on a device, about 18 ms once) instead of one allocated per stream, so
 * the call never allocates or frees.  The
 * array holds the stacks of the largest launch grid at max_depth 8, the library's limit;
 * a larger max_depth is refused with RT_ERR_ARG like everywhere else, and the frame size
 * does not matter.  All calls on one GPU share the array: two such renders that use the
 * workspace and overlap on one GPU (different streams or contexts) race; calls on one
 * stream, or separated by a synchronisation, do not.  It has no counterpart in the
 * reference.
 */
int rt_render_device_static(rt_ctx* ctx, const rt_params* p, void* stream, uint8_t* d_bgr,
                            int32_t* d_hit_id, uint32_t* d_hit_mask, uint32_t* d_spawn_mask,
                            rt_counters* d_counters);

/*
 * n image-only renders on one device with ONE kernel launch, enqueued on `stream`: render i
 * uses the scene of ctxs[i] (contexts may repeat), params[i] (no RT_FLAG_COUNTERS; all
 * renders with or all without the sphere tree) and writes
 * rows*width*3 bytes to the device buffer d_bgr[i].  It is what one rank of the multi-GPU
 * path does per step: its strip of each frame of a batch of frames, every frame with its
 * own camera/scene.  In the reference this is n calls of Tracer() (src/main.cc:115-141) on
 * n scenes, restricted to the rows of each partition.  Errors are reported on ctxs[0].
 */
int rt_render_device_batch(rt_ctx* const* ctxs, const rt_params* params, int32_t n,
                           void* stream, uint8_t* const* d_bgr);

/*
 * One frame on the n GPUs of a node from ONE host thread: Tracer() (src/main.cc:115-141) with
 * its row loop (src/main.cc:130) dealt out in bands of `band_rows` rows, band b to
 * ctxs[b mod n].  Every context (each on its own device, or several on one) must hold the
 * scene; `p` describes the frame (its band_* fields are ignored).  Each device renders its
 * bands in a few parts and copies every part straight into the rows it covers in `bgr`
 * (height*width*3 bytes on the host, row 0 = bottom scanline) while it renders the next, so
 * the n PCIe links work side by side and nothing is re-assembled afterwards.  Blocking.
 * Image only (no RT_FLAG_COUNTERS).  Errors are reported on ctxs[0].
 */
int rt_render_node(rt_ctx* const* ctxs, int32_t n, const rt_params* p, int32_t band_rows,
                   uint8_t* bgr);

/*
 * Test hooks: run the device versions of the reference's collision tests and of the
 * vector normalisation on caller-supplied rays, so that their double results can be
 * compared bit for bit with a CPU evaluation.
 *   rays      n * 6 doubles on the host: origin xyz, direction xyz (used as given)
 *   prim      index into the uploaded scene
 *   t_out     n doubles: the collision parameter (src/objects.h:91-102,124-138)
 *   hit_out   n bytes: 1 = TestCollision returned a value, 0 = nullopt.  The test is
 *             made the way the render kernel makes it: a single-precision filter that
 *             only ever answers "certainly nullopt" (applied when the direction has unit
 *             length, as every ray of the render kernel has), then the double evaluation
 *   unit_out  n * 3 doubles: normalize(direction) as gmtl does it (length, then three
 *             divisions; unchanged when the length is 0)                      nullable
 */
int rt_debug_intersect(rt_ctx* ctx, const double* rays, int32_t n, int32_t prim,
                       double* t_out, uint8_t* hit_out, double* unit_out);

/*
 * Test hook: the nearest-hit loop of TraceRay (src/main.cc:37-47) on its own, run exactly
 * as the render kernel runs it, for n caller-supplied rays (6 doubles each: origin,
 * unit direction).  index_out: n int32, the object hit first (-1 = none); t_out: n doubles,
 * its t (0 when none).  flags: RT_FLAG_LINEAR / RT_FLAG_TREE, RT_FLAG_NOGRID / RT_FLAG_GRID,
 * 0 = by scene size.  With RT_DEBUG_RECORDS (rays from the origin only, small scenes): the
 * search of the single-precision pass: the one plane it is certain the ray hits first (a plane
 * only when nothing else can be hit), -1 = certainly nothing, -2 = not certain; t_out is then
 * only accurate to rounding.  RT_ERR_STATE when the scene has no records (too large).
 */
int rt_debug_nearest(rt_ctx* ctx, const double* rays, int32_t n, uint32_t flags,
                     int32_t* index_out, double* t_out);

/*
 * Test hook: the shadow-feeler search of Object::Shade (src/objects.h:31-47) on its own,
 * run exactly as the render kernel runs it (its float filters first, then the double
 * tests), for n caller-supplied surface points.
 *   points      n * 3 doubles on the host
 *   light       index of the light the feelers aim at
 *   self        index of the object that is skipped (the shaded one), -1 = none
 *   flags       RT_FLAG_LINEAR / RT_FLAG_TREE, RT_FLAG_NOGRID / RT_FLAG_GRID as in
 *               rt_params.flags, 0 = by scene size; RT_DEBUG_RECORDS: the search of the
 *               single-precision pass (first_out -2 = not certain)
 *   first_out   n int32: index of the first occluder in container order, -1 = none
 *   factor_out  n doubles: its transmission factor (src/objects.h:40-46), 1.0 when none;
 *               colour arithmetic, evaluated in single precision on the device (with the
 *               anchor grids, from the single-precision hit test where that is certain:
 *               within 1.5e-3 of the reference's value)
 */
int rt_debug_occluder(rt_ctx* ctx, const double* points, int32_t n, int32_t light, int32_t self,
                      uint32_t flags, int32_t* first_out, double* factor_out);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
