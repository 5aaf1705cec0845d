/*
 * include/rt_hip.h -- C ABI of the MI355X (gfx950) trace path.
 *
 * This is the drop-in boundary for the per-pixel trace path of Copper280z/raytrace: what a cgo-style
 * `extern fn` block in Zig binds (see INTEGRATION.md for the shim).  Plain pointers and sizes only.
 * Every entry point names the reference interface it replaces (paths under /root/reference/src).
 *
 * Conventions
 *   - return value: 0 = RT_OK, negative = error (rt_strerror); nothing throws or aborts across the ABI
 *     (the reference propagates Zig error unions with `try`, raytracer.zig:517,580).
 *   - the caller owns every host pointer for the duration of the call only; the library copies.
 *   - device memory is owned by the opaque rt_ctx (one per GPU; not thread-safe per ctx).
 *   - image layout: index jr*nx+i, top row first (raytracer.zig:521-528), 3 channels interleaved.
 *   - material kinds: 0 lambertian, 1 metal, 2 dielectric (raytracer.zig:402-406).
 */
#ifndef RT_HIP_H
#define RT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_E_ARG = -1,    /* bad argument */
    RT_E_HIP = -2,    /* a HIP runtime call failed (rt_last_hip_error) */
    RT_E_NOMEM = -3,
    RT_E_STATE = -4,  /* e.g. render before upload */
    RT_E_IO = -5,
    RT_E_CAP = -6,    /* output capacity too small */
    RT_E_NCCL = -7    /* an RCCL call failed, or RCCL could not be loaded (rt_last_rccl_error) */
};

enum { RT_LAMBERTIAN = 0, RT_METAL = 1, RT_DIELECTRIC = 2 };

typedef struct rt_ctx rt_ctx;

/* camera.zig:9-14 BasicCamera / :20-24 MovableCamera: four vectors.  subtract_origin = 1 selects
 * MovableCamera.get_ray (camera.zig:39), 0 BasicCamera.get_ray (camera.zig:16).  Replaces the implicit
 * global `pub var cam` (raytracer.zig:710). */
typedef struct {
    float    lower_left[3], horizontal[3], vertical[3], origin[3];
    uint32_t subtract_origin;
} rt_camera;

/* SoA view of `world.objects: []const Sphere` (raytracer.zig:38-41, :712) with the Material union
 * (raytracer.zig:402-449) flattened: rgb = albedo|color, param = roughness|index.  Also used as the
 * material table of a triangle scene (then only kind/r/g/b/param are read). */
typedef struct {
    uint32_t        n;
    const float    *cx, *cy, *cz, *radius;   /* radius may be negative (raytracer.zig:677) */
    const uint32_t *kind;
    const float    *r, *g, *b, *param;
} rt_spheres;

/* Triangle soup (build-defined: the reference has no ray-triangle code; the meshes are the
 * zmesh parametric spheres of main.zig:128-138).  v0/v1/v2 are [n][3] floats.
 * The test (two-sided Moller-Trumbore, flat normal = normalize(cross(v1-v0, v2-v0)) -- its side follows the winding)
 * does not depend on which vertex is listed first: at upload every triangle's vertices are rotated cyclically (winding
 * kept) so that v0 faces the longest edge.  Rejected BY DEFINITION (grazing guard, needed for a conservative cull at
 * any camera distance): a ray with sin(largest corner) * cos(angle to the normal) < 2^-7 -- within 0.45..0.52 degrees of
 * the triangle's plane for corners up to 120 degrees -- and hence, from every direction, a triangle whose largest
 * corner exceeds 179.55 degrees (a line segment to 2^-7). */
typedef struct {
    uint32_t        n;
    const float    *v0, *v1, *v2;
    const uint32_t *material;                /* index into the material table */
} rt_triangles;

/* Everything that is a compile-time constant in the reference ("Config / flags"). */
typedef struct {
    uint32_t nx, ny;                 /* shape_s, raytracer.zig:26-29 */
    uint32_t spp_begin, spp_end;     /* global sample indices rendered by this call: [begin, end) */
    uint32_t spp_total;              /* divisor of the mean (num_samples, raytracer.zig:542) */
    uint32_t max_depth;              /* reference: 16 (raytracer.zig:556) */
    float    t_min;                  /* reference: 0.001 (raytracer.zig:555) */
    uint64_t seed;                   /* counter-RNG key (the reference seeds sfc64 with i*123, :600) */
    uint32_t tile_x0, tile_y0, tile_x1, tile_y1;  /* output rect [x0,x1) x [y0,y1); all 0 = whole image */
    uint32_t row_offset_q3;          /* 1: j = ny - jr as raytracer.zig:523; 0: j = ny-1-jr */
    uint32_t spp_chunk;              /* 0: one running sum per pixel, samples added in order as trace does
                                        (raytracer.zig:535-541); C > 0: global samples [k*C,(k+1)*C) are summed per
                                        chunk and the chunk sums added in order (more, smaller work items: a shorter
                                        tail); RT_SPP_CHUNK_AUTO: the library picks C from nx*ny and the length of
                                        [spp_begin,spp_end), so the f32 sum ORDER then depends on the image size and
                                        on how a job splits its samples (1-GPU and N-GPU images agree to ~1e-6, not
                                        bit for bit; pass an explicit C where bits must not depend on the split) */
} rt_config;
#define RT_SPP_CHUNK_AUTO 0xFFFFFFFFu

typedef struct {
    uint64_t rays;                   /* world.hit evaluations (primary + scattered) */
    uint64_t primary_rays;
    double   kernel_ms;              /* hipEvent time of the trace kernel(s) */
    uint64_t rays_by_depth[20];      /* filled only when RT_FLAG_DEPTH_HISTOGRAM is set */
} rt_stats;

enum {
    RT_FLAG_DEPTH_HISTOGRAM = 1u,    /* fill rt_stats.rays_by_depth */
    RT_FLAG_FORCE_BRUTE_FORCE = 2u,  /* spheres: always the brute-force (list) kernel */
    RT_FLAG_FORCE_BVH = 4u           /* spheres: always the BVH kernel */
};

/* ---- lifetime ------------------------------------------------------------------------------ */
/* Replaces nothing in the reference (its path is pure CPU); mirrors compute.zig:19-29 Compute.create. */
int  rt_create(int device, rt_ctx **out);
/* Number of GPUs rt_create accepts (the role of std.Thread.getCpuCount(), raytracer.zig:588, for
 * rt_render_multi), or RT_E_HIP. */
int  rt_device_count(void);
void rt_destroy(rt_ctx *ctx);
const char *rt_strerror(int code);
int  rt_last_hip_error(const rt_ctx *ctx);   /* hipError_t of the last failing HIP call */
int  rt_last_rccl_error(const rt_ctx *ctx);  /* ncclResult_t of the last failing RCCL call (rt_render_multi's root ctx) */
int  rt_set_flags(rt_ctx *ctx, uint32_t flags);

/* Kernel tuning knobs (no reference counterpart; the reference has no flags at all).  Per ctx, 0 = the
 * library's default.  None of them changes a result bit except through nothing: the image is a function of the
 * scene, camera and rt_config only (tests/test_gpu_parity.py compares variants bit for bit). */
typedef struct {
    uint32_t queue_batch;         /* work items a wave reserves per global atomic: 64..256 (default 64, 128 when chunked) */
    uint32_t flush_lanes;         /* BVH kernels: run a batch of exact tests when this many lanes hold a candidate: 1..65 (24) */
    uint32_t keep_traversing;     /* BVH kernels: leave the traversal loop when fewer lanes than this traverse: 1..64 (8; triangles 16) */
    uint32_t force_global_scene;  /* 1: never stage a small scene in LDS */
    uint32_t bvh_depth_cap;       /* BVH builders: cap on inner-node levels, 8..32 (host 24, device 32; under RT_NODES_WIDE the binary
                                   * tree that is collapsed: 22, an explicit cap only lowers it); applies to the NEXT rt_upload_scene */
    uint32_t build_on_device;     /* 1: the NEXT rt_upload_scene builds the tree on the GPU (Morton order; about 1.4 ms for 3e5
                                   * triangles against 70 ms on the host, but the trace on it runs 1.15x (book spheres) to 3x
                                   * (C4 mesh) slower than on the host's SAH tree -- for scenes that change every frame) */
    uint32_t node_format;         /* tree nodes of the NEXT rt_upload_scene of a large triangle scene (host build): 0 = default (RT_NODES_WIDE),
                                   * RT_NODES_BINARY80 = two children, 80 bytes with a refs record; RT_NODES_BINARY72 = two children, 72
                                   * bytes, the child refs folded into the planes' low mantissa bytes (one load fewer per step);
                                   * RT_NODES_WIDE = up to four children (two binary levels collapsed), 144 bytes, refs folded likewise
                                   * (half as many steps per ray).  Same image whatever the format */
} rt_tuning;
#define RT_NODES_BINARY80 1u
#define RT_NODES_BINARY72 2u
#define RT_NODES_WIDE 3u
int  rt_set_tuning(rt_ctx *ctx, const rt_tuning *tuning);   /* NULL = all defaults */

/* Restrict the ctx's own stream to a subset of the GPU's compute units (hipExtStreamCreateWithCUMask; bit i of
 * mask[] = CU i; on MI355X bit i lies on XCD i % 8) and size the persistent grid for that many CUs; NULL restores
 * the whole GPU.  No reference counterpart: it produces the "@1/8 GPU" figure of BASELINE.json's metric (32 CUs,
 * 19.66 TFLOP/s of FP32 vector peak).  Renders on a caller-supplied stream are not affected.
 * The mask must keep at least one CU on EVERY one of the 8 XCDs: under ROCm 7.2 a mask that empties an XCD takes the
 * process down inside the HIP runtime (as a stream mask and as HSA_CU_MASK alike), so such a mask -- e.g.
 * {0x01010101}, "one XCD", or the first n < 8 CUs -- is rejected with RT_E_ARG before any HIP call is made, as is a
 * mask that names no CU the device has.  A progressive render in flight on the ctx ends (its stream is replaced). */
int  rt_set_cu_mask(rt_ctx *ctx, const uint32_t *mask, uint32_t n_words);
/* The argument check of rt_set_cu_mask by itself (pure host arithmetic, no GPU needed; no reference counterpart): the
 * number of CUs of a device_cus-CU GPU the mask selects, or RT_E_ARG for a mask rt_set_cu_mask refuses. */
int  rt_check_cu_mask(const uint32_t *mask, uint32_t n_words, uint32_t device_cus);

/* ---- scene upload: the `world: Hittable` argument of Render (raytracer.zig:517) ------------- */
/* Exactly one of spheres / triangles per scene; `materials` (kind, r, g, b, param) is required with
 * triangles.  Replaces the reference's CPU acceleration build (build_BVH, raytracer.zig:279-376): the
 * library builds its own GPU BVH here. */
int rt_upload_scene(rt_ctx *ctx, const rt_spheres *spheres, const rt_triangles *triangles,
                    const rt_spheres *materials);

/* What the last upload built (any pointer may be NULL).  n_bvh_nodes and bvh_depth describe the tree as uploaded: for a
 * mesh in the default four-wide node format they count wide nodes and wide levels. */
int rt_scene_info(const rt_ctx *ctx, uint32_t *n_spheres, uint32_t *n_triangles, uint32_t *n_bvh_nodes,
                  uint32_t *bvh_depth);
/* Wall time of the last rt_upload_scene: repacking (+ BVH build + validation, when the host builds) on the host,
 * device allocation + copies, and the device-side build (0 when the host built) -- the role of the timings build_BVH
 * prints, raytracer.zig:301-308,690.  Any pointer may be NULL. */
int rt_upload_timing(const rt_ctx *ctx, double *host_build_ms, double *device_copy_ms, double *device_build_ms);
/* Profiling aid, no reference counterpart: after a render with RT_FLAG_DEPTH_HISTOGRAM on a BVH scene,
 * out = { wave traversal steps, lane traversal steps, wave exact-test iterations, lane exact-test
 * iterations, wave shading passes, lanes shaded } -- lane/(64*wave) is the fill of each phase. */
int rt_debug_wave_fill(const rt_ctx *ctx, uint64_t out[6]);
/* Profiling aid, no reference counterpart: after a render with RT_FLAG_DEPTH_HISTOGRAM, pairs {wave-level
 * executions, lanes} for the blocks of the shading pass, in this order: surface fetch, normalize(direction), sky,
 * work-item dequeue, Philox, camera ray, random_in_unit_sphere, cbrt fallback, lambertian, metal, dielectric,
 * hit-normal IEEE fallback.  Returns the number of values written (24) or RT_E_CAP. */
int rt_debug_block_counts(const rt_ctx *ctx, uint32_t cap, uint64_t *out);
/* Test aid, no reference counterpart, host arithmetic only (no GPU needed): the plane rounding of the packed tree nodes
 * (rt_tuning.node_format) -- the largest float <= v (down != 0) or the smallest float >= v (down == 0) whose low
 * mantissa byte equals payload (0..255); NaN when no finite float qualifies (v infinite, NaN, or within 256 floats of
 * +-FLT_MAX on the outward side): rt_upload_scene then refuses the scene (RT_E_STATE) instead of packing a wrong ref. */
float rt_debug_round_plane(float v, uint32_t payload, int down);
/* Name of the trace kernel the last render launched (for profiles). */
const char *rt_last_kernel_name(const rt_ctx *ctx);

/* ---- the trace path --------------------------------------------------------------------------
 * rt_render replaces the fan-out + join + average of render_and_write (raytracer.zig:588-617), i.e.
 * every Render/trace/color call (raytracer.zig:517-572), and returns host images:
 *   accum_mean [ny*nx*3] f32  linear mean radiance, sum over [spp_begin,spp_end) / spp_total
 *   rgb8       [ny*nx*3] u8   trunc(255.99*sqrt(mean)) (raytracer.zig:527,545-549)
 *   raycount   [ny*nx]   u32  world.hit evaluations per pixel (parity aid)
 * any of which may be NULL.  Blocking, like the reference call. */
int rt_render(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg,
              float *accum_mean, uint8_t *rgb8, uint32_t *raycount, rt_stats *stats);

/* Device-resident variant: ADDS the radiance sum of samples [spp_begin,spp_end) into the caller's
 * device buffer d_accum_sum [ny*nx*3] f32 (and ray counts into d_raycount [ny*nx] u32 if non-NULL) on
 * `stream` (a hipStream_t; NULL = the ctx's own stream).  If stats != NULL the call synchronises the
 * stream and fills it; otherwise it only enqueues. */
int rt_render_device(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg,
                     float *d_accum_sum, uint32_t *d_raycount, void *stream, rt_stats *stats);

/* No counterpart in the reference.  For callers that cannot allocate device memory: code under HIP stream/graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same contract as rt_render_device without the
 * stats (enqueue only; pass d_raycount for ray counts), but the pass's scratch -- work-queue head, counters and
 * the per-chunk partial-sum planes -- lives in a fixed RT_NOALLOC_SCRATCH_BYTES device array inside the library
 * (one copy per GPU, held until the process ends) instead of ctx-owned hipMalloc memory, and the call makes no
 * hipMalloc/hipFree or other allocating call.  Limit: the planes of one pass, n_chunks * nx*ny*3 floats, must fit
 * in RT_NOALLOC_SCRATCH_BYTES - 1024; an explicit spp_chunk that needs more returns RT_E_CAP, RT_SPP_CHUNK_AUTO
 * picks no more chunks than fit (so its sum order can differ from rt_render_device's AUTO), spp_chunk = 0 needs
 * none.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two contexts)
 * race; the call clears what it needs at its start. */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
int rt_render_device_noalloc(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg,
                             float *d_accum_sum, uint32_t *d_raycount, void *stream);

/* Progressive form for an interactive caller -- the GUI's "Trace" button (main.zig:342-345 -> start_trace,
 * main.zig:307-318), which today blocks the UI thread until the PPM is written; shaped like the reference's unused
 * compute slot (compute.zig:123-207: dispatch, then poll until the result can be mapped).  No callbacks:
 *     rt_progressive_begin(ctx, &cam, &cfg, n_passes);
 *     do { rc = rt_progressive_step(ctx, rgb8, NULL, &spp_done, NULL); show(rgb8); } while (rc == RT_PROGRESS_MORE);
 * The samples [spp_begin,spp_end) are rendered in n_passes contiguous parts into one device SUM buffer owned by the
 * ctx; each step renders one part, tonemaps the mean over the samples so far (the last step divides by spp_total,
 * so its image IS rt_render's with spp_chunk = the pass size) and copies it to the host.  Returns RT_PROGRESS_MORE
 * while passes remain, RT_OK after the last one, a negative code on error (which ends the render). */
#define RT_PROGRESS_MORE 1
int rt_progressive_begin(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg, uint32_t n_passes);
int rt_progressive_step(rt_ctx *ctx, uint8_t *rgb8, float *accum_mean, uint32_t *spp_done, rt_stats *stats);

/* The reference's own merge (quirk Q4), for A/B against a Zig render: n_images whole-image renders of
 * cfg->spp_total samples each, image i keyed with seed i*123 (raytracer.zig:600; cfg->seed is ignored), each
 * gamma'd and quantised to u8 (raytracer.zig:527), then the per-channel INTEGER mean of the u8 images
 * (raytracer.zig:610-617).  n_images plays the role of getCpuCount().  rgb8 is a host buffer [ny*nx*3]. */
int rt_render_reference_style(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg, uint32_t n_images,
                              uint8_t *rgb8, rt_stats *stats);

/* Single-process multi-GPU form of rt_render: the thread fan-out + join + average of render_and_write
 * (raytracer.zig:588-617) with GPUs in place of threads.  ctxs[0..n_ctx) are distinct contexts (normally one per
 * GPU) that each hold the same uploaded scene.  GPU g traces the g-th contiguous part of [spp_begin,spp_end)
 * (sizes differ by at most one, low g takes the extra) into its own sum buffer.  The exchange step -- the join and
 * sum over threads of raytracer.zig:605-617 -- is ONE RCCL reduce (ncclReduce, f32 sum, root = ctxs[0]'s GPU, over
 * xGMI) on a communicator the library makes with ncclCommInitAll over the contexts' GPUs at the first call and keeps
 * on ctxs[0]; RCCL is bound at run time (dlopen), RT_E_NCCL if that fails.  Contexts that SHARE a GPU (RCCL takes
 * one rank per GPU) fall back to copying the peers' buffers to the root and adding them there in context order.
 * Results as in rt_render (any of accum_mean, rgb8, raycount may be NULL).  stats: rays summed over GPUs, kernel_ms
 * = the slowest GPU's.  Blocking.  (A job that runs one process per GPU uses rt_render_device + an RCCL reduce of
 * its own instead: raytrace_amd/distributed.py.) */
int rt_render_multi(rt_ctx *const *ctxs, uint32_t n_ctx, const rt_camera *cam, const rt_config *cfg,
                    float *accum_mean, uint8_t *rgb8, uint32_t *raycount, rt_stats *stats);
/* RCCL smoke for machines with one GPU (no reference counterpart): a one-rank communicator on the ctx's GPU and
 * one in-place ncclReduce of n floats, checked.  RT_E_NCCL when RCCL cannot be loaded or fails. */
int rt_selftest_rccl(rt_ctx *ctx, uint32_t n);

/* Device tonemap: mean = sum / spp_total; rgb8 = trunc(255.99*sqrt(mean)) (raytracer.zig:527,542-549).
 * d_mean and d_rgb8 may each be NULL; d_mean may alias d_accum_sum. */
int rt_tonemap_device(rt_ctx *ctx, const float *d_accum_sum, uint32_t n_pixels, uint32_t spp_total,
                      float *d_mean, uint8_t *d_rgb8, void *stream);

/* ---- host-side mirrors of the reference API (no GPU needed) ---------------------------------- */
/* camera.zig:9-14 */
void rt_camera_basic(rt_camera *cam);
/* MovableCamera.SetPosition, camera.zig:25-37 */
void rt_camera_set_position(rt_camera *cam, const float lookfrom[3], const float lookat[3],
                            const float v_up[3], float vfov_deg, float aspect);
/* Random_Scene, raytracer.zig:644-689 (sphere list; sfc64 stream seeded with `seed`, reference 0).
 * Writes up to cap spheres into the SoA arrays; returns the sphere count or RT_E_CAP; half_size above
 * RT_RANDOM_SCENE_MAX_HALF_SIZE (the reference uses 11) is RT_E_ARG. */
#define RT_RANDOM_SCENE_MAX_HALF_SIZE 2048u
int rt_random_scene(uint32_t half_size, uint64_t seed, uint32_t cap,
                    float *cx, float *cy, float *cz, float *radius, uint32_t *kind,
                    float *r, float *g, float *b, float *param);
/* book_segments.zig:146-149,162: sphere (0,0,-1) r .5 + ground (0,-100.5,-1) r 100, lambertian .5 */
int rt_two_sphere_scene(uint32_t cap, float *cx, float *cy, float *cz, float *radius, uint32_t *kind,
                        float *r, float *g, float *b, float *param);
/* zmesh.Shape.initParametricSphere + scale + unweld as used at main.zig:128-143.  slices/stacks <= 0
 * selects the reference's per-radius tessellation.  v0/v1/v2 are [cap][3].  Returns the triangle
 * count or RT_E_CAP; call with cap = 0 to query the count.  A request for 2^26 triangles or more (what
 * rt_upload_scene refuses) is RT_E_ARG. */
int rt_tessellate_spheres(const rt_spheres *spheres, int slices, int stacks, uint32_t cap,
                          float *v0, float *v1, float *v2, uint32_t *material);
/* write_ppm_image, image.zig:18-37 (ASCII P3) */
int rt_write_ppm(const char *fname, const uint8_t *rgb8, uint32_t nx, uint32_t ny);
/* render_and_write, raytracer.zig:580-633: render all of cfg on the ctx's GPU and write the PPM. */
int rt_render_and_write(rt_ctx *ctx, const char *fname, const rt_camera *cam, const rt_config *cfg,
                        rt_stats *stats);

/* ---- device self-test (parity aid): evaluates the device's IEEE div/sqrt, cbrt, Philox and
 * u32->float on n inputs so tests can compare them bit-for-bit with the host.  All arrays are host
 * pointers of n elements (philox_out: 4n). */
int rt_selftest_math(rt_ctx *ctx, uint32_t n, const float *a, const float *b,
                     float *out_div, float *out_sqrt, float *out_cbrt,
                     const uint32_t *ctr0, uint32_t *philox_out);
/* The cbrt of random_in_unit_sphere (raytracer.zig:577) exactly as the trace kernels evaluate it (f32 fast path +
 * guarded fallback to the musl steps), on EVERY input the path can feed it: out[k] = cbrt(k * 2^-24) for
 * 0 <= k < 2^24 (host buffer of 2^24 floats); *n_risky = inputs that took the fallback.  Parity aid. */
int rt_selftest_cbrt_grid(rt_ctx *ctx, float *out, uint64_t *n_risky);

#ifdef __cplusplus
}
#endif
#endif
