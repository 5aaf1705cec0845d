// C-ABI implementation (include/rt_hip.h): context, scene upload, render entry points and the
// host-side mirrors of the reference API.  Built with hipcc as host C++.
//
// There is no CPU fallback: every render entry point needs a HIP device and fails with RT_E_HIP
// otherwise.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/rt_hip.h"
#include "../host/raytracer.hpp"
#include "rt_bvh.hpp"
#include "rt_kernels.h"

struct rt_ctx {
    int          device = 0;
    int          num_cus = 0;                     // CUs the ctx's stream may use (rt_set_cu_mask)
    int          device_cus = 0;                  // CUs of the GPU
    hipStream_t  stream = nullptr;
    hipEvent_t   ev0 = nullptr, ev1 = nullptr;
    uint32_t     flags = 0;
    rt_tuning    tuning = { 0, 0, 0, 0, 0, 0, 0 };   // 0 = the library's default for that knob
    bool         bvh_wide = false;                // d_bvh holds RT_WIDE_NODE_BYTES nodes
    bool         bvh_packed = false;              // d_bvh holds RT_PACKED_NODE_BYTES nodes
    int          last_hip = 0;
    // scene (one kind of primitive at a time: spheres or triangles)
    float4      *d_geom = nullptr, *d_tri = nullptr, *d_aux = nullptr, *d_albedo = nullptr, *d_bvh = nullptr;
    uint32_t     n_sph = 0, n_tri = 0, n_nodes = 0;
    int          bvh_depth = 0;
    double       upload_build_ms = 0, upload_copy_ms = 0, upload_device_build_ms = 0;   // last rt_upload_scene: host BVH build / device allocation + copies / device build
    TraceKernel  last_kernel = RT_K_BRUTE_SPHERES;
    uint64_t     fill_counters[6] = { 0, 0, 0, 0, 0, 0 };
    uint64_t     block_counters[2 * RT_PROF_BLOCKS] = { 0 };
    // per-render scratch owned by the ctx
    uint32_t    *d_queue = nullptr;            // work-queue head
    unsigned long long *d_stats = nullptr;     // RT_STATS_WORDS
    float       *d_accum = nullptr;  size_t accum_cap = 0;     // floats
    float       *d_partial = nullptr; size_t partial_cap = 0;  // chunked mode: n_chunks planes of P*3 floats
    uint32_t    *d_raycount = nullptr; size_t raycount_cap = 0;
    float       *d_mean = nullptr;   size_t mean_cap = 0;
    uint8_t     *d_rgb8 = nullptr;   size_t rgb8_cap = 0;
    float       *d_gather = nullptr; size_t gather_cap = 0;    // rt_render_multi root: peers' sum buffers
    unsigned char *noalloc_scratch = nullptr;  // this device's copy of `scratch` (rt_render_device_noalloc)
    // rt_render_multi with this ctx as root: the RCCL communicators over the contexts' GPUs (one rank per GPU),
    // made at the first call and reused while the device list stays the same
    std::vector<void *> rccl_comms;
    std::vector<int>    rccl_devices;
    int          last_rccl = 0;                // ncclResult_t of the last failing RCCL call
    // rt_progressive_*: the render in progress
    bool         prog_active = false;
    rt_camera    prog_cam;
    rt_config    prog_cfg;
    uint32_t     prog_passes = 0, prog_next = 0;
    rt_stats     prog_total;
};

// ---- rt_render_device_noalloc's scratch -----------------------------------------------------------------
// A trace pass needs device memory beyond its inputs and outputs: the work-queue head, the counters and the
// per-chunk partial-sum planes.  The other entry points take it from the ctx (hipMalloc, grown on demand); the
// no-alloc entry point takes it from this file-scope device array.  Every GPU the library runs on holds its own
// copy, at its own address, until the process ends.  `scratch` is only a host-side handle here: the device
// address comes from hipGetSymbolAddress (once per device, in rt_create, with that device current) and is what
// kernels, memsets and copies are given.
constexpr size_t NOALLOC_QUEUE_OFF = 0, NOALLOC_STATS_OFF = 256, NOALLOC_PARTIAL_OFF = 1024;
static_assert(NOALLOC_STATS_OFF + sizeof(unsigned long long) * RT_STATS_WORDS <= NOALLOC_PARTIAL_OFF, "scratch header layout");
static_assert(RT_NOALLOC_SCRATCH_BYTES > NOALLOC_PARTIAL_OFF && RT_NOALLOC_SCRATCH_BYTES % 256 == 0, "scratch size");
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
static std::mutex g_scratch_mutex;
static unsigned char *g_scratch_ptr[MAX_DEVICES] = { nullptr };   // per device: each GPU has its own copy

namespace {

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) {                             \
            if (ctx) (ctx)->last_hip = (int)e__;             \
            return e__ == hipErrorOutOfMemory ? RT_E_NOMEM : RT_E_HIP; \
        }                                                    \
    } while (0)

// Nothing may throw across the C boundary (include/rt_hip.h): host-side allocation failures become RT_E_NOMEM.
template <class F>
int guarded(F &&body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc &) { return RT_E_NOMEM; }
    catch (...) { return RT_E_STATE; }
}

template <typename T>
int ensure(rt_ctx *ctx, T *&ptr, size_t &cap, size_t need)
{
    if (need <= cap) return RT_OK;
    if (ptr) { HIP_TRY(ctx, hipFree(ptr)); ptr = nullptr; cap = 0; }
    HIP_TRY(ctx, hipMalloc((void **)&ptr, need * sizeof(T)));
    cap = need;
    return RT_OK;
}

// RN(1/y) for the kernels' div_recip (rt_device.h), or NaN when y is not eligible: the correction step is only
// proven for a correctly rounded reciprocal of a y whose significand is not all ones, and the range keeps every
// intermediate clear of underflow and overflow.  The host divide is IEEE (no fast-math in this build).
float recip_for_div(float y)
{
    uint32_t bits;
    std::memcpy(&bits, &y, sizeof bits);
    const float ay = y < 0 ? -y : y;
    if (!(ay >= 0x1.0p-60f && ay <= 0x1.0p60f) || (bits & 0x7fffffu) == 0x7fffffu) return std::nanf("");
    return 1.0f / y;
}

// ---- RCCL, bound at run time --------------------------------------------------------------------------------
// Only rt_render_multi with contexts on DISTINCT GPUs needs it (one rank per GPU, one reduce onto the root: the
// path's single exchange step).  dlopen instead of a link dependency: the single-GPU entry points work on a host
// without RCCL, and a process that already holds an RCCL (PyTorch bundles one) keeps using that copy (same SONAME).
struct Rccl {
    typedef int (*CommInitAll_t)(void **comms, int ndev, const int *devlist);
    typedef int (*CommDestroy_t)(void *comm);
    typedef int (*Reduce_t)(const void *send, void *recv, size_t count, int dtype, int op, int root, void *comm, hipStream_t st);
    typedef int (*Group_t)(void);
    typedef const char *(*ErrorString_t)(int);
    CommInitAll_t CommInitAll = nullptr; CommDestroy_t CommDestroy = nullptr; Reduce_t Reduce = nullptr;
    Group_t GroupStart = nullptr, GroupEnd = nullptr; ErrorString_t GetErrorString = nullptr;
    bool ok = false;
    enum { Uint32 = 3, Float32 = 7, Sum = 0 };          // ncclDataType_t / ncclRedOp_t values (rccl.h)
};
const Rccl &rccl()
{
    static const Rccl r = [] {
        Rccl x;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return x;
        x.CommInitAll = (Rccl::CommInitAll_t)dlsym(h, "ncclCommInitAll");
        x.CommDestroy = (Rccl::CommDestroy_t)dlsym(h, "ncclCommDestroy");
        x.Reduce = (Rccl::Reduce_t)dlsym(h, "ncclReduce");
        x.GroupStart = (Rccl::Group_t)dlsym(h, "ncclGroupStart");
        x.GroupEnd = (Rccl::Group_t)dlsym(h, "ncclGroupEnd");
        x.GetErrorString = (Rccl::ErrorString_t)dlsym(h, "ncclGetErrorString");
        x.ok = x.CommInitAll && x.CommDestroy && x.Reduce && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r;
}
#define RCCL_TRY(ctx, expr)                                  \
    do {                                                     \
        const int r__ = (expr);                              \
        if (r__ != 0) { (ctx)->last_rccl = r__; return RT_E_NCCL; } \
    } while (0)

void drop_comms(rt_ctx *root)
{
    if (!root->rccl_comms.empty() && rccl().ok)
        for (void *c : root->rccl_comms) if (c) (void)rccl().CommDestroy(c);
    root->rccl_comms.clear(); root->rccl_devices.clear();
}
// communicators of `devices` (rank g = devices[g]), cached on the root ctx
int get_comms(rt_ctx *root, const std::vector<int> &devices)
{
    if (!rccl().ok) return RT_E_NCCL;
    if (root->rccl_devices == devices && root->rccl_comms.size() == devices.size()) return RT_OK;
    drop_comms(root);
    std::vector<void *> comms(devices.size(), nullptr);
    RCCL_TRY(root, rccl().CommInitAll(comms.data(), (int)devices.size(), devices.data()));
    root->rccl_comms = comms; root->rccl_devices = devices;
    return RT_OK;
}

int check_config(const rt_config *cfg)
{
    if (!cfg || cfg->nx == 0 || cfg->ny == 0) return RT_E_ARG;
    if ((uint64_t)cfg->nx * cfg->ny > 0x7FFFFFFFull / 3) return RT_E_ARG;   // pixel index and 3*pix fit 32 bits
    if (cfg->nx >= (1u << 24) || cfg->ny >= (1u << 24)) return RT_E_ARG;     // f32(i), f32(j) exact
    if (cfg->spp_end < cfg->spp_begin || cfg->spp_total == 0) return RT_E_ARG;
    if (cfg->spp_end > 0x7FFFFFFFu) return RT_E_ARG;        // chunk bounds k * C + C are formed in 32 bits (advance_*)
    if (cfg->max_depth > 1000000u) return RT_E_ARG;
    if (cfg->spp_chunk != 0 && cfg->spp_chunk != RT_SPP_CHUNK_AUTO) {
        const uint64_t nc = ((uint64_t)cfg->spp_end + cfg->spp_chunk - 1) / cfg->spp_chunk - cfg->spp_begin / cfg->spp_chunk;
        const uint64_t tile_slots = (((uint64_t)cfg->nx + 7) / 8) * (((uint64_t)cfg->ny + 7) / 8) * 64;   // 8x8 tiles
        // work items fit 32 bits, with room for the reservations the waves make past the end of the queue
        if (nc > 4096 || nc * tile_slots > 0xFF000000ull) return RT_E_ARG;
    }
    const bool whole = !cfg->tile_x0 && !cfg->tile_y0 && !cfg->tile_x1 && !cfg->tile_y1;
    if (!whole) {
        if (cfg->tile_x0 >= cfg->tile_x1 || cfg->tile_y0 >= cfg->tile_y1) return RT_E_ARG;
        if (cfg->tile_x1 > cfg->nx || cfg->tile_y1 > cfg->ny) return RT_E_ARG;
    }
    return RT_OK;
}

// max_planes bounds the number of partial-sum planes (0 = no bound): RT_SPP_CHUNK_AUTO then picks a chunk size that
// needs no more than that many; an explicit spp_chunk is taken as given (the caller checks p.n_chunks).
void fill_params(TraceParams &p, const rt_camera *cam, const rt_config *cfg, uint32_t n_prim, TraceKernel kernel,
                 const rt_tuning &tune, uint64_t max_planes = 0)
{
    std::memset(&p, 0, sizeof p);
    for (int k = 0; k < 3; ++k) {
        p.ll[k] = cam->lower_left[k]; p.hz[k] = cam->horizontal[k];
        p.vt[k] = cam->vertical[k];   p.org[k] = cam->origin[k];
    }
    p.sub_origin = cam->subtract_origin ? 1u : 0u;
    p.nx = cfg->nx; p.ny = cfg->ny;
    p.inv_nx = recip_for_div((float)cfg->nx); p.inv_ny = recip_for_div((float)cfg->ny);
    p.cam_fast = (p.inv_nx == p.inv_nx && p.inv_ny == p.inv_ny) ? 1u : 0u;
    p.spp_begin = cfg->spp_begin; p.spp_end = cfg->spp_end;
    p.max_depth = cfg->max_depth; p.t_min = cfg->t_min;
    p.key0 = (uint32_t)cfg->seed; p.key1 = (uint32_t)(cfg->seed >> 32);
    const bool whole = !cfg->tile_x0 && !cfg->tile_y0 && !cfg->tile_x1 && !cfg->tile_y1;
    p.x0 = whole ? 0 : cfg->tile_x0; p.y0 = whole ? 0 : cfg->tile_y0;
    p.x1 = whole ? cfg->nx : cfg->tile_x1; p.y1 = whole ? cfg->ny : cfg->tile_y1;
    p.tiles_x = (p.x1 - p.x0 + 7u) / 8u;
    const uint32_t tiles_y = (p.y1 - p.y0 + 7u) / 8u;
    // sample chunking: chunk k = global samples [k*C, (k+1)*C); one work item per (pixel, chunk)
    uint32_t chunk = cfg->spp_chunk;
    const uint32_t n_spp = cfg->spp_end - cfg->spp_begin;
    const uint64_t pixel_items = (uint64_t)p.tiles_x * tiles_y * 64u;
    const bool triangles = kernel == RT_K_BVH_TRIANGLES;
    if (chunk == RT_SPP_CHUNK_AUTO) {
        // enough items to keep the tail short (~80 per lane of a full chip; measured on C3: 10 planes run as fast
        // as 17 or 25 and write less).  How small a chunk may get depends on what an item costs against its fixed
        // overhead (dequeue, write-out, one shading pass with few lanes), measured on MI355X in round 3:
        // the brute-force kernel's cheap rays want at least 8 samples per item, the sphere tree 2 -- 1 while that
        // still leaves fewer than 16 items per lane --, the triangle tree 1.  (Until round 3 the floor was 4 for all:
        // small launches then had less than one item per lane -- C1 ran 3x, a 400 x 225 mesh 3.5x slower.)
        const uint64_t want_items = 64ull * 256 * 1280;
        uint32_t min_chunk = kernel == RT_K_BRUTE_SPHERES ? 8u : (triangles ? 1u : 2u);
        if (kernel == RT_K_BVH_SPHERES && pixel_items * ((n_spp + 1) / 2) < 16ull * (64 * 256 * 16)) min_chunk = 1u;   // 262 144 resident lanes
        chunk = 0;
        if (n_spp >= 2 * min_chunk && pixel_items < want_items) {
            uint64_t nc = (want_items + pixel_items - 1) / pixel_items;
            if (nc > n_spp / min_chunk) nc = n_spp / min_chunk;
            if (nc > 64) nc = 64;
            if (max_planes != 0 && nc > max_planes) nc = max_planes;
            if (nc > 1) chunk = (uint32_t)((n_spp + nc - 1) / nc);
        }
    }
    p.spp_chunk = 0; p.chunk0 = 0; p.n_chunks = 1;
    if (chunk != 0 && n_spp > 0) {
        const uint32_t c0 = cfg->spp_begin / chunk, c1 = (uint32_t)(((uint64_t)cfg->spp_end + chunk - 1) / chunk);
        // (a range that starts inside a chunk touches one chunk more than its length suggests: AUTO then gives up
        // chunking rather than exceed the bound)
        const bool over = cfg->spp_chunk == RT_SPP_CHUNK_AUTO && max_planes != 0 && c1 - c0 > max_planes;
        if (c1 - c0 > 1 && !over) { p.spp_chunk = chunk; p.chunk0 = c0; p.n_chunks = c1 - c0; }
    }
    p.plane_stride = (size_t)cfg->nx * cfg->ny * 3;
    p.queue_batch = p.n_chunks > 1 ? 128u : 64u;    // coarse items (whole pixels): reserve less, the tail is long enough
    if (tune.queue_batch) p.queue_batch = tune.queue_batch;
    p.n_items = (uint32_t)(pixel_items * p.n_chunks);
    p.q3 = cfg->row_offset_q3 ? 1u : 0u;
    p.n_prim = n_prim;
    // leave the traversal loop to test/shade when fewer lanes than this still traverse (measured on MI355X: sphere
    // scenes from LDS 8; the triangle-mesh scene, whose steps wait on global memory, 16)
    p.keep_traversing = triangles ? 16 : 8;
    p.flush_lanes = 24;                              // 65 = only the "queue nearly full" rule
    if (tune.flush_lanes) p.flush_lanes = tune.flush_lanes;
    p.force_global_scene = tune.force_global_scene ? 1u : 0u;
    if (tune.keep_traversing) p.keep_traversing = tune.keep_traversing;
}

// Spheres up to this count are traced by the brute-force kernel (one scalar-fed loop, no divergence);
// above it the BVH kernel wins (measured on MI355X).
constexpr uint32_t BRUTE_FORCE_MAX_SPHERES = 16;

TraceKernel pick_kernel(const rt_ctx *ctx)
{
    if (ctx->n_tri) return RT_K_BVH_TRIANGLES;
    if (ctx->flags & RT_FLAG_FORCE_BRUTE_FORCE) return RT_K_BRUTE_SPHERES;
    if (ctx->flags & RT_FLAG_FORCE_BVH) return RT_K_BVH_SPHERES;
    return ctx->n_sph <= BRUTE_FORCE_MAX_SPHERES ? RT_K_BRUTE_SPHERES : RT_K_BVH_SPHERES;
}

// The device memory a trace pass needs beyond its inputs and outputs: the work-queue head, the counters and the
// planes of per-chunk partial sums.  The ctx's own (grown on demand with hipMalloc), or a fixed one the caller
// brings (rt_render_device_noalloc: partial_cap floats, never grown).
struct TraceScratch {
    uint32_t           *queue;
    unsigned long long *stats;
    float              *partial;       // null: the ctx's d_partial, grown on demand
    size_t              partial_cap;   // floats (fixed scratch only)
    bool                fixed;
};

// enqueue one trace pass on `stream`; timing events are recorded around the kernel (not for a fixed scratch:
// that path is for callers under stream capture)
int enqueue_trace(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg, float *d_accum_sum,
                  uint32_t *d_raycount, hipStream_t stream, const TraceScratch *fixed = nullptr)
{
    if (ctx->n_sph == 0 && ctx->n_tri == 0) return RT_E_STATE;
    TraceParams p;
    const size_t plane = (size_t)cfg->nx * cfg->ny * 3;
    uint64_t max_planes = 0;     // fixed scratch: as many planes as fit, at least "one" (= no chunking)
    if (fixed) { max_planes = fixed->partial_cap / plane; if (max_planes < 1) max_planes = 1; }
    fill_params(p, cam, cfg, ctx->n_tri ? ctx->n_tri : ctx->n_sph, pick_kernel(ctx), ctx->tuning, max_planes);
    if (fixed && p.n_chunks > 1 && (uint64_t)plane * p.n_chunks > fixed->partial_cap) return RT_E_CAP;
    uint32_t *const d_queue = fixed ? fixed->queue : ctx->d_queue;
    unsigned long long *const d_stats = fixed ? fixed->stats : ctx->d_stats;
    HIP_TRY(ctx, rt_launch_clear_pass(d_queue, d_stats, stream));
    const bool hist = (ctx->flags & RT_FLAG_DEPTH_HISTOGRAM) != 0;
    DeviceScene s;
    s.geom = ctx->d_geom; s.tri = ctx->d_tri; s.aux = ctx->d_aux; s.albedo = ctx->d_albedo; s.bvh = ctx->d_bvh;
    s.n_prim = p.n_prim; s.n_nodes = ctx->n_nodes; s.bvh_depth = ctx->bvh_depth; s.packed = ctx->bvh_packed; s.wide = ctx->bvh_wide;
    ctx->last_kernel = pick_kernel(ctx);
    float *d_partial = fixed ? fixed->partial : ctx->d_partial;
    if (p.n_chunks > 1) {        // scratch planes for the per-chunk partial sums (every cell is written)
        if (!fixed) {
            const int rc = ensure(ctx, ctx->d_partial, ctx->partial_cap, p.plane_stride * p.n_chunks);
            if (rc != RT_OK) return rc;
            d_partial = ctx->d_partial;
        }
        if (p.x0 != 0 || p.y0 != 0 || p.x1 != cfg->nx || p.y1 != cfg->ny)   // tiles leave cells untouched
            HIP_TRY(ctx, hipMemsetAsync(d_partial, 0, p.plane_stride * p.n_chunks * sizeof(float), stream));
    }
    if (!fixed) HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    if (cfg->spp_end > cfg->spp_begin) {
        HIP_TRY(ctx, rt_launch_trace(ctx->last_kernel, p, s, d_accum_sum, d_partial, d_raycount, d_queue,
                                     d_stats, hist, ctx->num_cus, stream));
        if (p.n_chunks > 1)
            HIP_TRY(ctx, rt_launch_reduce_chunks(d_accum_sum, d_partial, cfg->nx * cfg->ny, p.n_chunks,
                                                 p.plane_stride, stream));
    }
    if (!fixed) HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
    return RT_OK;
}

int read_stats(rt_ctx *ctx, hipStream_t stream, rt_stats *stats)
{
    unsigned long long h[RT_STATS_WORDS];
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    std::memset(stats, 0, sizeof *stats);
    stats->rays = h[0]; stats->primary_rays = h[1]; stats->kernel_ms = ms;
    for (int k = 0; k < 20; ++k) stats->rays_by_depth[k] = h[2 + k];
    for (int k = 0; k < 6; ++k) ctx->fill_counters[k] = h[22 + k];
    for (int k = 0; k < 2 * RT_PROF_BLOCKS; ++k) ctx->block_counters[k] = h[28 + k];
    return RT_OK;
}

void soa_out(const std::vector<rt::Sphere> &objs, float *cx, float *cy, float *cz, float *radius, uint32_t *kind,
             float *r, float *g, float *b, float *param)
{
    for (size_t k = 0; k < objs.size(); ++k) {
        const rt::Sphere &s = objs[k];
        cx[k] = s.center.x; cy[k] = s.center.y; cz[k] = s.center.z; radius[k] = s.radius;
        kind[k] = (uint32_t)s.material.tag;
        r[k] = s.material.rgb.x; g[k] = s.material.rgb.y; b[k] = s.material.rgb.z; param[k] = s.material.param;
    }
}

}  // namespace

extern "C" {

const char *rt_strerror(int code)
{
    switch (code) {
    case RT_OK: return "ok";
    case RT_E_ARG: return "invalid argument";
    case RT_E_HIP: return "HIP runtime error (see rt_last_hip_error)";
    case RT_E_NOMEM: return "out of memory";
    case RT_E_STATE: return "invalid state (no scene uploaded?)";
    case RT_E_IO: return "I/O error";
    case RT_E_NCCL: return "RCCL call failed or RCCL could not be loaded (see rt_last_rccl_error)";
    case RT_E_CAP: return "capacity too small (output buffer, or the fixed scratch of rt_render_device_noalloc)";
    default: return "unknown error";
    }
}

int rt_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return RT_E_HIP;
    return ndev;
}

int rt_create(int device, rt_ctx **out)
{
    if (!out) return RT_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RT_E_HIP;   // no GPU: fail loudly
    if (device < 0 || device >= ndev) return RT_E_ARG;
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return RT_E_NOMEM;
    ctx->device = device;
    int rc = RT_OK;
    do {
        hipError_t e;
        if ((e = hipSetDevice(device)) != hipSuccess) { ctx->last_hip = e; rc = RT_E_HIP; break; }
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { ctx->last_hip = e; rc = RT_E_HIP; break; }
        ctx->num_cus = ctx->device_cus = prop.multiProcessorCount;
        if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) { ctx->last_hip = e; rc = RT_E_HIP; break; }
        if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) { ctx->last_hip = e; rc = RT_E_HIP; break; }
        if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) { ctx->last_hip = e; rc = RT_E_HIP; break; }
        if ((e = hipMalloc((void **)&ctx->d_queue, 256)) != hipSuccess) { ctx->last_hip = e; rc = RT_E_NOMEM; break; }
        if ((e = hipMalloc((void **)&ctx->d_stats, sizeof(unsigned long long) * RT_STATS_WORDS)) != hipSuccess) {
            ctx->last_hip = e; rc = RT_E_NOMEM; break;
        }
        {   // this device's address of `scratch` (device `device` is current)
            std::lock_guard<std::mutex> lock(g_scratch_mutex);
            unsigned char *ptr = device < MAX_DEVICES ? g_scratch_ptr[device] : nullptr;
            if (!ptr) {
                if ((e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch))) != hipSuccess || !ptr) {
                    ctx->last_hip = e; rc = RT_E_HIP; break;
                }
                if (device < MAX_DEVICES) g_scratch_ptr[device] = ptr;
            }
            ctx->noalloc_scratch = ptr;
        }
    } while (0);
    if (rc != RT_OK) { rt_destroy(ctx); return rc; }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_geom); (void)hipFree(ctx->d_tri); (void)hipFree(ctx->d_aux); (void)hipFree(ctx->d_albedo);
    (void)hipFree(ctx->d_bvh);
    (void)hipFree(ctx->d_queue); (void)hipFree(ctx->d_stats);
    (void)hipFree(ctx->d_accum); (void)hipFree(ctx->d_partial); (void)hipFree(ctx->d_raycount); (void)hipFree(ctx->d_mean); (void)hipFree(ctx->d_rgb8);
    (void)hipFree(ctx->d_gather);
    drop_comms(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rt_last_hip_error(const rt_ctx *ctx) { return ctx ? ctx->last_hip : 0; }
int rt_last_rccl_error(const rt_ctx *ctx) { return ctx ? ctx->last_rccl : 0; }

int rt_set_flags(rt_ctx *ctx, uint32_t flags)
{
    if (!ctx) return RT_E_ARG;
    ctx->flags = flags;
    return RT_OK;
}

// gfx950 (MI355X): 8 XCDs; bit i of a HIP CU mask lies on XCD i % 8 (tools/ubench/cu_mask_probe.hip).
constexpr uint32_t RT_XCDS = 8;

// Pure argument check of rt_set_cu_mask (no HIP call): the number of usable CUs in the mask, or RT_E_ARG when the
// mask is empty, names only CUs the device does not have, or leaves one of the 8 XCDs without a CU -- the ROCm 7.2
// runtime dies inside hipExtStreamCreateWithCUMask / the first launch on such a stream, so that mask
// must never reach HIP.
int rt_check_cu_mask(const uint32_t *mask, uint32_t n_words, uint32_t device_cus)
{
    if (!mask || n_words == 0 || n_words > 64 || device_cus == 0) return RT_E_ARG;
    uint32_t per_xcd[RT_XCDS] = { 0 };
    int cus = 0;
    for (uint32_t w = 0; w < n_words; ++w)
        for (uint32_t b = 0; b < 32; ++b) {
            const uint32_t cu = w * 32 + b;
            if (cu < device_cus && ((mask[w] >> b) & 1u)) { per_xcd[cu % RT_XCDS]++; cus++; }
        }
    if (cus == 0) return RT_E_ARG;
    if (device_cus >= RT_XCDS)
        for (uint32_t x = 0; x < RT_XCDS; ++x)
            if (per_xcd[x] == 0) return RT_E_ARG;
    return cus;
}

int rt_set_cu_mask(rt_ctx *ctx, const uint32_t *mask, uint32_t n_words)
{
    if (!ctx || (mask && (n_words == 0 || n_words > 64))) return RT_E_ARG;
    int cus = ctx->device_cus;
    if (mask) {                       // checked before any HIP call
        cus = rt_check_cu_mask(mask, n_words, (uint32_t)ctx->device_cus);
        if (cus < 0) return cus;
    }
    ctx->prog_active = false;         // a progressive render in flight ends here (its stream goes away)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    if (mask) HIP_TRY(ctx, hipExtStreamCreateWithCUMask(&st, n_words, mask));
    else HIP_TRY(ctx, hipStreamCreate(&st));
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    ctx->stream = st;
    ctx->num_cus = cus;          // the persistent grid is sized for the CUs the stream may use
    return RT_OK;
}

int rt_set_tuning(rt_ctx *ctx, const rt_tuning *t)
{
    if (!ctx) return RT_E_ARG;
    const rt_tuning def = { 0, 0, 0, 0, 0, 0, 0 };
    if (!t) t = &def;
    // queue_batch: every wave reserves at most two batches past the end of the queue, and check_config leaves
    // 2^24 item numbers of headroom: 16384 resident waves x 2 x 256 stays inside it
    if (t->queue_batch && (t->queue_batch < 64 || t->queue_batch > 256)) return RT_E_ARG;
    if (t->flush_lanes > 65 || t->keep_traversing > 64) return RT_E_ARG;
    if (t->bvh_depth_cap && (t->bvh_depth_cap < 8 || t->bvh_depth_cap > 32)) return RT_E_ARG;
    if (t->build_on_device > 1 || t->node_format > RT_NODES_WIDE) return RT_E_ARG;
    ctx->tuning = *t;
    return RT_OK;
}

// Largest float <= v (down) or smallest float >= v (up) whose low mantissa byte is `payload`: the packed node format
// (rt_kernels.h) carries the child refs in those bytes, and a plane may only move outward.
static float with_low_byte(float v, uint32_t payload, bool down)
{
    if (!down) return -with_low_byte(-v, payload, true);
    uint32_t b; std::memcpy(&b, &v, 4);
    for (int step = 0; step < 4; ++step) {
        const uint32_t cb = (b & ~0xFFu) | payload;
        float f; std::memcpy(&f, &cb, 4);
        if (f <= v) return f;
        // one block of 256 floats towards -infinity
        if (b & 0x80000000u) b += 0x100u;
        else if ((b & ~0xFFu) == 0u) b = 0x80000000u;
        else b -= 0x100u;
    }
    return -INFINITY;
}

// with_low_byte for a node plane, checked: false when the result is not a finite float carrying the payload (planes at
// +-inf, NaN, or in the last block of 256 floats below FLT_MAX have no outward neighbour with that byte) -- the child
// ref would then be garbage and the kernel would index the node array with it.
static bool pack_plane(float v, uint32_t payload, bool down, float &out)
{
    out = with_low_byte(v, payload, down);
    uint32_t b; std::memcpy(&b, &out, 4);
    return std::isfinite(out) && (b & 0xFFu) == payload && (down ? out <= v : out >= v);
}

static int upload_vec(rt_ctx *ctx, float4 *&dst, const std::vector<float4> &src)
{
    if (src.empty()) return RT_OK;
    HIP_TRY(ctx, hipMalloc((void **)&dst, sizeof(float4) * src.size()));
    HIP_TRY(ctx, hipMemcpy(dst, src.data(), sizeof(float4) * src.size(), hipMemcpyHostToDevice));
    return RT_OK;
}

// Canonical vertex order of a triangle (part of the build-defined triangle test; oracle: orc_canonical_triangle): the
// vertices are rotated cyclically -- winding and normal side kept -- so that v0 faces the longest edge, i.e. sits at the
// largest corner.  triangle_exact's grazing guard bounds sin(corner at v0) * cos(theta): at the largest corner it depends
// on the triangle's shape only, not on which vertex the caller listed first.  Squared lengths in strict f32, ordered
// dot (this file is built with -ffp-contract=off); among equally long edges the one facing the lexicographically
// smallest vertex wins, so the result does not depend on the order the caller listed the vertices in.
static void canonical_vertex_order(const float *&v0, const float *&v1, const float *&v2)
{
    auto len2 = [](const float *p, const float *q) {
        const float x = p[0] - q[0], y = p[1] - q[1], z = p[2] - q[2];
        return (x * x + y * y) + z * z;
    };
    auto lex_less = [](const float *p, const float *q) {
        if (p[0] != q[0]) return p[0] < q[0];
        if (p[1] != q[1]) return p[1] < q[1];
        return p[2] < q[2];
    };
    const float *v[3] = { v0, v1, v2 };
    const float opp[3] = { len2(v2, v1), len2(v0, v2), len2(v1, v0) };      // squared edge facing v0, v1, v2
    int best = 0;
    for (int i = 1; i < 3; ++i)
        if (opp[i] > opp[best] || (opp[i] == opp[best] && lex_less(v[i], v[best]))) best = i;
    v0 = v[best]; v1 = v[(best + 1) % 3]; v2 = v[(best + 2) % 3];
}

static bool spheres_valid(const rt_spheres *s, bool need_geometry)
{
    if (!s || s->n == 0 || !s->kind || !s->r || !s->g || !s->b || !s->param) return false;
    if (need_geometry && (!s->cx || !s->cy || !s->cz || !s->radius)) return false;
    for (uint32_t k = 0; k < s->n; ++k)
        if (s->kind[k] > RT_DIELECTRIC) return false;
    return true;
}

int rt_upload_scene(rt_ctx *ctx, const rt_spheres *sph, const rt_triangles *tri, const rt_spheres *materials)
{
    return guarded([&]() -> int {
        if (!ctx) return RT_E_ARG;
        const bool have_sph = sph && sph->n, have_tri = tri && tri->n;
        if (have_sph == have_tri) return RT_E_ARG;            // exactly one kind of primitive per scene
        const auto t_begin = std::chrono::steady_clock::now();
        std::vector<float4> geom, tris, aux, alb;
        std::vector<rtbvh::Prim> prims;
        const rt_spheres *mats = have_sph ? sph : materials;
        const uint32_t n_prim = have_sph ? sph->n : tri->n;
        const bool device_build = ctx->tuning.build_on_device != 0 && n_prim >= 2;   // (the one-primitive tree is a constant)
        if (!spheres_valid(mats, have_sph)) return RT_E_ARG;
        if (mats->n >= 0x7FFFFFFFu) return RT_E_ARG;
        for (uint32_t k = 0; k < mats->n; ++k) {
            float kind_bits; std::memcpy(&kind_bits, &mats->kind[k], 4);
            aux.push_back(make_float4(have_sph ? sph->radius[k] : 0.0f, kind_bits, mats->param[k],
                                      have_sph ? recip_for_div(sph->radius[k]) : 0.0f));
            // two records per material: {r, g, b, schlick r0^2}, {1/index, 0, 0, 0}.  The two dielectric constants
            // (raytracer.zig:464-466, :482) are the same IEEE operations the scatter code would do per hit.
            const float idx = mats->param[k];
            float r0 = (1.0f - idx) / (1.0f + idx);
            r0 = r0 * r0;
            alb.push_back(make_float4(mats->r[k], mats->g[k], mats->b[k], r0));
            alb.push_back(make_float4(1.0f / idx, 0.0f, 0.0f, 0.0f));
        }
        if (have_sph) {
            for (uint32_t k = 0; k < sph->n; ++k) {
                const float rad = sph->radius[k];
                geom.push_back(make_float4(sph->cx[k], sph->cy[k], sph->cz[k], rad * rad));   // raytracer.zig:51
                if (!device_build) prims.push_back(rtbvh::sphere_prim(sph->cx[k], sph->cy[k], sph->cz[k], rad));
            }
        } else {
            if (!tri->v0 || !tri->v1 || !tri->v2 || !tri->material || tri->n >= (1u << 26)) return RT_E_ARG;   // 26-bit triangle ids in the kernels' candidate entries
            for (uint32_t k = 0; k < tri->n; ++k) {
                if (tri->material[k] >= mats->n) return RT_E_ARG;
                const float *a = tri->v0 + 3 * (size_t)k, *b = tri->v1 + 3 * (size_t)k, *c = tri->v2 + 3 * (size_t)k;
                canonical_vertex_order(a, b, c);
                float mat_bits; std::memcpy(&mat_bits, &tri->material[k], 4);
                tris.push_back(make_float4(a[0], a[1], a[2], mat_bits));
                tris.push_back(make_float4(b[0], b[1], b[2], 0.0f));
                tris.push_back(make_float4(c[0], c[1], c[2], 0.0f));
                if (!device_build) prims.push_back(rtbvh::triangle_prim(a, b, c));
            }
        }
        // wide nodes: two binary levels per node and up to 3 stack entries per wide level, 32-entry stacks -> binary cap 22
        const uint32_t fmt = ctx->tuning.node_format ? ctx->tuning.node_format : RT_NODES_WIDE;
        const bool want_wide = have_tri && !device_build && fmt == RT_NODES_WIDE && n_prim >= 1024u && n_prim <= (1u << 22) &&
                               (!ctx->tuning.bvh_depth_cap || ctx->tuning.node_format == RT_NODES_WIDE);   // (an explicit cap asks for a binary tree of that depth)
        const int depth_cap = want_wide ? std::min(22, ctx->tuning.bvh_depth_cap ? (int)ctx->tuning.bvh_depth_cap : 22) : ctx->tuning.bvh_depth_cap ? (int)ctx->tuning.bvh_depth_cap : 24;   // 24-entry stacks: 5 workgroups per CU instead of 3
        const rtbvh::Tree tree = device_build ? rtbvh::Tree() : rtbvh::build(prims, depth_cap);
        if (!device_build && !rtbvh::validate(tree, prims)) return RT_E_STATE;   // NaN/inf geometry ends here, not on the GPU
        // host tree (64-byte nodes: lo0, hi0, lo1, hi1, refs) -> device nodes (RT_NODE_BYTES, rt_kernels.h)
        const size_t n_nodes = device_build ? (size_t)n_prim - 1 : tree.nodes.size();
        if (n_nodes * (uint64_t)RT_NODE_BYTES >= 0x80000000ull) return RT_E_ARG;   // byte offsets keep bit 31 free
        // large triangle scenes (not the ones the kernels stage in LDS): packed 72-byte nodes, refs inside the planes
        // 4-wide nodes: node k of the binary tree and its (inner) children become one node over the grandchildren
        std::vector<float> wnodes;
        int wide_depth = 0;
        if (want_wide) {
            struct Item { uint32_t bin; uint32_t wide; int level; };
            std::vector<Item> todo{ { 0u, 0u, 1 } };
            uint32_t n_wide = 1;
            const float inf = INFINITY;
            auto slot = [&](uint32_t wi) { if (wnodes.size() < (size_t)(wi + 1) * (RT_WIDE_NODE_BYTES / 4)) wnodes.resize((size_t)(wi + 1) * (RT_WIDE_NODE_BYTES / 4)); return wnodes.data() + (size_t)wi * (RT_WIDE_NODE_BYTES / 4); };
            while (!todo.empty()) {
                const Item it = todo.back(); todo.pop_back();
                wide_depth = std::max(wide_depth, it.level);
                // gather up to four (box, ref) pairs: a leaf child stays, an inner child is replaced by its two children
                float lo[4][3], hi[4][3]; uint32_t ref[4]; bool inner[4]; int nc = 0;
                const float *w = tree.nodes[it.bin].w;
                for (int c = 0; c < 2; ++c) {
                    uint32_t r; std::memcpy(&r, &w[12 + c], 4);
                    if (r & rtbvh::LEAF_BIT) {
                        for (int k = 0; k < 3; ++k) { lo[nc][k] = w[6 * c + k]; hi[nc][k] = w[6 * c + 3 + k]; }
                        ref[nc] = r; inner[nc] = false; nc++;
                    } else {
                        const float *g = tree.nodes[r].w;
                        for (int cc = 0; cc < 2; ++cc) {
                            uint32_t rr; std::memcpy(&rr, &g[12 + cc], 4);
                            for (int k = 0; k < 3; ++k) { lo[nc][k] = g[6 * cc + k]; hi[nc][k] = g[6 * cc + 3 + k]; }
                            ref[nc] = rr; inner[nc] = !(rr & rtbvh::LEAF_BIT); nc++;
                        }
                    }
                }
                uint32_t packed_ref[4] = { 0, 0, 0, 0 };
                for (int c = 0; c < nc; ++c) {
                    if (inner[c]) { packed_ref[c] = n_wide; todo.push_back({ ref[c], n_wide, it.level + 1 }); n_wide++; }
                    else packed_ref[c] = RT_PACKED_LEAF_BIT | (ref[c] & ~rtbvh::LEAF_BIT);
                }
                float *d = slot(it.wide);
                for (int ax = 0; ax < 3; ++ax) {
                    float *a = d + 12 * ax;                                  // {hi0..hi3, lo0..lo3, hi0..hi3}
                    for (int c = 0; c < 4; ++c) {
                        float l = inf, h = -inf;
                        if (c < nc) {
                            const uint32_t b = (packed_ref[c] >> (8 * ax)) & 0xFFu;
                            if (!pack_plane(lo[c][ax], b, true, l) || !pack_plane(hi[c][ax], b, false, h)) return RT_E_STATE;
                        }
                        a[c] = h; a[4 + c] = l; a[8 + c] = h;
                    }
                }
            }
            if (n_wide >= (1u << 23) || wide_depth > 11) return RT_E_STATE;
            wnodes.resize((size_t)n_wide * (RT_WIDE_NODE_BYTES / 4));
        }
        const bool wide = want_wide;
        const bool packed = !wide && have_tri && !device_build && fmt != RT_NODES_BINARY80 && n_nodes < (1u << 23) && n_prim < (1u << 23) &&
                            (n_nodes * (size_t)RT_NODE_BYTES > 40 * 1024 || tree.depth > 16);
        std::vector<float> pnodes(packed ? n_nodes * (size_t)(RT_PACKED_NODE_BYTES / 4) : 0);
        for (size_t k = 0; packed && k < tree.nodes.size(); ++k) {
            const float *w = tree.nodes[k].w;
            float *d = pnodes.data() + k * (RT_PACKED_NODE_BYTES / 4);
            uint32_t ref[2];
            for (int c = 0; c < 2; ++c) {
                uint32_t r; std::memcpy(&r, &w[12 + c], 4);
                ref[c] = (r & rtbvh::LEAF_BIT) ? (RT_PACKED_LEAF_BIT | (r & ~rtbvh::LEAF_BIT)) : r;
            }
            for (int ax = 0; ax < 3; ++ax) {
                const uint32_t b0 = (ref[0] >> (8 * ax)) & 0xFFu, b1 = (ref[1] >> (8 * ax)) & 0xFFu;
                float lo0, hi0, lo1, hi1;
                if (!pack_plane(w[ax], b0, true, lo0) || !pack_plane(w[3 + ax], b0, false, hi0) ||
                    !pack_plane(w[6 + ax], b1, true, lo1) || !pack_plane(w[9 + ax], b1, false, hi1)) return RT_E_STATE;
                float *a = d + 6 * ax;
                a[0] = hi0; a[1] = hi1; a[2] = lo0; a[3] = lo1; a[4] = hi0; a[5] = hi1;
            }
        }
        std::vector<float4> nodes(packed || wide ? 0 : tree.nodes.size() * (RT_NODE_BYTES / 16));
        for (size_t k = 0; !packed && !wide && k < tree.nodes.size(); ++k) {
            const float *w = tree.nodes[k].w;
            float *d = reinterpret_cast<float *>(nodes.data() + k * (RT_NODE_BYTES / 16));
            for (int ax = 0; ax < 3; ++ax) {
                const float lo0 = w[ax], hi0 = w[3 + ax], lo1 = w[6 + ax], hi1 = w[9 + ax];
                float *a = d + 6 * ax;
                a[0] = hi0; a[1] = hi1; a[2] = lo0; a[3] = lo1; a[4] = hi0; a[5] = hi1;
            }
            for (int c = 0; c < 2; ++c) {
                uint32_t ref; std::memcpy(&ref, &w[12 + c], 4);
                if (!(ref & rtbvh::LEAF_BIT)) ref *= (uint32_t)RT_NODE_BYTES;
                std::memcpy(&d[18 + c], &ref, 4);
            }
        }

        const auto t_built = std::chrono::steady_clock::now();
        ctx->prog_active = false;         // a progressive render of the old scene ends here
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_geom); (void)hipFree(ctx->d_tri); (void)hipFree(ctx->d_aux); (void)hipFree(ctx->d_albedo);
        (void)hipFree(ctx->d_bvh);
        ctx->d_geom = ctx->d_tri = ctx->d_aux = ctx->d_albedo = ctx->d_bvh = nullptr;
        ctx->n_sph = ctx->n_tri = ctx->n_nodes = 0; ctx->bvh_depth = 0;
        int rc;
        if ((rc = upload_vec(ctx, ctx->d_geom, geom)) != RT_OK) return rc;
        if ((rc = upload_vec(ctx, ctx->d_tri, tris)) != RT_OK) return rc;
        if ((rc = upload_vec(ctx, ctx->d_aux, aux)) != RT_OK) return rc;
        if ((rc = upload_vec(ctx, ctx->d_albedo, alb)) != RT_OK) return rc;
        if ((rc = upload_vec(ctx, ctx->d_bvh, nodes)) != RT_OK) return rc;
        if (packed) {
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_bvh, pnodes.size() * sizeof(float)));
            HIP_TRY(ctx, hipMemcpy(ctx->d_bvh, pnodes.data(), pnodes.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (wide) {
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_bvh, wnodes.size() * sizeof(float)));
            HIP_TRY(ctx, hipMemcpy(ctx->d_bvh, wnodes.data(), wnodes.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        ctx->bvh_packed = packed;
        ctx->bvh_wide = wide;
        const auto t_copied = std::chrono::steady_clock::now();
        int depth = tree.depth;
        if (device_build) {
            uint32_t bad = 0;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_bvh, n_nodes * (size_t)RT_NODE_BYTES));
            // (the Morton tree is deeper than the SAH tree; under a cap of 24 the forced middle splits cost the C4 trace 25 %)
            const uint32_t dev_cap = ctx->tuning.bvh_depth_cap ? ctx->tuning.bvh_depth_cap : (uint32_t)rtbvh::MAX_DEPTH;
            HIP_TRY(ctx, rt_device_build_bvh(have_tri, n_prim, ctx->d_geom, ctx->d_aux, ctx->d_tri, ctx->d_bvh, dev_cap,
                                             &depth, &bad, ctx->stream));
            if (bad || depth > rtbvh::MAX_DEPTH) {            // NaN/inf geometry: no scene rather than a tree that lies
                (void)hipFree(ctx->d_bvh); ctx->d_bvh = nullptr;
                return RT_E_STATE;
            }
        }
        ctx->n_sph = have_sph ? sph->n : 0;
        ctx->n_tri = have_tri ? tri->n : 0;
        ctx->n_nodes = wide ? (uint32_t)(wnodes.size() / (RT_WIDE_NODE_BYTES / 4)) : (uint32_t)n_nodes;
        ctx->bvh_depth = wide ? wide_depth : depth;
        const auto t_end = std::chrono::steady_clock::now();
        ctx->upload_build_ms = std::chrono::duration<double, std::milli>(t_built - t_begin).count();
        ctx->upload_copy_ms = std::chrono::duration<double, std::milli>(t_copied - t_built).count();
        ctx->upload_device_build_ms = device_build ? std::chrono::duration<double, std::milli>(t_end - t_copied).count() : 0.0;
        return RT_OK;
    });
}

int rt_scene_info(const rt_ctx *ctx, uint32_t *n_spheres, uint32_t *n_triangles, uint32_t *n_bvh_nodes,
                  uint32_t *bvh_depth)
{
    if (!ctx) return RT_E_ARG;
    if (n_spheres) *n_spheres = ctx->n_sph;
    if (n_triangles) *n_triangles = ctx->n_tri;
    if (n_bvh_nodes) *n_bvh_nodes = ctx->n_nodes;
    if (bvh_depth) *bvh_depth = (uint32_t)ctx->bvh_depth;
    return RT_OK;
}

int rt_upload_timing(const rt_ctx *ctx, double *host_build_ms, double *device_copy_ms, double *device_build_ms)
{
    if (!ctx) return RT_E_ARG;
    if (host_build_ms) *host_build_ms = ctx->upload_build_ms;
    if (device_copy_ms) *device_copy_ms = ctx->upload_copy_ms;
    if (device_build_ms) *device_build_ms = ctx->upload_device_build_ms;
    return RT_OK;
}

int rt_debug_wave_fill(const rt_ctx *ctx, uint64_t out[6])
{
    if (!ctx || !out) return RT_E_ARG;
    for (int k = 0; k < 6; ++k) out[k] = ctx->fill_counters[k];
    return RT_OK;
}

float rt_debug_round_plane(float v, uint32_t payload, int down)
{
    float out;
    return pack_plane(v, payload & 0xFFu, down != 0, out) ? out : std::nanf("");
}

int rt_debug_block_counts(const rt_ctx *ctx, uint32_t cap, uint64_t *out)
{
    if (!ctx || !out) return RT_E_ARG;
    if (cap < 2 * RT_PROF_BLOCKS) return RT_E_CAP;
    for (int k = 0; k < 2 * RT_PROF_BLOCKS; ++k) out[k] = ctx->block_counters[k];
    return 2 * RT_PROF_BLOCKS;
}

const char *rt_last_kernel_name(const rt_ctx *ctx)
{
    return ctx ? rt_trace_kernel_name(ctx->last_kernel) : "";
}

int rt_render_device(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg, float *d_accum_sum,
                     uint32_t *d_raycount, void *stream_, rt_stats *stats)
{
    if (!ctx || !cam || !d_accum_sum) return RT_E_ARG;
    int rc = check_config(cfg);
    if (rc != RT_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
    rc = enqueue_trace(ctx, cam, cfg, d_accum_sum, d_raycount, stream);
    if (rc != RT_OK) return rc;
    if (stats) return read_stats(ctx, stream, stats);
    return RT_OK;
}

int rt_render_device_noalloc(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg, float *d_accum_sum,
                             uint32_t *d_raycount, void *stream_)
{
    if (!ctx || !cam || !d_accum_sum) return RT_E_ARG;
    int rc = check_config(cfg);
    if (rc != RT_OK) return rc;
    if (!ctx->noalloc_scratch) return RT_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
    unsigned char *ptr = ctx->noalloc_scratch;            // the device address, never the array's name
    TraceScratch sc;
    sc.queue = (uint32_t *)(ptr + NOALLOC_QUEUE_OFF);
    sc.stats = (unsigned long long *)(ptr + NOALLOC_STATS_OFF);
    sc.partial = (float *)(ptr + NOALLOC_PARTIAL_OFF);
    sc.partial_cap = (RT_NOALLOC_SCRATCH_BYTES - NOALLOC_PARTIAL_OFF) / sizeof(float);
    sc.fixed = true;
    return enqueue_trace(ctx, cam, cfg, d_accum_sum, d_raycount, stream, &sc);   // RT_E_CAP: planes do not fit
}

int rt_tonemap_device(rt_ctx *ctx, const float *d_accum_sum, uint32_t n_pixels, uint32_t spp_total,
                      float *d_mean, uint8_t *d_rgb8, void *stream_)
{
    if (!ctx || !d_accum_sum || spp_total == 0) return RT_E_ARG;
    if ((uint64_t)n_pixels * 3 > 0xFFFFFFFFull) return RT_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
    HIP_TRY(ctx, rt_launch_tonemap(d_accum_sum, n_pixels, spp_total, d_mean, d_rgb8, stream));
    return RT_OK;
}

int rt_render(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg, float *accum_mean, uint8_t *rgb8,
              uint32_t *raycount, rt_stats *stats)
{
    if (!ctx || !cam) return RT_E_ARG;
    int rc = check_config(cfg);
    if (rc != RT_OK) return rc;
    if (ctx->n_sph == 0 && ctx->n_tri == 0) return RT_E_STATE;
    ctx->prog_active = false;             // shares d_accum / d_mean / d_rgb8 with a progressive render: that one ends here
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)cfg->nx * cfg->ny;
    if ((rc = ensure(ctx, ctx->d_accum, ctx->accum_cap, npix * 3)) != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_mean, ctx->mean_cap, npix * 3)) != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_rgb8, ctx->rgb8_cap, npix * 3)) != RT_OK) return rc;
    if (raycount && (rc = ensure(ctx, ctx->d_raycount, ctx->raycount_cap, npix)) != RT_OK) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, npix * 3 * sizeof(float), st));
    if (raycount) HIP_TRY(ctx, hipMemsetAsync(ctx->d_raycount, 0, npix * sizeof(uint32_t), st));
    rc = enqueue_trace(ctx, cam, cfg, ctx->d_accum, raycount ? ctx->d_raycount : nullptr, st);
    if (rc != RT_OK) return rc;
    HIP_TRY(ctx, rt_launch_tonemap(ctx->d_accum, (uint32_t)npix, cfg->spp_total, ctx->d_mean, ctx->d_rgb8, st));
    if (accum_mean) HIP_TRY(ctx, hipMemcpyAsync(accum_mean, ctx->d_mean, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    if (rgb8) HIP_TRY(ctx, hipMemcpyAsync(rgb8, ctx->d_rgb8, npix * 3, hipMemcpyDeviceToHost, st));
    if (raycount) HIP_TRY(ctx, hipMemcpyAsync(raycount, ctx->d_raycount, npix * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (stats) { if ((rc = read_stats(ctx, st, stats)) != RT_OK) return rc; }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return RT_OK;
}

int rt_progressive_begin(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg, uint32_t n_passes)
{
    if (!ctx || !cam || n_passes == 0) return RT_E_ARG;
    int rc = check_config(cfg);
    if (rc != RT_OK) return rc;
    if (ctx->n_sph == 0 && ctx->n_tri == 0) return RT_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)cfg->nx * cfg->ny;
    if ((rc = ensure(ctx, ctx->d_accum, ctx->accum_cap, npix * 3)) != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_mean, ctx->mean_cap, npix * 3)) != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_rgb8, ctx->rgb8_cap, npix * 3)) != RT_OK) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, npix * 3 * sizeof(float), ctx->stream));
    const uint32_t span = cfg->spp_end - cfg->spp_begin;
    ctx->prog_cam = *cam; ctx->prog_cfg = *cfg;
    ctx->prog_passes = n_passes > span ? (span ? span : 1u) : n_passes;
    ctx->prog_next = 0;
    std::memset(&ctx->prog_total, 0, sizeof ctx->prog_total);
    ctx->prog_active = true;
    return RT_OK;
}

int rt_progressive_step(rt_ctx *ctx, uint8_t *rgb8, float *accum_mean, uint32_t *spp_done, rt_stats *stats)
{
    if (!ctx) return RT_E_ARG;
    if (!ctx->prog_active) return RT_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const rt_config &cfg = ctx->prog_cfg;
    const size_t npix = (size_t)cfg.nx * cfg.ny;
    // pass k renders the k-th contiguous part of [spp_begin, spp_end): sizes differ by at most one, early passes take the extra
    const uint32_t span = cfg.spp_end - cfg.spp_begin, n = ctx->prog_passes, k = ctx->prog_next;
    const uint32_t base = span / n, extra = span % n;
    rt_config sub = cfg;
    sub.spp_begin = cfg.spp_begin + k * base + (k < extra ? k : extra);
    sub.spp_end = sub.spp_begin + base + (k < extra ? 1 : 0);
    hipStream_t st = ctx->stream;
    int rc = enqueue_trace(ctx, &ctx->prog_cam, &sub, ctx->d_accum, nullptr, st);
    if (rc != RT_OK) { ctx->prog_active = false; return rc; }
    // the picture so far: the mean over the samples rendered up to now (the last pass divides by spp_total, as rt_render does)
    const bool last = k + 1 == n;
    const uint32_t done = sub.spp_end - cfg.spp_begin;
    const uint32_t divisor = last ? cfg.spp_total : (done ? done : 1u);
    HIP_TRY(ctx, rt_launch_tonemap(ctx->d_accum, (uint32_t)npix, divisor, ctx->d_mean, ctx->d_rgb8, st));
    if (rgb8) HIP_TRY(ctx, hipMemcpyAsync(rgb8, ctx->d_rgb8, npix * 3, hipMemcpyDeviceToHost, st));
    if (accum_mean) HIP_TRY(ctx, hipMemcpyAsync(accum_mean, ctx->d_mean, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    rt_stats one;
    if ((rc = read_stats(ctx, st, &one)) != RT_OK) { ctx->prog_active = false; return rc; }
    rt_stats &t = ctx->prog_total;
    t.rays += one.rays; t.primary_rays += one.primary_rays; t.kernel_ms += one.kernel_ms;
    for (int d = 0; d < 20; ++d) t.rays_by_depth[d] += one.rays_by_depth[d];
    if (spp_done) *spp_done = done;
    if (stats) *stats = t;
    ctx->prog_next = k + 1;
    if (last) { ctx->prog_active = false; return RT_OK; }
    return RT_PROGRESS_MORE;
}

int rt_render_reference_style(rt_ctx *ctx, const rt_camera *cam, const rt_config *cfg, uint32_t n_images,
                              uint8_t *rgb8, rt_stats *stats)
{
    if (!ctx || !cam || !rgb8 || n_images == 0 || n_images > 65536) return RT_E_ARG;
    int rc = check_config(cfg);
    if (rc != RT_OK) return rc;
    if (ctx->n_sph == 0 && ctx->n_tri == 0) return RT_E_STATE;
    ctx->prog_active = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)cfg->nx * cfg->ny;
    if ((rc = ensure(ctx, ctx->d_accum, ctx->accum_cap, npix * 3)) != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_rgb8, ctx->rgb8_cap, npix * 3)) != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_raycount, ctx->raycount_cap, npix * 3)) != RT_OK) return rc;   // reused as the u32 sum
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_raycount, 0, npix * 3 * sizeof(uint32_t), st));
    rt_stats total; std::memset(&total, 0, sizeof total);
    for (uint32_t i = 0; i < n_images; ++i) {
        rt_config c = *cfg;
        c.seed = (uint64_t)i * 123u;                              // raytracer.zig:600
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, npix * 3 * sizeof(float), st));
        if ((rc = enqueue_trace(ctx, cam, &c, ctx->d_accum, nullptr, st)) != RT_OK) return rc;
        HIP_TRY(ctx, rt_launch_tonemap(ctx->d_accum, (uint32_t)npix, c.spp_total, nullptr, ctx->d_rgb8, st));   // :527
        HIP_TRY(ctx, rt_launch_u8_sum(ctx->d_raycount, ctx->d_rgb8, (uint32_t)(npix * 3), st));              // :611-614
        rt_stats one;
        if ((rc = read_stats(ctx, st, &one)) != RT_OK) return rc;
        total.rays += one.rays; total.primary_rays += one.primary_rays; total.kernel_ms += one.kernel_ms;
        for (int k = 0; k < 20; ++k) total.rays_by_depth[k] += one.rays_by_depth[k];
    }
    HIP_TRY(ctx, rt_launch_u8_mean(ctx->d_raycount, n_images, ctx->d_rgb8, (uint32_t)(npix * 3), st));      // :616
    HIP_TRY(ctx, hipMemcpyAsync(rgb8, ctx->d_rgb8, npix * 3, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (stats) *stats = total;
    return RT_OK;
}

int rt_render_multi(rt_ctx *const *ctxs, uint32_t n_ctx, const rt_camera *cam, const rt_config *cfg,
                    float *accum_mean, uint8_t *rgb8, uint32_t *raycount, rt_stats *stats)
{
    return guarded([&]() -> int {
    if (!ctxs || n_ctx == 0 || n_ctx > 64 || !cam) return RT_E_ARG;
    int rc = check_config(cfg);
    if (rc != RT_OK) return rc;
    bool distinct = n_ctx > 1;            // every context on a GPU of its own: the exchange step is an RCCL reduce
    std::vector<int> devices(n_ctx);
    for (uint32_t g = 0; g < n_ctx; ++g) {
        if (!ctxs[g]) return RT_E_ARG;
        for (uint32_t k = 0; k < g; ++k) {
            if (ctxs[k] == ctxs[g]) return RT_E_ARG;
            if (ctxs[k]->device == ctxs[g]->device) distinct = false;
        }
        if (ctxs[g]->n_sph == 0 && ctxs[g]->n_tri == 0) return RT_E_STATE;
        devices[g] = ctxs[g]->device;
    }
    for (uint32_t g = 0; g < n_ctx; ++g) ctxs[g]->prog_active = false;    // (uses every ctx's d_accum)
    rt_ctx *root = ctxs[0];
    const size_t npix = (size_t)cfg->nx * cfg->ny, nfl = npix * 3;
    const size_t gather_words = nfl + (raycount ? npix : 0);      // per peer: its sums, then its ray counts
    if (distinct && (rc = get_comms(root, devices)) != RT_OK) return rc;
    // contiguous sample ranges, sizes differ by at most one, low indices take the extra (raytracer.zig:594-600
    // hands every thread its own samples; here GPU g owns global sample indices [b_g, e_g))
    const uint32_t span = cfg->spp_end - cfg->spp_begin, base = span / n_ctx, extra = span % n_ctx;
    // 1. every GPU traces its range into its own SUM buffer (launches are asynchronous: no host threads needed)
    for (uint32_t g = 0; g < n_ctx; ++g) {
        rt_ctx *c = ctxs[g];
        HIP_TRY(c, hipSetDevice(c->device));
        if ((rc = ensure(c, c->d_accum, c->accum_cap, nfl)) != RT_OK) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, nfl * sizeof(float), c->stream));
        if (raycount) {
            if ((rc = ensure(c, c->d_raycount, c->raycount_cap, npix)) != RT_OK) return rc;
            HIP_TRY(c, hipMemsetAsync(c->d_raycount, 0, npix * sizeof(uint32_t), c->stream));
        }
        rt_config sub = *cfg;
        sub.spp_begin = cfg->spp_begin + g * base + (g < extra ? g : extra);
        sub.spp_end = sub.spp_begin + base + (g < extra ? 1 : 0);
        if ((rc = enqueue_trace(c, cam, &sub, c->d_accum, raycount ? c->d_raycount : nullptr, c->stream)) != RT_OK) return rc;
    }
    // 2. the path's one exchange step (the reference's join + sum over threads, raytracer.zig:605-617)
    HIP_TRY(root, hipSetDevice(root->device));
    if ((rc = ensure(root, root->d_mean, root->mean_cap, nfl)) != RT_OK) return rc;
    if ((rc = ensure(root, root->d_rgb8, root->rgb8_cap, nfl)) != RT_OK) return rc;
    if (distinct) {
        // one rank per GPU, ncclReduce(sum) onto rank 0 over xGMI, each on its ctx's stream behind the trace kernel
        RCCL_TRY(root, rccl().GroupStart());
        for (uint32_t g = 0; g < n_ctx; ++g) {
            rt_ctx *c = ctxs[g];
            const hipError_t he = hipSetDevice(c->device);
            if (he != hipSuccess) { (void)rccl().GroupEnd(); c->last_hip = (int)he; return RT_E_HIP; }   // never leave the group open
            int r = rccl().Reduce(c->d_accum, c->d_accum, nfl, Rccl::Float32, Rccl::Sum, 0, root->rccl_comms[g], c->stream);
            if (r == 0 && raycount)
                r = rccl().Reduce(c->d_raycount, c->d_raycount, npix, Rccl::Uint32, Rccl::Sum, 0, root->rccl_comms[g], c->stream);
            if (r != 0) { (void)rccl().GroupEnd(); root->last_rccl = r; return RT_E_NCCL; }
        }
        RCCL_TRY(root, rccl().GroupEnd());
    } else if (n_ctx > 1) {
        // contexts that share a GPU (RCCL takes one rank per GPU): the peers' buffers are copied next to the root's
        // and added there in context order (deterministic)
        if ((rc = ensure(root, root->d_gather, root->gather_cap, gather_words * (n_ctx - 1))) != RT_OK) return rc;
        for (uint32_t g = 1; g < n_ctx; ++g) {
            rt_ctx *c = ctxs[g];
            HIP_TRY(c, hipSetDevice(c->device));
            float *dst = root->d_gather + (size_t)(g - 1) * gather_words;
            const bool same = c->device == root->device;
            if (same) HIP_TRY(c, hipMemcpyAsync(dst, c->d_accum, nfl * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            else HIP_TRY(c, hipMemcpyPeerAsync(dst, root->device, c->d_accum, c->device, nfl * sizeof(float), c->stream));
            if (raycount) {
                if (same) HIP_TRY(c, hipMemcpyAsync(dst + nfl, c->d_raycount, npix * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
                else HIP_TRY(c, hipMemcpyPeerAsync(dst + nfl, root->device, c->d_raycount, c->device, npix * sizeof(uint32_t), c->stream));
            }
        }
    }
    rt_stats total; std::memset(&total, 0, sizeof total);
    for (uint32_t g = 0; g < n_ctx; ++g) {          // join (raytracer.zig:605-607)
        rt_ctx *c = ctxs[g];
        HIP_TRY(c, hipSetDevice(c->device));
        rt_stats one;
        if ((rc = read_stats(c, c->stream, &one)) != RT_OK) return rc;
        total.rays += one.rays; total.primary_rays += one.primary_rays;
        if (one.kernel_ms > total.kernel_ms) total.kernel_ms = one.kernel_ms;     // GPUs run side by side
        for (int k = 0; k < 20; ++k) total.rays_by_depth[k] += one.rays_by_depth[k];
    }
    // 3. root tonemaps and hands the images back
    HIP_TRY(root, hipSetDevice(root->device));
    hipStream_t st = root->stream;
    if (!distinct && n_ctx > 1) {
        HIP_TRY(root, rt_launch_reduce_chunks(root->d_accum, root->d_gather, (uint32_t)npix, n_ctx - 1, gather_words, st));
        if (raycount)
            HIP_TRY(root, rt_launch_u32_add_planes(root->d_raycount, (const uint32_t *)(root->d_gather + nfl), (uint32_t)npix,
                                                   n_ctx - 1, gather_words, st));
    }
    HIP_TRY(root, rt_launch_tonemap(root->d_accum, (uint32_t)npix, cfg->spp_total, root->d_mean, root->d_rgb8, st));
    if (accum_mean) HIP_TRY(root, hipMemcpyAsync(accum_mean, root->d_mean, nfl * sizeof(float), hipMemcpyDeviceToHost, st));
    if (rgb8) HIP_TRY(root, hipMemcpyAsync(rgb8, root->d_rgb8, nfl, hipMemcpyDeviceToHost, st));
    if (raycount) HIP_TRY(root, hipMemcpyAsync(raycount, root->d_raycount, npix * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(root, hipStreamSynchronize(st));
    if (stats) *stats = total;
    return RT_OK;
    });
}

// RCCL smoke: a communicator over this ctx's GPU alone (one rank), one in-place ncclReduce(sum) of n floats, checked.
// What a one-GPU machine can exercise of rt_render_multi's exchange step.
int rt_selftest_rccl(rt_ctx *ctx, uint32_t n)
{
    return guarded([&]() -> int {
    if (!ctx || n == 0 || n > (1u << 24)) return RT_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = get_comms(ctx, std::vector<int>{ ctx->device });
    if (rc != RT_OK) return rc;
    float *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d, (size_t)n * sizeof(float)));
    std::vector<float> h(n), back(n);
    for (uint32_t k = 0; k < n; ++k) h[k] = (float)(k % 1000) * 0.25f;
    hipError_t e = hipMemcpyAsync(d, h.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    int r = 0;
    if (e == hipSuccess) r = rccl().Reduce(d, d, n, Rccl::Float32, Rccl::Sum, 0, ctx->rccl_comms[0], ctx->stream);
    if (e == hipSuccess && r == 0) e = hipMemcpyAsync(back.data(), d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && r == 0) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (r != 0) { ctx->last_rccl = r; return RT_E_NCCL; }
    if (e != hipSuccess) { ctx->last_hip = (int)e; return RT_E_HIP; }
    for (uint32_t k = 0; k < n; ++k) if (back[k] != h[k]) return RT_E_STATE;     // a one-rank sum is the identity
    return RT_OK;
    });
}

int rt_render_and_write(rt_ctx *ctx, const char *fname, const rt_camera *cam, const rt_config *cfg, rt_stats *stats)
{
    return guarded([&]() -> int {
        if (!fname || !cfg) return RT_E_ARG;
        std::vector<uint8_t> img((size_t)cfg->nx * cfg->ny * 3);
        const int rc = rt_render(ctx, cam, cfg, nullptr, img.data(), nullptr, stats);
        if (rc != RT_OK) return rc;
        return rt_write_ppm(fname, img.data(), cfg->nx, cfg->ny);
    });
}

// ---- host-side mirrors ------------------------------------------------------------------------

void rt_camera_basic(rt_camera *cam)
{
    if (cam) *cam = rt::BasicCamera{}.to_abi();
}

void rt_camera_set_position(rt_camera *cam, const float lookfrom[3], const float lookat[3], const float v_up[3],
                            float vfov_deg, float aspect)
{
    if (!cam || !lookfrom || !lookat || !v_up) return;
    rt::MovableCamera c;
    c.SetPosition({ lookfrom[0], lookfrom[1], lookfrom[2] }, { lookat[0], lookat[1], lookat[2] },
                  { v_up[0], v_up[1], v_up[2] }, vfov_deg, aspect);
    *cam = c.to_abi();
}

int rt_random_scene(uint32_t half_size, uint64_t seed, uint32_t cap, float *cx, float *cy, float *cz, float *radius,
                    uint32_t *kind, float *r, float *g, float *b, float *param)
{
    return guarded([&]() -> int {
        // (2 half_size)^2 grid cells are visited whatever `cap` says: bound the work and the memory of a bad argument
        if (half_size > RT_RANDOM_SCENE_MAX_HALF_SIZE) return RT_E_ARG;
        rt::RndGen rng = rt::RndGen::init(seed);
        const rt::Hittable world = rt::Random_Scene(half_size, rng);
        if (world.objects.size() > cap) return RT_E_CAP;
        if (!cx || !cy || !cz || !radius || !kind || !r || !g || !b || !param) return RT_E_ARG;
        soa_out(world.objects, cx, cy, cz, radius, kind, r, g, b, param);
        return (int)world.objects.size();
    });
}

int rt_two_sphere_scene(uint32_t cap, float *cx, float *cy, float *cz, float *radius, uint32_t *kind, float *r,
                        float *g, float *b, float *param)
{
    return guarded([&]() -> int {
        const rt::Hittable world = rt::Two_Sphere_Scene();
        if (world.objects.size() > cap) return RT_E_CAP;
        if (!cx || !cy || !cz || !radius || !kind || !r || !g || !b || !param) return RT_E_ARG;
        soa_out(world.objects, cx, cy, cz, radius, kind, r, g, b, param);
        return (int)world.objects.size();
    });
}

int rt_tessellate_spheres(const rt_spheres *sph, int slices, int stacks, uint32_t cap, float *v0, float *v1,
                          float *v2, uint32_t *material)
{
    return guarded([&]() -> int {
        if (!sph || !sph->cx || !sph->cy || !sph->cz || !sph->radius) return RT_E_ARG;
        // at most 2 * slices * stacks triangles per sphere, and rt_upload_scene takes fewer than 2^26: a larger request is
        // refused before any memory is touched
        {
            uint64_t worst = 0;
            for (uint32_t k = 0; k < sph->n; ++k) {
                uint64_t sl = slices > 0 && stacks > 0 ? (uint64_t)slices : (sph->radius[k] > 50 ? 150u : sph->radius[k] < 1.0f ? 20u : 30u);
                uint64_t st = slices > 0 && stacks > 0 ? (uint64_t)stacks : sl;
                worst += 2 * sl * st;
                if (worst >= (1ull << 26)) return RT_E_ARG;
            }
        }
        std::vector<rt::Sphere> objs(sph->n);
        for (uint32_t k = 0; k < sph->n; ++k) {
            objs[k].center = { sph->cx[k], sph->cy[k], sph->cz[k] };
            objs[k].radius = sph->radius[k];
        }
        const std::vector<rt::Triangle> tris = rt::Tessellate_Spheres(objs, slices, stacks);
        if (cap == 0) return (int)tris.size();
        if (tris.size() > cap) return RT_E_CAP;
        if (!v0 || !v1 || !v2 || !material) return RT_E_ARG;
        for (size_t k = 0; k < tris.size(); ++k) {
            v0[3 * k] = tris[k].v0.x; v0[3 * k + 1] = tris[k].v0.y; v0[3 * k + 2] = tris[k].v0.z;
            v1[3 * k] = tris[k].v1.x; v1[3 * k + 1] = tris[k].v1.y; v1[3 * k + 2] = tris[k].v1.z;
            v2[3 * k] = tris[k].v2.x; v2[3 * k + 1] = tris[k].v2.y; v2[3 * k + 2] = tris[k].v2.z;
            material[k] = tris[k].material;
        }
        return (int)tris.size();
    });
}

int rt_write_ppm(const char *fname, const uint8_t *rgb8, uint32_t nx, uint32_t ny)
{
    return guarded([&]() -> int {
        if (!fname || !rgb8) return RT_E_ARG;
        return rt::img_io::write_ppm_image(fname, rgb8, nx, ny) ? RT_OK : RT_E_IO;
    });
}

int rt_selftest_cbrt_grid(rt_ctx *ctx, float *out, uint64_t *n_risky)
{
    if (!ctx || !out) return RT_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << 24;
    float *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d, n * sizeof(float) + 256));
    unsigned long long *dc = (unsigned long long *)(d + n);
    hipError_t e;
    unsigned long long cnt = 0;
    do {
        if ((e = hipMemsetAsync(dc, 0, sizeof *dc, ctx->stream)) != hipSuccess) break;
        if ((e = rt_launch_selftest_cbrt_grid(d, dc, ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(out, d, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(&cnt, dc, sizeof cnt, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    (void)hipFree(d);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return RT_E_HIP; }
    if (n_risky) *n_risky = cnt;
    return RT_OK;
}

int rt_selftest_math(rt_ctx *ctx, uint32_t n, const float *a, const float *b, float *out_div, float *out_sqrt,
                     float *out_cbrt, const uint32_t *ctr0, uint32_t *philox_out)
{
    if (!ctx || !a || !b || !out_div || !out_sqrt || !out_cbrt || !ctr0 || !philox_out) return RT_E_ARG;
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d = nullptr;
    const size_t words = (size_t)n * 10;   // a, b, div, sqrt, cbrt, ctr0, philox[4]
    HIP_TRY(ctx, hipMalloc((void **)&d, words * 4));
    float *da = d, *db = d + n, *ddiv = d + 2 * (size_t)n, *dsq = d + 3 * (size_t)n, *dcb = d + 4 * (size_t)n;
    uint32_t *dc = (uint32_t *)(d + 5 * (size_t)n), *dp = (uint32_t *)(d + 6 * (size_t)n);
    int rc = RT_OK;
    hipError_t e;
    do {
        if ((e = hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = hipMemcpy(dc, ctr0, n * 4, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = rt_launch_selftest(n, da, db, ddiv, dsq, dcb, dc, dp, ctx->stream)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpy(out_div, ddiv, n * 4, hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(out_sqrt, dsq, n * 4, hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(out_cbrt, dcb, n * 4, hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(philox_out, dp, (size_t)n * 16, hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    if (e != hipSuccess) { ctx->last_hip = (int)e; rc = RT_E_HIP; }
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
