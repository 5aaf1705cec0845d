// Host<->device launch interface of the trace kernels (internal; the public boundary is include/rt_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct TraceParams {
    float    ll[3], hz[3], vt[3], org[3];   // camera.zig:21-24
    uint32_t sub_origin;                    // camera.zig:39 vs :16
    uint32_t nx, ny;
    float    inv_nx, inv_ny;                // RN(1/nx), RN(1/ny) for div_recip; cam_fast = 0 sends the camera ray down the IEEE divide
    uint32_t cam_fast;
    uint32_t spp_begin, spp_end;            // global sample indices [begin, end)
    uint32_t max_depth;
    float    t_min;
    uint32_t key0, key1;                    // Philox key = seed
    uint32_t x0, y0, x1, y1;                // output rect
    uint32_t tiles_x;                       // 8x8-pixel tiles across the rect
    uint32_t n_items;                       // work items = tiles * 64 (one per pixel slot)
    uint32_t q3;                            // 1: j = ny - jr (raytracer.zig:523)
    uint32_t n_prim;
    uint32_t spp_chunk, chunk0, n_chunks;   // n_chunks > 1: samples summed per chunk into partial planes
    uint32_t queue_batch;                   // work items a wave reserves per global atomic (>= 64)
    size_t   plane_stride;                  // floats between partial planes
    uint32_t force_global_scene;            // BVH kernels: 1 = never stage the scene in LDS (test/tuning knob)
    uint32_t flush_lanes;                   // BVH kernels: run the exact tests when this many lanes hold candidates
    uint32_t keep_traversing;               // BVH kernels: leave the traversal loop when fewer lanes than this traverse
};

// Device scene (made at upload, rt_api.cpp rt_upload_scene).  16-byte records so one scalar/vector load fetches a whole record.
//   spheres:   geom[i]       = { cx, cy, cz, radius*radius }   (all the intersection code reads)
//              aux[i]        = { radius, bits(kind), param, RN(1/radius) or NaN (no fast reciprocal: IEEE divide) }
//                              (material i = sphere i)
//              albedo[2i]    = { r, g, b, schlick r0^2 },  albedo[2i+1] = { 1/index, 0, 0, 0 }   (TWO records per material)
//   triangles: tri[3i+0] = { v0.xyz, bits(material) }, tri[3i+1] = { v1.xyz, 0 }, tri[3i+2] = { v2.xyz, 0 }
//              aux/albedo are the material table (aux.x and aux.w unused)
//   bvh:       RT_NODE_BYTES (80) per node = five float4, root = node 0.  The host builder's 64-byte node (rt_bvh.hpp Node)
//              is converted to this layout at upload:
// Device BVH node, RT_NODE_BYTES = 80: per axis (x, y, z; 24 bytes each) the planes of the two child boxes as
// {hi0, hi1, lo0, lo1, hi0, hi1} -- a lane reads 16 bytes at +8 (ray runs up the axis) or +0 (down) and gets
// {near0, near1, far0, far1} without a min/max --, then the two child refs at byte 72: bit 31 set = primitive index,
// else the BYTE offset of the child node.  Made from the host tree (rt_bvh.hpp) at upload, rt_api.cpp.
constexpr int RT_NODE_BYTES = 80;
// Packed node of large triangle scenes, 72 bytes: the three axis records only; child c's 24-bit ref (bit 23 = leaf:
// primitive index below it, else the child's node INDEX) is spread over the low mantissa bytes of its planes -- byte k
// of the ref in BOTH lo_c and hi_c of axis k (x, y, z), planes rounded outward to the next float carrying that byte.
constexpr int RT_PACKED_NODE_BYTES = 72;
constexpr uint32_t RT_PACKED_LEAF_BIT = 0x800000u;
// Wide node of large triangle scenes (rt_tuning.node_format = RT_NODES_WIDE, the default for large meshes), 144 bytes: two levels of the binary tree collapsed into one
// node of up to four children -- half as many dependent node fetches per ray.  Per axis (48 bytes)
// {hi0..hi3, lo0..lo3, hi0..hi3}: a lane reads 32 bytes at +16 (ray runs up the axis) or +0 (down) and gets
// {near0..near3, far0..far3}.  Refs as in the packed node (24 bits in the low mantissa bytes of the child's planes;
// inner = index of a wide node).  A missing child has lo = +inf, hi = -inf (never hit) and a zero ref.
constexpr int RT_WIDE_NODE_BYTES = 144;
constexpr int RT_WIDE_STACK = 34;        // 3 pushes per level, leaves included: trees of at most 11 wide levels (binary depth cap 22)

struct DeviceScene {
    const float4 *geom = nullptr;
    const float4 *tri = nullptr;
    const float4 *aux = nullptr;
    const float4 *albedo = nullptr;
    const float4 *bvh = nullptr;
    uint32_t      n_prim = 0;
    uint32_t      n_nodes = 0;
    int           bvh_depth = 0;
    bool          packed = false;        // bvh holds RT_PACKED_NODE_BYTES nodes (triangles only)
    bool          wide = false;          // bvh holds RT_WIDE_NODE_BYTES nodes (triangles only); bvh_depth = wide levels
};

enum TraceKernel { RT_K_BRUTE_SPHERES = 0, RT_K_BVH_SPHERES = 1, RT_K_BVH_TRIANGLES = 2 };

// stats buffer layout (unsigned long long[]): [0] rays, [1] primary rays, [2..21] rays by depth,
// [22..27] BVH wave-fill counters of histogram builds: wave/lane traversal steps, wave/lane exact-test
// iterations, wave/lane shading passes
// Profiling builds (RT_FLAG_DEPTH_HISTOGRAM) also count how often the blocks of the shading pass run (wave level) and
// with how many lanes; stats[28 + 2*b] = wave-level executions of block b, stats[29 + 2*b] = lanes that ran it.
enum { RT_PB_SURFACE, RT_PB_UNIT, RT_PB_SKY, RT_PB_DEQUEUE, RT_PB_PHILOX, RT_PB_CAMERA, RT_PB_RIUS, RT_PB_CBRT_SLOW,
       RT_PB_LAMBERTIAN, RT_PB_METAL, RT_PB_DIELECTRIC, RT_PB_NORMAL_SLOW, RT_PROF_BLOCKS };
enum { RT_HIST_WORDS = 20, RT_STATS_WORDS = 28 + 2 * RT_PROF_BLOCKS };

hipError_t rt_launch_trace(TraceKernel which, const TraceParams &p, const DeviceScene &s, float *accum_sum,
                           float *partial, uint32_t *raycount, uint32_t *queue, unsigned long long *stats,
                           bool histogram, int num_cus, hipStream_t stream);
const char *rt_trace_kernel_name(TraceKernel which);

// rt_build.hip: the (n - 1)-node tree of the scene already on the device, built there.  n >= 2; nodes holds
// (n - 1) * RT_NODE_BYTES bytes; depth_cap <= 32 bounds the inner-node levels (raised to ceil(log2 n) if below);
// *depth = levels used; *bad != 0: a primitive is not finite.  Synchronises.
hipError_t rt_device_build_bvh(bool triangles, uint32_t n, const float4 *geom, const float4 *aux, const float4 *tri,
                               float4 *nodes, uint32_t depth_cap, int *depth, uint32_t *bad, hipStream_t stream);

hipError_t rt_launch_clear_pass(uint32_t *queue, unsigned long long *stats, hipStream_t stream);

hipError_t rt_launch_reduce_chunks(float *accum_sum, const float *partial, uint32_t n_pixels, uint32_t n_chunks,
                                   size_t plane_stride, hipStream_t stream);

hipError_t rt_launch_u32_add_planes(uint32_t *dst, const uint32_t *planes, uint32_t n, uint32_t n_planes,
                                    size_t plane_stride, hipStream_t stream);

hipError_t rt_launch_tonemap(const float *accum_sum, uint32_t n_pixels, uint32_t spp_total,
                             float *mean, uint8_t *rgb8, hipStream_t stream);

hipError_t rt_launch_u8_sum(uint32_t *sum, const uint8_t *rgb8, uint32_t n, hipStream_t stream);
hipError_t rt_launch_u8_mean(const uint32_t *sum, uint32_t n_images, uint8_t *rgb8, uint32_t n, hipStream_t stream);

hipError_t rt_launch_selftest_cbrt_grid(float *out, unsigned long long *n_risky, hipStream_t stream);
hipError_t rt_launch_selftest(uint32_t n, const float *a, const float *b, float *out_div, float *out_sqrt,
                              float *out_cbrt, const uint32_t *ctr0, uint32_t *philox_out, hipStream_t stream);
