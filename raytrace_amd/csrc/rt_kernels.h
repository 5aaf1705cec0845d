// Host<->device launch interface of the trace kernels (internal; the public boundary is include/rt_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct TraceParams {
    float    ll[3], hz[3], vt[3], org[3];   // camera.zig:21-24
    uint32_t sub_origin;                    // camera.zig:39 vs :16
    uint32_t nx, ny;
    uint32_t spp_begin, spp_end;            // global sample indices [begin, end)
    uint32_t max_depth;
    float    t_min;
    uint32_t key0, key1;                    // Philox key = seed
    uint32_t x0, y0, x1, y1;                // output rect
    uint32_t tiles_x;                       // 8x8-pixel tiles across the rect
    uint32_t n_items;                       // work items = tiles * 64 (one per pixel slot)
    uint32_t q3;                            // 1: j = ny - jr (raytracer.zig:523)
    uint32_t n_prim;
};

// Device scene (sphere SoA, 16-byte records so one scalar/vector load fetches a whole record):
//   geom[i]   = { cx, cy, cz, radius*radius }       -- all the inner loop reads
//   aux[i]    = { radius, bits(kind), param, 0 }    -- read once per hit
//   albedo[i] = { r, g, b, 0 }
struct DeviceSpheres {
    const float4 *geom;
    const float4 *aux;
    const float4 *albedo;
    uint32_t      n;
};

// stats buffer layout (unsigned long long[]): [0] rays, [1] primary rays, [2..21] rays by depth
enum { RT_STATS_WORDS = 22 };

hipError_t rt_launch_trace_spheres(const TraceParams &p, const DeviceSpheres &s, float *accum_sum,
                                   uint32_t *raycount, uint32_t *queue, unsigned long long *stats,
                                   bool histogram, int grid_blocks, hipStream_t stream);
int rt_trace_spheres_blocks_per_cu(bool histogram);

hipError_t rt_launch_tonemap(const float *accum_sum, uint32_t n_pixels, uint32_t spp_total,
                             float *mean, uint8_t *rgb8, hipStream_t stream);

hipError_t rt_launch_selftest(uint32_t n, const float *a, const float *b, float *out_div, float *out_sqrt,
                              float *out_cbrt, const uint32_t *ctr0, uint32_t *philox_out, hipStream_t stream);
