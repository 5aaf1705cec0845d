// Trace kernels for gfx950 (MI355X): Render / trace / color of the reference
// (raytracer.zig:517-572) as one persistent wave64 kernel.
//
// Shape of the kernel
//   * persistent workers: the grid is sized to fill the chip once; every lane owns one pixel at a
//     time and walks its samples in order (so the per-pixel sum has the reference's order), pulling
//     the next pixel from a global queue with one wave-aggregated atomic (ballot + popcount).
//   * the recursion of color() is a per-lane state machine: one loop iteration = one world.hit.
//     A lane whose path ended (miss, absorbed, depth cap) regenerates its next camera ray in the
//     same iteration, so every lane of the wave carries a live ray into the intersection loop
//     regardless of how bounce depths diverge.
//   * the intersection loop runs over the sphere records with a wave-uniform index: the 16-byte
//     records come through the scalar cache into SGPRs (s_load_dwordx4), the VALU does only the
//     discriminant arithmetic, and the loop has no divergence except the rare root evaluation.
//   * no MFMA: this is branchy f32 VALU work.
//
// Arithmetic contract: see rt_device.h.
#include "rt_kernels.h"
#include "rt_device.h"

using namespace rtd;

namespace {

constexpr uint32_t NOPIX = 0xFFFFFFFFu;
constexpr uint32_t NOHIT = 0xFFFFFFFFu;
constexpr float    T_MAX = 3.402823466e+38f;   // std.math.floatMax(f32), raytracer.zig:555

enum : uint32_t { K_LAMBERTIAN = 0, K_METAL = 1, K_DIELECTRIC = 2 };

struct Lane {
    f3       o, d;        // current ray (direction never normalised, as in the reference)
    f3       thr;         // product of attenuations so far
    f3       col;         // running sum over this pixel's samples (trace: raytracer.zig:535-541)
    uint32_t pix;         // output index jr*nx+i
    uint32_t i, j;        // pixel column and the reference's j (raytracer.zig:523)
    uint32_t sample;      // global sample index
    uint32_t depth;
    uint32_t nrays;       // world.hit evaluations of this pixel
};

// trace's per-sample prologue: raytracer.zig:537-539 and camera.zig:16/:39
__device__ __forceinline__ void camera_ray(const TraceParams &P, Lane &L)
{
    uint32_t r[4];
    philox4x32_10(L.pix, L.sample, 0u, 0u, P.key0, P.key1, r);
    const float u = ((float)L.i + u32_to_unit_float(r[0])) / (float)P.nx;
    const float v = ((float)L.j + u32_to_unit_float(r[1])) / (float)P.ny;
    f3 dir = mk(P.ll[0], P.ll[1], P.ll[2]) + u * mk(P.hz[0], P.hz[1], P.hz[2]) + v * mk(P.vt[0], P.vt[1], P.vt[2]);
    const f3 org = mk(P.org[0], P.org[1], P.org[2]);
    if (P.sub_origin) dir = dir - org;
    L.o = org; L.d = dir;
    L.thr = mk(1.0f, 1.0f, 1.0f);
    L.depth = 0;
}

// Everything color() does after world.hit returned (raytracer.zig:555-571) for one lane.
// Returns true when the path continues with a new ray in L.o/L.d.
__device__ __forceinline__ bool shade(const TraceParams &P, Lane &L, bool hit, float t, f3 normal_num,
                                      float4 aux, float4 alb, float a_dd)
{
    const float len_d = sqrtf(a_dd);                 // vector.zig:10-13 length(direction)
    if (!hit) {                                       // sky, raytracer.zig:569-571
        const f3 unit = L.d / len_d;
        const float tt = 0.5f * (unit.y + 1.0f);
        const f3 sky = mk((1.0f - tt) + tt * 0.5f, (1.0f - tt) + tt * 0.7f, (1.0f - tt) + tt * 1.0f);
        L.col = L.col + L.thr * sky;
        return false;
    }
    if (L.depth > P.max_depth) return false;          // raytracer.zig:556-558 (after the hit test)

    const float radius = aux.x;
    const uint32_t kind = __float_as_uint(aux.y);
    const float param = aux.z;
    const f3 p = L.o + t * L.d;                       // ray.zig:10-12
    const f3 n = (p - normal_num) / radius;           // raytracer.zig:60 (normal_num holds the centre)
    const f3 albedo = mk(alb.x, alb.y, alb.z);

    uint32_t r[4];
    philox4x32_10(L.pix, L.sample, L.depth + 1u, 0u, P.key0, P.key1, r);

    f3 dir;
    if (kind == K_DIELECTRIC) {                       // Dielectric.scatter, raytracer.zig:469-499
        const f3 unit = L.d / len_d;
        const float dn_u = dot(unit, n);
        const f3 reflected = unit - (2.0f * dn_u) * n;             // :450-452
        const float ddn = dot(L.d, n);
        f3 on; float ni_over_nt, cosine;
        if (ddn > 0) { on = -n; ni_over_nt = param; cosine = (param * ddn) / len_d; }
        else { on = n; ni_over_nt = 1.0f / param; cosine = (-ddn) / len_d; }
        const float dt = dot(unit, on);                            // refract, :453-463
        const float disc = 1.0f - (ni_over_nt * ni_over_nt) * (1.0f - (dt * dt));
        float reflect_prob = 1.0f;
        f3 refracted = mk(0.0f, 0.0f, 0.0f);
        if (disc > 0) {
            const float sq = sqrtf(disc);
            const f3 w = unit - dt * on;   // uv - n*splat(dt)
            refracted = mk(ni_over_nt * w.x - on.x * sq, ni_over_nt * w.y - on.y * sq, ni_over_nt * w.z - on.z * sq);
            float r0 = (1.0f - param) / (1.0f + param);            // schlick, :464-468
            r0 = r0 * r0;
            reflect_prob = r0 + (1.0f - r0) * pow5(1.0f - cosine);
        }
        dir = (u32_to_unit_float(r[0]) < reflect_prob) ? reflected : refracted;   // :492-496
    } else {
        // random_in_unit_sphere, raytracer.zig:574-578
        const f3 cube = mk(u32_to_unit_float(r[0]), u32_to_unit_float(r[1]), u32_to_unit_float(r[2]));
        const f3 cn = cube / sqrtf(dot(cube, cube));
        const float cb = cbrt_f32(u32_to_unit_float(r[3]));
        const f3 rius = mk((2.0f * cn.x - 1.0f) * cb, (2.0f * cn.y - 1.0f) * cb, (2.0f * cn.z - 1.0f) * cb);
        if (kind == K_LAMBERTIAN) {                   // Lambertian.scatter, raytracer.zig:438-445
            const f3 target = (p + n) + rius;
            dir = target - p;
        } else {                                      // Metal.scatter, raytracer.zig:423-433
            const f3 unit = L.d / len_d;
            const float dn_u = dot(unit, n);
            const f3 reflected = unit - (2.0f * dn_u) * n;
            dir = reflected + param * rius;
            if (!(dot(dir, n) > 0)) return false;     // absorbed: the path returns black
        }
    }
    L.thr = L.thr * albedo;
    L.o = p; L.d = dir;
    L.depth += 1;
    return true;
}

template <bool HIST>
__global__ __launch_bounds__(256) void rt_trace_spheres_kernel(
    const TraceParams P,
    const float4 *__restrict__ geom, const float4 *__restrict__ aux, const float4 *__restrict__ albedo,
    float *__restrict__ accum, uint32_t *__restrict__ raycount,
    uint32_t *__restrict__ queue, unsigned long long *__restrict__ stats)
{
    __shared__ uint32_t s_hist[20];
    if (HIST) {
        if (threadIdx.x < 20) s_hist[threadIdx.x] = 0;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;

    Lane L;
    L.o = mk(0, 0, 0); L.d = mk(0, 0, 1); L.thr = mk(0, 0, 0); L.col = mk(0, 0, 0);
    L.pix = NOPIX; L.i = 0; L.j = 0; L.sample = P.spp_end; L.depth = 0; L.nrays = 0;
    bool alive = true, has_ray = false;
    unsigned long long my_rays = 0, my_primary = 0;

    for (;;) {
        if (!has_ray && alive) {
            if (L.sample >= P.spp_end) {
                if (L.pix != NOPIX) {                 // pixel finished: this lane is its only writer
                    float *dst = accum + 3ull * L.pix;
                    dst[0] += L.col.x; dst[1] += L.col.y; dst[2] += L.col.z;
                    if (raycount) raycount[L.pix] += L.nrays;
                    my_rays += L.nrays;
                    L.pix = NOPIX;
                }
                // wave-aggregated dequeue: one atomic for all lanes of the wave that need a pixel
                const unsigned long long m = __ballot(1);
                const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(queue, (uint32_t)__popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);   // leader = first active lane
                const uint32_t item = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (item >= P.n_items) {
                    alive = false;
                } else {
                    // item -> pixel: 8x8 tiles, row-major inside a tile, so a wave starts on one tile
                    const uint32_t tile = item >> 6, within = item & 63u;
                    const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
                    const uint32_t i = P.x0 + tx * 8u + (within & 7u);
                    const uint32_t jr = P.y0 + ty * 8u + (within >> 3);
                    if (i < P.x1 && jr < P.y1) {
                        L.pix = jr * P.nx + i;        // raytracer.zig:521-528 output index
                        L.i = i;
                        L.j = P.q3 ? (P.ny - jr) : (P.ny - 1u - jr);   // raytracer.zig:523
                        L.sample = P.spp_begin;
                        L.col = mk(0, 0, 0);
                        L.nrays = 0;
                    }
                }
            }
            if (L.pix != NOPIX && L.sample < P.spp_end) {
                camera_ray(P, L);
                has_ray = true;
                my_primary += 1;
            }
        }
        if (__ballot(alive) == 0ull) break;           // every lane has drained the queue

        // ---- world.hit: closest root over all spheres (hittableList semantics, raytracer.zig:92-113;
        //      Sphere.hit, raytracer.zig:42-78).  Lanes without a ray run the loop on dummy data.
        const float a = dot(L.d, L.d);
        float best_t = T_MAX;
        uint32_t best = NOHIT;
        for (uint32_t s = 0; s < P.n_prim; ++s) {
            const float4 g = geom[s];                 // wave-uniform address -> scalar load
            const f3 oc = mk(L.o.x - g.x, L.o.y - g.y, L.o.z - g.z);
            const float b = dot(oc, L.d);
            const float c = dot(oc, oc) - g.w;        // g.w = radius*radius (same product, hoisted to upload)
            const float disc = b * b - a * c;
            if (disc > 0) {
                const float sq = sqrtf(disc);
                float t = (-b - sq) / a;
                if (!((t < best_t) && (t > P.t_min))) t = (-b + sq) / a;
                if ((t < best_t) && (t > P.t_min)) { best_t = t; best = s; }
            }
        }

        if (has_ray) {
            L.nrays += 1;
            if (HIST) atomicAdd(&s_hist[L.depth < 19u ? L.depth : 19u], 1u);
            const bool hit = best != NOHIT;
            float4 hg = make_float4(0, 0, 0, 1), ha = make_float4(1, 0, 0, 0), hc = make_float4(0, 0, 0, 0);
            if (hit) { hg = geom[best]; ha = aux[best]; hc = albedo[best]; }
            const bool cont = shade(P, L, hit, best_t, mk(hg.x, hg.y, hg.z), ha, hc, a);
            if (!cont) { L.sample += 1; has_ray = false; }
        }
    }

    atomicAdd(&stats[0], my_rays);
    atomicAdd(&stats[1], my_primary);
    if (HIST) {
        __syncthreads();
        if (threadIdx.x < 20) atomicAdd(&stats[2 + threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    }
}

// mean = sum / spp_total (raytracer.zig:542), sqrt gamma (:545-549), 255.99 truncation (:527)
__global__ __launch_bounds__(256) void rt_tonemap_kernel(const float *accum_sum, uint32_t n_values, float spp_total,
                                                         float *mean, uint8_t *rgb8)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_values) return;
    const float m = accum_sum[k] / spp_total;
    if (mean) mean[k] = m;
    if (rgb8) {
        const float q = 255.99f * sqrtf(m);
        int iv = (int)q;
        iv = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
        rgb8[k] = (uint8_t)iv;
    }
}

__global__ void rt_selftest_kernel(uint32_t n, const float *a, const float *b, float *out_div, float *out_sqrt,
                                   float *out_cbrt, const uint32_t *ctr0, uint32_t *philox_out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    out_div[k] = a[k] / b[k];
    out_sqrt[k] = sqrtf(a[k]);
    out_cbrt[k] = cbrt_f32(a[k]);
    uint32_t r[4];
    philox4x32_10(ctr0[k], k, 7u, 0u, 0x12345678u, 0x9abcdef0u, r);
    philox_out[4 * k + 0] = r[0]; philox_out[4 * k + 1] = r[1];
    philox_out[4 * k + 2] = r[2]; philox_out[4 * k + 3] = r[3];
}

}  // namespace

int rt_trace_spheres_blocks_per_cu(bool histogram)
{
    int nb = 0;
    hipError_t e = histogram
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rt_trace_spheres_kernel<true>, 256, 0)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rt_trace_spheres_kernel<false>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 4;
    return nb > 8 ? 8 : nb;
}

hipError_t rt_launch_trace_spheres(const TraceParams &p, const DeviceSpheres &s, float *accum_sum,
                                   uint32_t *raycount, uint32_t *queue, unsigned long long *stats,
                                   bool histogram, int grid_blocks, hipStream_t stream)
{
    if (histogram)
        hipLaunchKernelGGL(rt_trace_spheres_kernel<true>, dim3(grid_blocks), dim3(256), 0, stream,
                           p, s.geom, s.aux, s.albedo, accum_sum, raycount, queue, stats);
    else
        hipLaunchKernelGGL(rt_trace_spheres_kernel<false>, dim3(grid_blocks), dim3(256), 0, stream,
                           p, s.geom, s.aux, s.albedo, accum_sum, raycount, queue, stats);
    return hipGetLastError();
}

hipError_t rt_launch_tonemap(const float *accum_sum, uint32_t n_pixels, uint32_t spp_total,
                             float *mean, uint8_t *rgb8, hipStream_t stream)
{
    const uint32_t n = n_pixels * 3u;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_tonemap_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream,
                       accum_sum, n, (float)spp_total, mean, rgb8);
    return hipGetLastError();
}

hipError_t rt_launch_selftest(uint32_t n, const float *a, const float *b, float *out_div, float *out_sqrt,
                              float *out_cbrt, const uint32_t *ctr0, uint32_t *philox_out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_selftest_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream,
                       n, a, b, out_div, out_sqrt, out_cbrt, ctr0, philox_out);
    return hipGetLastError();
}
