// Trace kernels for gfx950 (MI355X): Render / trace / color of the reference
// (raytracer.zig:517-572) as one persistent wave64 kernel, in three intersection flavours.
//
// Shape of the kernel
//   * persistent workers: the grid is sized to fill the chip once; every lane owns one pixel at a
//     time and walks its samples in order (so the per-pixel sum has the reference's order), pulling
//     the next pixel from a global queue with one wave-aggregated atomic (ballot + popcount).
//   * the recursion of color() is a per-lane state machine: one outer iteration = one world.hit.
//     A lane whose path ended (miss, absorbed, depth cap) regenerates its next camera ray in the
//     same iteration, so every lane of the wave carries a live ray into the intersection step
//     regardless of how bounce depths diverge.
//   * world.hit, brute force (small scenes): a loop over the sphere records with a wave-uniform
//     index; the 16-byte records come through the scalar cache into SGPRs and the VALU does only the
//     discriminant arithmetic.
//   * world.hit, BVH (spheres or triangles): per-lane stack traversal of a binary BVH whose nodes
//     carry both child boxes, stack and candidate lists in LDS.  The box test only CULLS (padded
//     boxes); leaves are queued per lane and the exact primitive tests run in wave-wide batches, so
//     the expensive IEEE sqrt/divide code is executed by many lanes at once instead of once per
//     divergent leaf visit.  The answer is the list traversal's closest hit (raytracer.zig:92-113):
//     lexicographic minimum of (t, primitive index).
//   * no MFMA: this is branchy f32 VALU work.
//
// Arithmetic contract: see rt_device.h.
#include "rt_kernels.h"
#include "rt_device.h"

using namespace rtd;

namespace {

constexpr uint32_t NOPIX = 0xFFFFFFFFu;
constexpr uint32_t NOHIT = 0xFFFFFFFFu;
constexpr uint32_t DONE = 0xFFFFFFFFu;
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr float    T_MAX = 3.402823466e+38f;   // std.math.floatMax(f32), raytracer.zig:555
constexpr int      BLOCK = 256;
constexpr int      CAND = 8;                   // per-lane candidate queue length (LDS)

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

struct Surface {          // HitRecord (raytracer.zig:31-36) with the material flattened
    f3       p, n;
    uint32_t kind;
    float    param;
    f3       albedo;
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

// sky, raytracer.zig:569-571
__device__ __forceinline__ void shade_miss(Lane &L, float a_dd)
{
    const float len_d = sqrtf(a_dd);
    const f3 unit = L.d / len_d;
    const float tt = 0.5f * (unit.y + 1.0f);
    const f3 sky = mk((1.0f - tt) + tt * 0.5f, (1.0f - tt) + tt * 0.7f, (1.0f - tt) + tt * 1.0f);
    L.col = L.col + L.thr * sky;
}

// Material.scatter (raytracer.zig:407-414) for one lane.  Returns true when the path continues
// with the scattered ray in L.o/L.d.
__device__ __forceinline__ bool shade_hit(const TraceParams &P, Lane &L, const Surface &S, float a_dd)
{
    const float len_d = sqrtf(a_dd);                 // vector.zig:10-13 length(direction)
    const f3 p = S.p, n = S.n;
    uint32_t r[4];
    philox4x32_10(L.pix, L.sample, L.depth + 1u, 0u, P.key0, P.key1, r);

    f3 dir;
    if (S.kind == K_DIELECTRIC) {                     // Dielectric.scatter, raytracer.zig:469-499
        const float param = S.param;
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
        if (S.kind == K_LAMBERTIAN) {                 // Lambertian.scatter, raytracer.zig:438-445
            const f3 target = (p + n) + rius;
            dir = target - p;
        } else {                                      // Metal.scatter, raytracer.zig:423-433
            const f3 unit = L.d / len_d;
            const float dn_u = dot(unit, n);
            const f3 reflected = unit - (2.0f * dn_u) * n;
            dir = reflected + S.param * rius;
            if (!(dot(dir, n) > 0)) return false;     // absorbed: the path returns black
        }
    }
    L.thr = L.thr * S.albedo;
    L.o = p; L.d = dir;
    L.depth += 1;
    return true;
}

// ---- exact primitive tests -------------------------------------------------------------------------
// Sphere.hit (raytracer.zig:42-78) against a running closest hit.  Inside a list the reference accepts
// the near root if it lies in (t_min, closest) and otherwise the far root; that equals "first root
// above t_min, accepted iff below closest", which is independent of the visiting order.  Ties in t go
// to the lower index (first in list order).
__device__ __forceinline__ void sphere_exact(float4 g, uint32_t prim, const Lane &L, float a, float t_min,
                                             float &best_t, uint32_t &best)
{
    const f3 oc = mk(L.o.x - g.x, L.o.y - g.y, L.o.z - g.z);
    const float b = dot(oc, L.d);
    const float c = dot(oc, oc) - g.w;                // g.w = radius*radius (raytracer.zig:51)
    const float disc = b * b - a * c;
    if (disc > 0) {
        const float sq = sqrtf(disc);
        float t = (-b - sq) / a;
        if (!(t > t_min)) t = (-b + sq) / a;
        if ((t > t_min) && ((t < best_t) || (t == best_t && prim < best))) { best_t = t; best = prim; }
    }
}

// Build-defined two-sided Moller-Trumbore (the reference has no triangle code); same operation order
// as the CPU checker.
__device__ __forceinline__ void triangle_exact(f3 v0, f3 v1, f3 v2, uint32_t prim, const Lane &L, float t_min,
                                               float &best_t, uint32_t &best)
{
    const f3 e1 = v1 - v0, e2 = v2 - v0;
    const f3 pvec = cross(L.d, e2);
    const float det = dot(e1, pvec);
    if (!(det != 0.0f)) return;
    const float inv_det = 1.0f / det;
    const f3 tvec = L.o - v0;
    const float u = dot(tvec, pvec) * inv_det;
    if (!(u >= 0.0f && u <= 1.0f)) return;
    const f3 qvec = cross(tvec, e1);
    const float v = dot(L.d, qvec) * inv_det;
    if (!(v >= 0.0f && (u + v) <= 1.0f)) return;
    const float t = dot(e2, qvec) * inv_det;
    if ((t < T_MAX) && (t > t_min) && ((t < best_t) || (t == best_t && prim < best))) { best_t = t; best = prim; }
}

// slab test of one padded box against [t_lo, t_hi]; tn = entry distance
__device__ __forceinline__ bool box_hit(f3 lo, f3 hi, f3 o, f3 inv, float t_lo, float t_hi, float &tn)
{
    const float x0 = (lo.x - o.x) * inv.x, x1 = (hi.x - o.x) * inv.x;
    const float y0 = (lo.y - o.y) * inv.y, y1 = (hi.y - o.y) * inv.y;
    const float z0 = (lo.z - o.z) * inv.z, z1 = (hi.z - o.z) * inv.z;
    const float n = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_lo));
    const float f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_hi));
    tn = n;
    return n <= f;
}

enum { KIND_BRUTE_SPHERES = 0, KIND_BVH_SPHERES = 1, KIND_BVH_TRIANGLES = 2 };

template <int KIND, int STACK, bool HIST>
__global__ __launch_bounds__(BLOCK) void rt_trace_kernel(
    const TraceParams P,
    const float4 *__restrict__ geom, const float4 *__restrict__ tri, const float4 *__restrict__ aux,
    const float4 *__restrict__ albedo, const float4 *__restrict__ bvh,
    float *__restrict__ accum, uint32_t *__restrict__ raycount,
    uint32_t *__restrict__ queue, unsigned long long *__restrict__ stats)
{
    constexpr bool USE_BVH = KIND != KIND_BRUTE_SPHERES;
    __shared__ uint32_t s_hist[20];
    __shared__ uint32_t s_stack[USE_BVH ? STACK * BLOCK : 1];
    __shared__ uint32_t s_cand[USE_BVH ? CAND * BLOCK : 1];
    if (HIST) {
        if (threadIdx.x < 20) s_hist[threadIdx.x] = 0;
        __syncthreads();
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;

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

        // ---- world.hit ---------------------------------------------------------------------------
        const float a = dot(L.d, L.d);
        float best_t = T_MAX;
        uint32_t best = NOHIT;

        if (KIND == KIND_BRUTE_SPHERES) {
            // closest root over all spheres in list order (raytracer.zig:92-113).  Lanes without a
            // ray run the loop on dummy data.
            for (uint32_t s = 0; s < P.n_prim; ++s) {
                const float4 g = geom[s];             // wave-uniform address -> scalar load
                const f3 oc = mk(L.o.x - g.x, L.o.y - g.y, L.o.z - g.z);
                const float b = dot(oc, L.d);
                const float c = dot(oc, oc) - g.w;
                const float disc = b * b - a * c;
                if (disc > 0) {
                    const float sq = sqrtf(disc);
                    float t = (-b - sq) / a;
                    if (!((t < best_t) && (t > P.t_min))) t = (-b + sq) / a;
                    if ((t < best_t) && (t > P.t_min)) { best_t = t; best = s; }
                }
            }
        } else {
            // per-lane BVH traversal; every iteration of the loop is an inner-node visit for every
            // lane that still traverses, leaves are only queued.
            f3 dd = L.d;                              // guard the reciprocal against zero components
            if (fabsf(dd.x) < 1e-30f) dd.x = copysignf(1e-30f, dd.x);
            if (fabsf(dd.y) < 1e-30f) dd.y = copysignf(1e-30f, dd.y);
            if (fabsf(dd.z) < 1e-30f) dd.z = copysignf(1e-30f, dd.z);
            const f3 inv = mk(__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y), __builtin_amdgcn_rcpf(dd.z));
            uint32_t node = has_ray ? 0u : DONE;
            uint32_t sp = 0, ncand = 0;

            auto flush = [&]() {
                for (uint32_t k = 0; __ballot(k < ncand) != 0ull; ++k) {
                    if (k < ncand) {
                        const uint32_t prim = s_cand[k * BLOCK + tid];
                        if (KIND == KIND_BVH_SPHERES) {
                            sphere_exact(geom[prim], prim, L, a, P.t_min, best_t, best);
                        } else {
                            const float4 p0 = tri[3u * prim], p1 = tri[3u * prim + 1u], p2 = tri[3u * prim + 2u];
                            triangle_exact(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), prim, L,
                                           P.t_min, best_t, best);
                        }
                    }
                }
                ncand = 0;
            };

            while (__ballot(node != DONE) != 0ull) {
                if (node != DONE) {
                    const float4 q0 = bvh[4u * node], q1 = bvh[4u * node + 1u], q2 = bvh[4u * node + 2u],
                                 q3 = bvh[4u * node + 3u];
                    float tn0, tn1;
                    const bool h0 = box_hit(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), L.o, inv, P.t_min, best_t, tn0);
                    const bool h1 = box_hit(mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w), L.o, inv, P.t_min, best_t, tn1);
                    const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                    const bool l0 = (c0 & LEAF_BIT) != 0u, l1 = (c1 & LEAF_BIT) != 0u;
                    if (h0 && l0) { s_cand[ncand * BLOCK + tid] = c0 & ~LEAF_BIT; ncand += 1; }
                    if (h1 && l1) { s_cand[ncand * BLOCK + tid] = c1 & ~LEAF_BIT; ncand += 1; }
                    const bool v0 = h0 && !l0, v1 = h1 && !l1;
                    if (v0 && v1) {                   // descend into the nearer child, stack the other
                        const bool first0 = tn0 <= tn1;
                        s_stack[sp * BLOCK + tid] = first0 ? c1 : c0;
                        sp += 1;
                        node = first0 ? c0 : c1;
                    } else if (v0) {
                        node = c0;
                    } else if (v1) {
                        node = c1;
                    } else if (sp > 0) {
                        sp -= 1;
                        node = s_stack[sp * BLOCK + tid];
                    } else {
                        node = DONE;
                    }
                }
                // batch the exact tests: run them for the whole wave when some lane's queue is nearly full
                if (__ballot(ncand > (uint32_t)(CAND - 2)) != 0ull) flush();
            }
            flush();
        }

        if (has_ray) {
            L.nrays += 1;
            if (HIST) atomicAdd(&s_hist[L.depth < 19u ? L.depth : 19u], 1u);
            bool cont = false;
            if (best == NOHIT) {
                shade_miss(L, a);
            } else if (L.depth <= P.max_depth) {      // raytracer.zig:556-558: the cap is tested after the hit
                Surface S;
                S.p = L.o + best_t * L.d;             // ray.zig:10-12
                uint32_t mat;
                if (KIND == KIND_BVH_TRIANGLES) {
                    const float4 p0 = tri[3u * best], p1 = tri[3u * best + 1u], p2 = tri[3u * best + 2u];
                    const f3 v0 = mk(p0.x, p0.y, p0.z);
                    const f3 e1 = mk(p1.x, p1.y, p1.z) - v0, e2 = mk(p2.x, p2.y, p2.z) - v0;
                    const f3 cr = cross(e1, e2);
                    S.n = cr / sqrtf(dot(cr, cr));
                    mat = __float_as_uint(p0.w);
                } else {
                    const float4 g = geom[best];
                    mat = best;
                    const float radius = aux[mat].x;
                    S.n = (S.p - mk(g.x, g.y, g.z)) / radius;      // raytracer.zig:60
                }
                const float4 ax = aux[mat], al = albedo[mat];
                S.kind = __float_as_uint(ax.y); S.param = ax.z; S.albedo = mk(al.x, al.y, al.z);
                cont = shade_hit(P, L, S, a);
            }
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

template <int KIND, int STACK, bool HIST>
hipError_t launch_one(const TraceParams &p, const DeviceScene &s, float *accum_sum, uint32_t *raycount,
                      uint32_t *queue, unsigned long long *stats, int num_cus, hipStream_t stream)
{
    auto kern = rt_trace_kernel<KIND, STACK, HIST>;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (per_cu > 8) per_cu = 8;
    long long want = ((long long)p.n_items + BLOCK - 1) / BLOCK;    // no more blocks than work
    long long grid = (long long)num_cus * per_cu;
    if (grid > want) grid = want;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), 0, stream,
                       p, s.geom, s.tri, s.aux, s.albedo, s.bvh, accum_sum, raycount, queue, stats);
    return hipGetLastError();
}

}  // namespace

const char *rt_trace_kernel_name(TraceKernel which)
{
    switch (which) {
    case RT_K_BRUTE_SPHERES: return "rt_trace_kernel<brute_spheres>";
    case RT_K_BVH_SPHERES: return "rt_trace_kernel<bvh_spheres>";
    default: return "rt_trace_kernel<bvh_triangles>";
    }
}

hipError_t rt_launch_trace(TraceKernel which, const TraceParams &p, const DeviceScene &s, float *accum_sum,
                           uint32_t *raycount, uint32_t *queue, unsigned long long *stats,
                           bool histogram, int num_cus, hipStream_t stream)
{
#define RT_LAUNCH(KIND, STACK)                                                                            \
    (histogram ? launch_one<KIND, STACK, true>(p, s, accum_sum, raycount, queue, stats, num_cus, stream)  \
               : launch_one<KIND, STACK, false>(p, s, accum_sum, raycount, queue, stats, num_cus, stream))
    if (which == RT_K_BRUTE_SPHERES) return RT_LAUNCH(KIND_BRUTE_SPHERES, 1);
    if (s.bvh_depth > 32 || !s.bvh) return hipErrorInvalidValue;
    if (which == RT_K_BVH_SPHERES) return s.bvh_depth <= 16 ? RT_LAUNCH(KIND_BVH_SPHERES, 16) : RT_LAUNCH(KIND_BVH_SPHERES, 32);
    return s.bvh_depth <= 16 ? RT_LAUNCH(KIND_BVH_TRIANGLES, 16) : RT_LAUNCH(KIND_BVH_TRIANGLES, 32);
#undef RT_LAUNCH
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
