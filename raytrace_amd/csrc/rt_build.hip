// Device-side build of the traversal tree (rt_set_tuning build_on_device = 1).  The counterpart in the reference is
// build_BVH, raytracer.zig:279-376 (host, recursive, sorts at every level); as with the host builder (rt_bvh.hpp) only
// the closest-hit answer of the list traversal is the contract, the tree's shape is free.
//
//   1. padded box + centroid of every primitive (the formulas of rtbvh::sphere_prim / triangle_prim), centroid bounds
//   2. 63-bit extended Morton code (Vinkler, Bittner, Havran 2017): the centroid's x, y, z (17 bits each inside the centroid
//      bounds) interleaved with the primitive's SIZE (box diagonal over the bounds' diagonal, 12 bits, one after each of
//      the first 12 xyz triples) -- in a scene that mixes 40-unit ground triangles with 0.05-unit ones, the plain code
//      leaves the big ones as leaves deep inside the small ones' subtrees, every ancestor box grows to their size, and
//      the trace ran 5.6x slower than on the host's SAH tree; with the size bit the big ones split off as soon as the
//      cells reach their size.  Radix sort of (code, index).
//   3. top-down, one launch per level: a node over the sorted range [lo, hi) splits where the highest differing key
//      bit flips (the cell boundary: a binary search), unless a half would then not fit the levels left under the depth
//      cap -- then, and for equal keys, at the middle.  (Splitting every range at its middle instead -- a closed-form,
//      perfectly balanced tree -- ran the trace 5.6x slower than the host's SAH tree: on every level some node
//      straddles each big cell boundary and its box spans both sides.)  A subtree over m primitives has m - 1 nodes,
//      so the node over [lo, hi) at slot s puts its left child at s + 1 and its right child at s + (left size): depth-
//      first slots like the host builder's, no allocation, the same layout on every run.
//   4. bottom-up, one launch per level: child boxes -> the node's plane record and its own box.
#include "rt_kernels.h"
#include <hipcub/hipcub.hpp>
#include <cfloat>

namespace {

constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr int rt_build_max_levels = 32;

struct BBox { float lo[3], hi[3]; };

// order-preserving map float -> uint32 (for atomic min/max)
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float ord2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

// leaf boxes (same arithmetic as rt_bvh.hpp sphere_prim / triangle_prim, strict f32), centroids, centroid bounds.
// bounds[0..2] = min, [3..5] = max as ordered uints; flags[0] |= 1 for a primitive whose box is not finite and ordered.
__global__ __launch_bounds__(256) void prim_boxes_kernel(bool triangles, uint32_t n, const float4 *__restrict__ geom,
                                                         const float4 *__restrict__ aux, const float4 *__restrict__ tri,
                                                         BBox *__restrict__ leafbox, float4 *__restrict__ centroid,
                                                         uint32_t *bounds, uint32_t *flags)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = { 0, 0, 0 };
    bool ok = true;
    if (k < n) {
        BBox b;
        if (triangles) {
            const float4 p0 = tri[3u * k], p1 = tri[3u * k + 1u], p2 = tri[3u * k + 2u];
            const float v0[3] = { p0.x, p0.y, p0.z }, v1[3] = { p1.x, p1.y, p1.z }, v2[3] = { p2.x, p2.y, p2.z };
            float ext = 0;
            for (int a = 0; a < 3; ++a) {
                const float lo = fminf(v0[a], fminf(v1[a], v2[a])), hi = fmaxf(v0[a], fmaxf(v1[a], v2[a]));
                ext = fmaxf(ext, hi - lo);
            }
            for (int a = 0; a < 3; ++a) {
                const float lo = fminf(v0[a], fminf(v1[a], v2[a])), hi = fmaxf(v0[a], fmaxf(v1[a], v2[a]));
                const float pad = 1.25e-3f * ext + 1e-5f * fmaxf(fabsf(lo), fabsf(hi)) + 1e-30f;
                b.lo[a] = lo - pad; b.hi[a] = hi + pad;
                c[a] = (v0[a] + v1[a] + v2[a]) * (1.0f / 3.0f);
                ok = ok && (v0[a] - v0[a] == 0.0f) && (v1[a] - v1[a] == 0.0f) && (v2[a] - v2[a] == 0.0f);   // finite
            }
        } else {
            const float4 g = geom[k];
            const float ctr[3] = { g.x, g.y, g.z };
            const float r = fabsf(aux[k].x);
            for (int a = 0; a < 3; ++a) {
                const float pad = 1.25e-3f * r + 1e-5f * (fabsf(ctr[a]) + r) + 1e-30f;
                b.lo[a] = ctr[a] - r - pad; b.hi[a] = ctr[a] + r + pad; c[a] = ctr[a];
            }
        }
        for (int a = 0; a < 3; ++a) ok = ok && b.lo[a] <= b.hi[a] && (b.hi[a] - b.lo[a]) < FLT_MAX && (c[a] - c[a] == 0.0f);
        leafbox[k] = b;
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        centroid[k] = make_float4(c[0], c[1], c[2], sqrtf(dx * dx + dy * dy + dz * dz));
    }
    if (!ok) atomicOr(flags, 1u);
    // wave-level min/max of the centroids, one atomic per wave and bound
    for (int a = 0; a < 3; ++a) {
        float lo = (k < n && ok) ? c[a] : FLT_MAX, hi = (k < n && ok) ? c[a] : -FLT_MAX;
        for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&bounds[a], f2ord(lo)); atomicMax(&bounds[3 + a], f2ord(hi)); }
    }
}

// 63 key bits, most significant first: 12 groups {x, y, z, s}, then 5 groups {x, y, z}
__device__ __forceinline__ unsigned long long extended_morton(uint32_t x17, uint32_t y17, uint32_t z17, uint32_t s12)
{
    unsigned long long key = 0;
    for (int b = 0; b < 17; ++b) {
        key = (key << 1) | ((x17 >> (16 - b)) & 1u);
        key = (key << 1) | ((y17 >> (16 - b)) & 1u);
        key = (key << 1) | ((z17 >> (16 - b)) & 1u);
        if (b < 12) key = (key << 1) | ((s12 >> (11 - b)) & 1u);
    }
    return key;
}

__global__ __launch_bounds__(256) void morton_kernel(uint32_t n, const float4 *__restrict__ centroid, const uint32_t *bounds,
                                                     unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float4 c = centroid[k];
    const float cc[3] = { c.x, c.y, c.z };
    uint32_t q[3];
    float diag2 = 0.0f;
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(bounds[a]), hi = ord2f(bounds[3 + a]);
        const float ext = hi - lo;
        float t = ext > 0.0f ? (cc[a] - lo) / ext : 0.0f;             // [0, 1]
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const uint32_t v = (uint32_t)(t * 131071.0f);                  // 17 bits
        q[a] = v > 131071u ? 131071u : v;
        diag2 += ext * ext;
    }
    float sz = diag2 > 0.0f ? c.w / sqrtf(diag2) : 0.0f;               // primitive diagonal / diagonal of the centroid bounds
    sz = fminf(fmaxf(sz, 0.0f), 1.0f);
    const uint32_t s12 = (uint32_t)(sz * 4095.0f) > 4095u ? 4095u : (uint32_t)(sz * 4095.0f);
    keys[k] = extended_morton(q[0], q[1], q[2], s12);
    vals[k] = k;
}

// per level: the slots of its nodes, order[level_start[L] .. level_start[L + 1])
struct Levels { uint32_t start[rt_build_max_levels + 2]; uint32_t count; };

__device__ __forceinline__ uint32_t find_split(const unsigned long long *__restrict__ keys, uint32_t lo, uint32_t hi)
{
    // size of the left part of [lo, hi): the keys that share the first key's bits down to the highest bit in which the
    // first and the last key differ; 0 when all keys are equal
    const unsigned long long first = keys[lo], last = keys[hi - 1u];
    if (first == last) return 0u;
    const int d = __clzll((long long)(first ^ last));          // number of leading bits the whole range shares
    uint32_t a = lo, b = hi - 1u;                               // keys[a] is on the left side, keys[b] on the right
    while (b - a > 1u) {
        const uint32_t m = a + (b - a) / 2u;
        if (__clzll((long long)(first ^ keys[m])) > d) a = m; else b = m;
    }
    return b - lo;
}

// top-down step of one level: node ranges -> children (refs into the node records, ranges and slots for the next level)
__global__ __launch_bounds__(256) void split_kernel(uint32_t level, uint32_t depth_cap, Levels *lv, uint32_t *__restrict__ order,
                                                    uint32_t *__restrict__ rng_lo, uint32_t *__restrict__ rng_hi,
                                                    const unsigned long long *__restrict__ keys,
                                                    const uint32_t *__restrict__ sorted, float *__restrict__ nodes)
{
    const uint32_t t = lv->start[level] + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lv->start[level + 1u]) return;
    const uint32_t me = order[t], lo = rng_lo[me], hi = rng_hi[me], size = hi - lo;     // size >= 2
    uint32_t left = find_split(keys, lo, hi);
    // capacity of a child subtree: 2^(levels left below this node)
    const uint32_t below = depth_cap - level - 1u;
    const unsigned long long cap = below >= 32u ? 0xFFFFFFFFull : (1ull << below);
    if (left == 0u || left > cap || size - left > cap) left = size / 2u;
    uint32_t ref[2];
    for (uint32_t c = 0; c < 2u; ++c) {
        const uint32_t clo = c ? lo + left : lo, chi = c ? hi : lo + left;
        if (chi - clo == 1u) {
            ref[c] = LEAF_BIT | sorted[clo];
        } else {
            const uint32_t slot = c ? me + left : me + 1u;
            rng_lo[slot] = clo; rng_hi[slot] = chi;
            order[atomicAdd(&lv->count, 1u)] = slot;
            ref[c] = slot * (uint32_t)RT_NODE_BYTES;
        }
    }
    float *d = nodes + (size_t)me * (RT_NODE_BYTES / 4);
    d[18] = __uint_as_float(ref[0]); d[19] = __uint_as_float(ref[1]);
}
__global__ void close_level_kernel(uint32_t level, Levels *lv) { lv->start[level + 2u] = lv->count; }

// bottom-up step of one level: the children's boxes -> the node's planes and its own box
__global__ __launch_bounds__(256) void fit_kernel(uint32_t level, const Levels *lv, const uint32_t *__restrict__ order,
                                                  const BBox *__restrict__ leafbox, BBox *__restrict__ nodebox,
                                                  float *__restrict__ nodes)
{
    const uint32_t t = lv->start[level] + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lv->start[level + 1u]) return;
    const uint32_t me = order[t];
    float *d = nodes + (size_t)me * (RT_NODE_BYTES / 4);
    BBox b[2];
    for (int c = 0; c < 2; ++c) {
        const uint32_t ref = __float_as_uint(d[18 + c]);
        b[c] = (ref & LEAF_BIT) ? leafbox[ref & ~LEAF_BIT] : nodebox[ref / (uint32_t)RT_NODE_BYTES];
    }
    BBox u;
    for (int a = 0; a < 3; ++a) {
        float *w = d + 6 * a;                                  // {hi0, hi1, lo0, lo1, hi0, hi1}, rt_kernels.h
        w[0] = b[0].hi[a]; w[1] = b[1].hi[a]; w[2] = b[0].lo[a]; w[3] = b[1].lo[a]; w[4] = b[0].hi[a]; w[5] = b[1].hi[a];
        u.lo[a] = fminf(b[0].lo[a], b[1].lo[a]); u.hi[a] = fmaxf(b[0].hi[a], b[1].hi[a]);
    }
    nodebox[me] = u;
}

}  // namespace

// Builds the (n - 1)-node tree for the scene already on the device into `nodes` ((n - 1) * RT_NODE_BYTES bytes).
// n >= 2; depth_cap in [ceil(log2 n), 32] bounds the inner-node levels.  *depth = levels used; *bad != 0: some primitive
// is not finite (the tree is then not to be used).  Allocates its temporaries (about 90 bytes per primitive) and frees
// them before returning; synchronises the stream.
hipError_t rt_device_build_bvh(bool triangles, uint32_t n, const float4 *geom, const float4 *aux, const float4 *tri,
                               float4 *nodes, uint32_t depth_cap, int *depth, uint32_t *bad, hipStream_t stream)
{
    if (n < 2u || !nodes || !depth || !bad || depth_cap > (uint32_t)rt_build_max_levels) return hipErrorInvalidValue;
    while (depth_cap < (uint32_t)rt_build_max_levels && (1ull << depth_cap) < n) depth_cap++;
    if ((1ull << depth_cap) < n) return hipErrorInvalidValue;
    hipError_t e;
    BBox *leafbox = nullptr, *nodebox = nullptr;
    float4 *centroid = nullptr;
    unsigned long long *keys = nullptr, *keys2 = nullptr;
    uint32_t *vals = nullptr, *vals2 = nullptr, *small = nullptr, *order = nullptr, *rng_lo = nullptr, *rng_hi = nullptr;
    Levels *lv = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    auto cleanup = [&]() {
        (void)hipFree(leafbox); (void)hipFree(nodebox); (void)hipFree(centroid); (void)hipFree(keys); (void)hipFree(keys2);
        (void)hipFree(vals); (void)hipFree(vals2); (void)hipFree(small); (void)hipFree(temp); (void)hipFree(order);
        (void)hipFree(rng_lo); (void)hipFree(rng_hi); (void)hipFree(lv);
    };
#define RT_B(call) do { e = (call); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    RT_B(hipMalloc((void **)&leafbox, sizeof(BBox) * (size_t)n));
    RT_B(hipMalloc((void **)&nodebox, sizeof(BBox) * (size_t)(n - 1u)));
    RT_B(hipMalloc((void **)&centroid, sizeof(float4) * (size_t)n));
    RT_B(hipMalloc((void **)&keys, 8 * (size_t)n));
    RT_B(hipMalloc((void **)&keys2, 8 * (size_t)n));
    RT_B(hipMalloc((void **)&vals, 4 * (size_t)n));
    RT_B(hipMalloc((void **)&vals2, 4 * (size_t)n));
    RT_B(hipMalloc((void **)&order, 4 * (size_t)n));
    RT_B(hipMalloc((void **)&rng_lo, 4 * (size_t)n));
    RT_B(hipMalloc((void **)&rng_hi, 4 * (size_t)n));
    RT_B(hipMalloc((void **)&small, 8 * sizeof(uint32_t)));
    RT_B(hipMalloc((void **)&lv, sizeof(Levels)));
    const uint32_t init[8] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u };   // min x3, max x3, flags
    RT_B(hipMemcpyAsync(small, init, sizeof init, hipMemcpyHostToDevice, stream));
    const unsigned blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(prim_boxes_kernel, dim3(blocks), dim3(256), 0, stream, triangles, n, geom, aux, tri, leafbox, centroid,
                       small, small + 6);
    RT_B(hipGetLastError());
    hipLaunchKernelGGL(morton_kernel, dim3(blocks), dim3(256), 0, stream, n, centroid, small, keys, vals);
    RT_B(hipGetLastError());
    hipcub::DoubleBuffer<unsigned long long> dk(keys, keys2);
    hipcub::DoubleBuffer<uint32_t> dv(vals, vals2);
    RT_B(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, dk, dv, (int)n, 0, 63, stream));
    RT_B(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    RT_B(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, dk, dv, (int)n, 0, 63, stream));
    // root: slot 0 over [0, n)
    Levels h;
    memset(&h, 0, sizeof h);
    h.start[0] = 0; h.start[1] = 1; h.count = 1;
    const uint32_t zero = 0;
    RT_B(hipMemcpyAsync(lv, &h, sizeof h, hipMemcpyHostToDevice, stream));
    RT_B(hipMemcpyAsync(order, &zero, 4, hipMemcpyHostToDevice, stream));
    RT_B(hipMemcpyAsync(rng_lo, &zero, 4, hipMemcpyHostToDevice, stream));
    RT_B(hipMemcpyAsync(rng_hi, &n, 4, hipMemcpyHostToDevice, stream));
    for (uint32_t level = 0; level < depth_cap; ++level) {
        // a level holds at most 2^level nodes and at most n / 2 (every node covers two primitives or more)
        const unsigned long long most = level >= 31u ? n : (1ull << level);
        const uint32_t cnt = (uint32_t)(most < n / 2u + 1u ? most : n / 2u + 1u);
        hipLaunchKernelGGL(split_kernel, dim3((cnt + 255u) / 256u), dim3(256), 0, stream, level, depth_cap, lv, order, rng_lo,
                           rng_hi, dk.Current(), dv.Current(), reinterpret_cast<float *>(nodes));
        RT_B(hipGetLastError());
        hipLaunchKernelGGL(close_level_kernel, dim3(1), dim3(1), 0, stream, level, lv);
        RT_B(hipGetLastError());
    }
    for (int level = (int)depth_cap - 1; level >= 0; --level) {
        const unsigned long long most = level >= 31 ? n : (1ull << level);
        const uint32_t cnt = (uint32_t)(most < n / 2u + 1u ? most : n / 2u + 1u);
        hipLaunchKernelGGL(fit_kernel, dim3((cnt + 255u) / 256u), dim3(256), 0, stream, (uint32_t)level, lv, order, leafbox,
                           nodebox, reinterpret_cast<float *>(nodes));
        RT_B(hipGetLastError());
    }
    uint32_t flags = 0;
    RT_B(hipMemcpyAsync(&flags, small + 6, 4, hipMemcpyDeviceToHost, stream));
    RT_B(hipMemcpyAsync(&h, lv, sizeof h, hipMemcpyDeviceToHost, stream));
    RT_B(hipStreamSynchronize(stream));
#undef RT_B
    cleanup();
    *bad = flags;
    int used = 0;
    while (used < (int)depth_cap && h.start[used + 1] > h.start[used]) used++;
    *depth = used;
    // every node placed exactly once (n - 1 slots): anything else means the ranges went wrong
    if (h.count != n - 1u) *bad |= 2u;
    return hipSuccess;
}
