// Host-side BVH builder for the GPU traversal kernels (binned SAH, one primitive per leaf).
//
// This is NOT the reference's build_BVH (raytracer.zig:279-376; depth 5, 2-axis slab test, quirks
// Q1/Q2 in SURVEY 8(a)).  The contract with the reference is only the closest-hit answer of the
// list traversal (raytracer.zig:92-113): the GPU BVH culls conservatively (boxes are padded) and the
// primitive tests on the survivors are the exact ones, so the answer is the list's.
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <future>
#include <thread>
#include <vector>

namespace rtbvh {

struct Box {
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    float hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float p[3]) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.0f : dx * dy + dy * dz + dz * dx;
    }
};

struct Prim { Box box; float c[3]; };

// 64-byte node: two child boxes + two child references.
//   q0 = { lo0.x, lo0.y, lo0.z, hi0.x }   q1 = { hi0.y, hi0.z, lo1.x, lo1.y }
//   q2 = { lo1.z, hi1.x, hi1.y, hi1.z }   q3 = { bits(child0), bits(child1), 0, 0 }
// child reference: bit 31 set = leaf holding primitive (ref & 0x7fffffff); else inner node index.
// Every node has two valid children (a one-primitive scene lists its leaf twice; there is no "empty"
// reference, because an inverted box is NOT unhittable for a min/max slab test).
struct Node { float w[16]; };
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr int MAX_DEPTH = 32;          // hard limit of the traversal stacks the kernels are built with

struct Tree {
    std::vector<Node> nodes;
    int depth = 0;          // number of inner-node levels (stack entries a traversal can need)
};

namespace detail {

inline void set_child(Node &n, int which, const Box &b, uint32_t ref)
{
    float *w = n.w;
    if (which == 0) { w[0] = b.lo[0]; w[1] = b.lo[1]; w[2] = b.lo[2]; w[3] = b.hi[0]; w[4] = b.hi[1]; w[5] = b.hi[2]; }
    else { w[6] = b.lo[0]; w[7] = b.lo[1]; w[8] = b.lo[2]; w[9] = b.hi[0]; w[10] = b.hi[1]; w[11] = b.hi[2]; }
    uint32_t *u = reinterpret_cast<uint32_t *>(w);
    u[12 + which] = ref;
}

struct Builder {
    const std::vector<Prim> &prims;
    std::vector<uint32_t> idx;
    Tree tree;
    int depth_cap = MAX_DEPTH;   // SAH splits that would make the tree deeper than this fall back to the median
    std::atomic<int> depth_seen{ 0 };
    std::atomic<int> spare_threads{ 0 };       // subtrees that may still be handed to a thread of their own
    explicit Builder(const std::vector<Prim> &p) : prims(p), idx(p.size()) { for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i; }

    // Returns the reference of the subtree over idx[first, first+count) and its box.  A subtree over `count`
    // primitives has exactly count - 1 inner nodes, so its nodes get the fixed range [me, me + count - 1) of the
    // (pre-sized) node array in depth-first order: the tree does not depend on which subtrees ran in parallel.
    uint32_t build(uint32_t first, uint32_t count, int depth, uint32_t me, Box &out_box)
    {
        out_box = Box();
        for (uint32_t i = 0; i < count; ++i) out_box.grow(prims[idx[first + i]].box);
        if (count == 1) return LEAF_BIT | idx[first];

        Box cb;
        for (uint32_t i = 0; i < count; ++i) cb.grow(prims[idx[first + i]].c);
        // binned SAH over the three axes; 32 bins for large scenes (C4: the trace runs 3 % faster on that tree than on the
        // 16-bin one, 11 % faster than on the 8-bin one), 16 for the small ones, where it makes no difference
        constexpr int NBMAX = 32;
        const int NB = prims.size() >= 4096 ? 32 : 16;
        float best_cost = FLT_MAX; int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3; ++ax) {
            const float ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0)) continue;
            Box bb[NBMAX]; uint32_t bc[NBMAX] = { 0 };
            const float scale = NB / ext;
            for (uint32_t i = 0; i < count; ++i) {
                const Prim &p = prims[idx[first + i]];
                int b = (int)((p.c[ax] - cb.lo[ax]) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                bb[b].grow(p.box); bc[b]++;
            }
            float right_area[NBMAX]; uint32_t right_cnt[NBMAX];
            Box acc; uint32_t cnt = 0;
            for (int b = NB - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.half_area(); right_cnt[b] = cnt; }
            acc = Box(); cnt = 0;
            for (int b = 0; b < NB - 1; ++b) {
                acc.grow(bb[b]); cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                const float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
            }
        }
        uint32_t mid = 0;
        const uint64_t cap = depth + 1 >= depth_cap ? 1ull : (1ull << (depth_cap - depth - 1));   // per-child capacity
        if (best_axis >= 0) {
            const float ext = cb.hi[best_axis] - cb.lo[best_axis];
            const float scale = NB / ext;
            auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t pi) {
                int b = (int)((prims[pi].c[best_axis] - cb.lo[best_axis]) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                return b <= best_bin;
            });
            mid = (uint32_t)(it - (idx.begin() + first));
        }
        if (mid == 0 || mid == count || mid > cap || count - mid > cap) {
            // degenerate or too deep: median split along the widest centroid axis
            int ax = 0;
            if (cb.hi[1] - cb.lo[1] > cb.hi[ax] - cb.lo[ax]) ax = 1;
            if (cb.hi[2] - cb.lo[2] > cb.hi[ax] - cb.lo[ax]) ax = 2;
            mid = count / 2;
            std::nth_element(idx.begin() + first, idx.begin() + first + mid, idx.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return prims[a].c[ax] < prims[b].c[ax]; });
        }
        for (int seen = depth_seen.load(); seen < depth + 1 && !depth_seen.compare_exchange_weak(seen, depth + 1); ) { }
        Box b0, b1;
        uint32_t r0, r1;
        // the two halves touch disjoint ranges of idx and of the node array
        if (count >= 8192 && spare_threads.fetch_sub(1) > 0) {
            auto left = std::async(std::launch::async, [&]() { return build(first, mid, depth + 1, me + 1, b0); });
            r1 = build(first + mid, count - mid, depth + 1, me + mid, b1);
            r0 = left.get();
            spare_threads.fetch_add(1);
        } else {
            if (count >= 8192) spare_threads.fetch_add(1);
            r0 = build(first, mid, depth + 1, me + 1, b0);
            r1 = build(first + mid, count - mid, depth + 1, me + mid, b1);
        }
        set_child(tree.nodes[me], 0, b0, r0);
        set_child(tree.nodes[me], 1, b1, r1);
        return me;
    }
};

}  // namespace detail

// Build over `prims` (boxes must already be padded).  nodes[0] is the root.  depth_cap (<= MAX_DEPTH) bounds
// the number of inner-node levels: a shallower tree needs a smaller per-lane stack in LDS, which buys
// occupancy for the latency-bound large-scene kernel.  Needs 2^depth_cap >= prims.size().
// Large subtrees are built by up to `threads` threads (0 = the host's hardware threads, at most 16); the result
// is the same tree whatever the thread count.
inline Tree build(const std::vector<Prim> &prims, int depth_cap = MAX_DEPTH, int threads = 0)
{
    detail::Builder b(prims);
    b.depth_cap = depth_cap < 1 ? 1 : (depth_cap > MAX_DEPTH ? MAX_DEPTH : depth_cap);
    while (b.depth_cap < MAX_DEPTH && (1ull << b.depth_cap) < prims.size()) b.depth_cap++;
    if (prims.empty()) return b.tree;
    if (prims.size() == 1) {
        b.tree.nodes.emplace_back();
        detail::set_child(b.tree.nodes[0], 0, prims[0].box, LEAF_BIT | 0u);
        detail::set_child(b.tree.nodes[0], 1, prims[0].box, LEAF_BIT | 0u);
        b.tree.depth = 1;
        return b.tree;
    }
    Box root;
    b.tree.nodes.resize(prims.size() - 1);
    const unsigned hw = std::thread::hardware_concurrency();
    b.spare_threads = threads > 0 ? threads - 1 : (int)std::min(hw ? hw : 1u, 16u) - 1;
    b.build(0, (uint32_t)prims.size(), 0, 0u, root);
    b.tree.depth = b.depth_seen.load();
    return std::move(b.tree);
}

// Structural check run at every upload (O(n)): every reference in range, every primitive in exactly
// one leaf (twice for the one-primitive tree), every child box containing its subtree's primitives,
// depth within the traversal stack.  A tree that fails is never sent to the GPU.
inline bool validate(const Tree &t, const std::vector<Prim> &prims)
{
    const size_t n = prims.size();
    if (n == 0) return t.nodes.empty();
    if (t.nodes.size() != (n == 1 ? 1 : n - 1) || t.depth < 1 || t.depth > MAX_DEPTH) return false;
    std::vector<uint32_t> seen(n, 0), node_seen(t.nodes.size(), 0);
    struct Item { uint32_t node; int depth; };
    std::vector<Item> stack{ { 0u, 1 } };
    node_seen[0] = 1;
    // subtree boxes are checked against the primitives' own boxes at the leaves and, for inner children,
    // against the union of the grandchildren boxes (which is how the builder made them)
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        if (it.depth > t.depth) return false;
        const Node &nd = t.nodes[it.node];
        const uint32_t *u = reinterpret_cast<const uint32_t *>(nd.w);
        for (int c = 0; c < 2; ++c) {
            const float *b = nd.w + 6 * c;
            const uint32_t ref = u[12 + c];
            for (int k = 0; k < 3; ++k) if (!(b[k] <= b[3 + k])) return false;
            if (ref & LEAF_BIT) {
                const uint32_t pi = ref & ~LEAF_BIT;
                if (pi >= n) return false;
                seen[pi]++;
                for (int k = 0; k < 3; ++k)
                    if (b[k] > prims[pi].box.lo[k] || b[3 + k] < prims[pi].box.hi[k]) return false;
            } else {
                if (ref >= t.nodes.size() || node_seen[ref]) return false;
                node_seen[ref] = 1;
                const Node &ch = t.nodes[ref];
                for (int k = 0; k < 3; ++k) {
                    const float lo = std::min(ch.w[k], ch.w[6 + k]), hi = std::max(ch.w[3 + k], ch.w[9 + k]);
                    if (b[k] > lo || b[3 + k] < hi) return false;
                }
                stack.push_back({ ref, it.depth + 1 });
            }
        }
    }
    for (size_t i = 0; i < n; ++i) if (seen[i] != (n == 1 ? 2u : 1u)) return false;
    for (uint32_t v : node_seen) if (!v) return false;
    return true;
}

// Conservative padding: the box test may never cull a primitive its exact test would accept.  The static part
// is here -- 1.25e-3 of the size (covers ray origins on or inside the primitive: the exact tests' own slop there is
// a few ulps of the size, and the cone below needs 2^-10 of the size of slack at the near plane) plus 1e-5 of the
// coordinate magnitude (the cancellation error of the slab arithmetic, 2^-24 |o_k|, when plane and origin are
// close).  The part that grows with the distance of the ray origin is applied per ray by the kernels, which test
// a cone instead of a line (rt_kernels.hip, RT_CONE_*).
inline Prim sphere_prim(float cx, float cy, float cz, float radius)
{
    Prim p;
    const float c[3] = { cx, cy, cz };
    const float r = std::fabs(radius);
    for (int k = 0; k < 3; ++k) {
        const float pad = 1.25e-3f * r + 1e-5f * (std::fabs(c[k]) + r) + 1e-30f;
        p.box.lo[k] = c[k] - r - pad; p.box.hi[k] = c[k] + r + pad; p.c[k] = c[k];
    }
    return p;
}
inline Prim triangle_prim(const float v0[3], const float v1[3], const float v2[3])
{
    Prim p;
    float ext = 0;
    for (int k = 0; k < 3; ++k) {
        const float lo = std::min(v0[k], std::min(v1[k], v2[k])), hi = std::max(v0[k], std::max(v1[k], v2[k]));
        ext = std::max(ext, hi - lo);
    }
    for (int k = 0; k < 3; ++k) {
        const float lo = std::min(v0[k], std::min(v1[k], v2[k])), hi = std::max(v0[k], std::max(v1[k], v2[k]));
        const float pad = 1.25e-3f * ext + 1e-5f * std::max(std::fabs(lo), std::fabs(hi)) + 1e-30f;
        p.box.lo[k] = lo - pad; p.box.hi[k] = hi + pad;
        p.c[k] = (v0[k] + v1[k] + v2[k]) * (1.0f / 3.0f);
    }
    return p;
}

}  // namespace rtbvh
