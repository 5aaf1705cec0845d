// CPU check of the GPU BVH: builds trees with raytrace_amd/csrc/rt_bvh.hpp and runs the SAME traversal
// algorithm the kernel runs (two-child nodes, stack, candidate queue with batched flushes, padded-box
// slab test) on the host, comparing its closest hit with a brute-force list for random spheres/rays.
// Exits non-zero on the first disagreement.  Built and run by tests/test_bvh_cpu.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../raytrace_amd/csrc/rt_bvh.hpp"

struct Sph { float c[3], r; };
struct Ray { float o[3], d[3]; };

static float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// Sphere.hit against a running closest hit (same arithmetic as the kernel's sphere_exact)
static void sphere_exact(const Sph &s, uint32_t prim, const Ray &r, float a, float t_min, float &best_t, uint32_t &best)
{
    const float oc[3] = { r.o[0] - s.c[0], r.o[1] - s.c[1], r.o[2] - s.c[2] };
    const float b = dot3(oc, r.d);
    const float c = dot3(oc, oc) - s.r * s.r;
    const float disc = b * b - a * c;
    if (disc > 0) {
        const float sq = std::sqrt(disc);
        float t = (-b - sq) / a;
        if (!(t > t_min)) t = (-b + sq) / a;
        if ((t > t_min) && ((t < best_t) || (t == best_t && prim < best))) { best_t = t; best = prim; }
    }
}

static bool box_hit(const float *lo, const float *hi, const Ray &r, const float *inv, float t_lo, float t_hi, float &tn)
{
    float n = t_lo, f = t_hi;
    for (int k = 0; k < 3; ++k) {
        const float a0 = (lo[k] - r.o[k]) * inv[k], a1 = (hi[k] - r.o[k]) * inv[k];
        n = std::fmax(n, std::fmin(a0, a1));
        f = std::fmin(f, std::fmax(a0, a1));
    }
    tn = n;
    return n <= f;
}

static int g_max_sp = 0, g_max_cand = 0;

static void traverse(const rtbvh::Tree &t, const std::vector<Sph> &sph, const Ray &r, float t_min, int stack_cap,
                     float &best_t, uint32_t &best)
{
    constexpr int CAND = 8;
    const float a = dot3(r.d, r.d);
    float inv[3];
    for (int k = 0; k < 3; ++k) {
        float d = r.d[k];
        if (std::fabs(d) < 1e-30f) d = std::copysign(1e-30f, d);
        inv[k] = 1.0f / d;
    }
    std::vector<uint32_t> stack(stack_cap);
    uint32_t cand[CAND];
    uint32_t node = 0, sp = 0, ncand = 0;
    const uint32_t DONE = 0xFFFFFFFFu;
    auto test_one = [&]() { if (ncand) { --ncand; sphere_exact(sph[cand[ncand]], cand[ncand], r, a, t_min, best_t, best); } };
    auto flush = [&]() { while (ncand) test_one(); };
    while (node != DONE) {
        if (node >= t.nodes.size()) { std::printf("node index out of range\n"); std::exit(2); }
        const float *w = t.nodes[node].w;
        const uint32_t *u = reinterpret_cast<const uint32_t *>(w);
        float tn0, tn1;
        const bool h0 = box_hit(w + 0, w + 3, r, inv, t_min, best_t, tn0);
        const bool h1 = box_hit(w + 6, w + 9, r, inv, t_min, best_t, tn1);
        const uint32_t c0 = u[12], c1 = u[13];
        const bool l0 = c0 & rtbvh::LEAF_BIT, l1 = c1 & rtbvh::LEAF_BIT;
        if (h0 && l0) { if (ncand >= CAND) { std::printf("candidate overflow\n"); std::exit(2); } cand[ncand++] = c0 & ~rtbvh::LEAF_BIT; }
        if (h1 && l1) { if (ncand >= CAND) { std::printf("candidate overflow\n"); std::exit(2); } cand[ncand++] = c1 & ~rtbvh::LEAF_BIT; }
        for (uint32_t k = 0; k < ncand; ++k) if (cand[k] >= sph.size()) { std::printf("primitive index out of range\n"); std::exit(2); }
        const bool v0 = h0 && !l0, v1 = h1 && !l1;
        if (v0 && v1) {
            const bool first0 = tn0 <= tn1;
            if ((int)sp >= stack_cap) { std::printf("stack overflow\n"); std::exit(2); }
            stack[sp++] = first0 ? c1 : c0;
            node = first0 ? c0 : c1;
        } else if (v0) node = c0;
        else if (v1) node = c1;
        else if (sp > 0) node = stack[--sp];
        else node = DONE;
        if ((int)sp > g_max_sp) g_max_sp = (int)sp;
        if ((int)ncand > g_max_cand) g_max_cand = (int)ncand;
        if (ncand >= 2 && (node & 1u)) test_one();          // stands for "enough lanes hold candidates": an arbitrary early batch
        while (ncand > (uint32_t)(CAND - 2)) test_one();
    }
    flush();
}

int main()
{
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    long checked = 0;
    for (int trial = 0; trial < 60; ++trial) {
        const int n = trial < 6 ? trial + 1 : (trial < 40 ? 2 + (int)(U(rng) * 400) : 1000 + (int)(U(rng) * 3000));
        std::vector<Sph> sph(n);
        std::vector<rtbvh::Prim> prims(n);
        for (int i = 0; i < n; ++i) {
            const bool planar = trial % 3 == 0;
            sph[i].c[0] = (U(rng) - 0.5f) * 22; sph[i].c[1] = planar ? 0.2f : (U(rng) - 0.5f) * 22; sph[i].c[2] = (U(rng) - 0.5f) * 22;
            sph[i].r = (trial % 5 == 1 && i % 7 == 0) ? -0.15f : 0.05f + U(rng) * 0.4f;
            if (i == 0 && trial % 2 == 0) { sph[i].c[0] = 0; sph[i].c[1] = -1000; sph[i].c[2] = 0; sph[i].r = 1000; }
            if (trial % 7 == 3 && i > 0) { std::memcpy(sph[i].c, sph[0].c, sizeof sph[i].c); }   // all centres equal: degenerate SAH
            prims[i] = rtbvh::sphere_prim(sph[i].c[0], sph[i].c[1], sph[i].c[2], sph[i].r);
        }
        const int cap = trial % 2 ? 12 : rtbvh::MAX_DEPTH;     // 2^12 >= every n used here: the cap must hold
        const rtbvh::Tree t = rtbvh::build(prims, cap);
        if (t.depth > cap) { std::printf("trial %d: depth %d exceeds the cap %d (n=%d)\n", trial, t.depth, cap, n); return 1; }
        if (!rtbvh::validate(t, prims)) { std::printf("trial %d: validate failed (n=%d)\n", trial, n); return 1; }
        const int stack_cap = t.depth <= 16 ? 16 : 32;
        for (int q = 0; q < 3000; ++q) {
            Ray r;
            const int mode = q % 4;
            for (int k = 0; k < 3; ++k) { r.o[k] = (U(rng) - 0.5f) * 30; r.d[k] = (U(rng) - 0.5f) * 2; }
            if (mode == 1) { const Sph &s = sph[q % n]; for (int k = 0; k < 3; ++k) r.o[k] = s.c[k] + std::fabs(s.r) * (k == 1 ? 1.0f : 0.0f); }   // on a surface
            if (mode == 2) { r.d[q % 3] = 0.0f; }                               // axis-parallel rays (zero component)
            if (mode == 3) { r.o[1] = 2 + U(rng) * 3; r.d[1] = -U(rng) * 0.2f; }  // grazing over the plane
            const float a = dot3(r.d, r.d);
            if (!(a > 0)) continue;
            float bt = 3.402823466e+38f; uint32_t bp = 0xFFFFFFFFu;
            for (int i = 0; i < n; ++i) sphere_exact(sph[i], (uint32_t)i, r, a, 0.001f, bt, bp);
            float gt = 3.402823466e+38f; uint32_t gp = 0xFFFFFFFFu;
            traverse(t, sph, r, 0.001f, stack_cap, gt, gp);
            if (gp != bp || std::memcmp(&gt, &bt, 4) != 0) {
                std::printf("trial %d ray %d: bvh (%u, %.9g) != list (%u, %.9g)\n", trial, q, gp, gt, bp, bt);
                return 1;
            }
            ++checked;
        }
    }
    // large subtrees are built by several threads: the tree must not depend on the thread count
    {
        const int n = 60000;
        std::vector<rtbvh::Prim> prims(n);
        for (int i = 0; i < n; ++i)
            prims[i] = rtbvh::sphere_prim((U(rng) - 0.5f) * 40, (U(rng) - 0.5f) * 4, (U(rng) - 0.5f) * 40, 0.01f + U(rng) * 0.2f);
        const rtbvh::Tree one = rtbvh::build(prims, 24, 1), many = rtbvh::build(prims, 24, 8);
        if (!rtbvh::validate(one, prims) || !rtbvh::validate(many, prims)) { std::printf("threaded build: validate failed\n"); return 1; }
        if (one.depth != many.depth || one.nodes.size() != many.nodes.size() ||
            std::memcmp(one.nodes.data(), many.nodes.data(), one.nodes.size() * sizeof(rtbvh::Node)) != 0) {
            std::printf("threaded build differs from the single-threaded one\n");
            return 1;
        }
    }
    std::printf("ok: %ld rays agree; max stack %d, max queued candidates %d\n", checked, g_max_sp, g_max_cand);
    return 0;
}
