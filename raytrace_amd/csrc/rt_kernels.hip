// Trace kernels for gfx950 (MI355X): Render / trace / color of the reference
// (raytracer.zig:517-572) as one persistent wave64 kernel, in three intersection flavours.
//
// Shape of the kernel
//   * persistent workers: the grid is sized to fill the chip once; every lane owns one work item at a
//     time -- a pixel, or one chunk of a pixel's global sample indices (rt_config.spp_chunk) -- and walks
//     its samples in order, so every sum has a fixed order.  Items come from a global queue in
//     wave-reserved ranges (one atomic per 64-128 items); chunk sums land in per-chunk planes that a
//     second kernel adds in sample order.
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
//     lexicographic minimum of (t, primitive index).  Traversal and shading are decoupled: finished
//     lanes are shaded and restarted at the root while the stragglers keep traversing.  Small scenes
//     are staged in LDS (boxes interleaved for packed FMAs), large ones are read through L1/L2.
//   * no MFMA: this is branchy f32 VALU work.
//
// Arithmetic contract: see rt_device.h.
#include "rt_kernels.h"
#include "rt_device.h"

#include <type_traits>

using namespace rtd;

namespace {

constexpr uint32_t NOPIX = 0xFFFFFFFFu;
constexpr uint32_t NOHIT = 0xFFFFFFFFu;
// the wave's lane mask of a predicate, taken straight from the compare's SGPR pair (HIP's __ballot goes through an integer
// and costs a v_cndmask + v_cmp per use)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
constexpr uint32_t DONE = 0xFFFFFFFFu;
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr float    T_MAX = 3.402823466e+38f;   // std.math.floatMax(f32), raytracer.zig:555
constexpr int      BLOCK = 256;

enum : uint32_t { K_LAMBERTIAN = 0, K_METAL = 1, K_DIELECTRIC = 2 };

// profiling builds: count one wave-level execution and the active lanes of a block (LDS counters behind the depth
// histogram); `prof` is null in production builds and the call folds away
__device__ __forceinline__ void prof_block(uint32_t *prof, int block)
{
    if (prof) {
        const unsigned long long m = wave_ballot(1);
        if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(&prof[2 * block], 1u);
        atomicAdd(&prof[2 * block + 1], 1u);
    }
}

struct Lane {
    f3       o, d;        // current ray (direction never normalised, as in the reference)
    f3       thr;         // product of attenuations so far
    f3       col;         // running sum over this pixel's samples (trace: raytracer.zig:535-541)
    uint32_t pix;         // output index jr*nx+i
    uint32_t i, j;        // pixel column and the reference's j (raytracer.zig:523)
    uint32_t sample;      // global sample index
    uint32_t send;        // end of this lane's sample range (the whole range, or one chunk of it)
    uint32_t part;        // chunk slot (index of the partial-sum plane) when samples are chunked
    uint32_t depth;
    uint32_t nrays;       // world.hit evaluations of this pixel
    // the lane's NEXT work item, already dequeued and decoded (NOPIX = none); see advance()
    uint32_t nx_pix, nx_i, nx_j, nx_part;
    bool     dry;         // the queue handed this lane an item beyond the end: it will get no more
};

struct Surface {          // HitRecord (raytracer.zig:31-36) with the material flattened
    f3       p, n;
    uint32_t kind;
    float    param;
    f3       albedo;
    float    inv_param, r0sq;   // dielectric: 1/index and schlick's r0*r0, made once at upload (same IEEE operations)
};

// trace's per-sample prologue: raytracer.zig:537-539 and camera.zig:16/:39.  r = Philox block 0 of the sample.
__device__ __forceinline__ void camera_ray(const TraceParams &P, Lane &L, const uint32_t r[4])
{
    // numerators are 0 or >= 2^-24 and at most ~2^24: div_recip's range conditions hold by construction
    const float un = (float)L.i + u32_to_unit_float(r[0]), vn = (float)L.j + u32_to_unit_float(r[1]);
    const float u = P.cam_fast ? div_recip(un, (float)P.nx, P.inv_nx) : un / (float)P.nx;
    const float v = P.cam_fast ? div_recip(vn, (float)P.ny, P.inv_ny) : vn / (float)P.ny;
    f3 dir = mk(P.ll[0], P.ll[1], P.ll[2]) + u * mk(P.hz[0], P.hz[1], P.hz[2]) + v * mk(P.vt[0], P.vt[1], P.vt[2]);
    const f3 org = mk(P.org[0], P.org[1], P.org[2]);
    if (P.sub_origin) dir = dir - org;
    L.o = org; L.d = dir;
    L.thr = mk(1.0f, 1.0f, 1.0f);
    L.depth = 0;
}

// HitRecord.normal = (p - center) / radius (raytracer.zig:60), three correctly rounded quotients.  ax = {radius,
// kind, param, RN(1/radius) or NaN}: with the host-made reciprocal each quotient is a multiply and two FMAs
// (div_recip) instead of the ~11-instruction IEEE divide.  |pc| ~ |radius| at a hit point, so nothing overflows;
// zero or tiny components (< 2^-100) and spheres without a usable reciprocal take the IEEE divide, wave-wide.
__device__ __forceinline__ f3 sphere_normal(f3 pc, float4 ax, uint32_t *prof = nullptr)
{
    f3 n = mk(div_recip(pc.x, ax.x, ax.w), div_recip(pc.y, ax.x, ax.w), div_recip(pc.z, ax.x, ax.w));
    const float m = fminf(fminf(__builtin_fabsf(pc.x), __builtin_fabsf(pc.y)), __builtin_fabsf(pc.z));
    const bool slow = !(m >= 0x1.0p-100f) || !(ax.w == ax.w);
    if (wave_ballot(slow) != 0ull) { prof_block(prof, RT_PB_NORMAL_SLOW); n = pc / ax.x; }
    return n;
}

// Material.scatter (raytracer.zig:407-414) for one lane.  unit = normalize(direction) (unused for
// lambertian), r = Philox block depth+1.  Returns true when the path continues with the scattered ray
// in L.o/L.d.
__device__ __forceinline__ bool scatter_ray(Lane &L, const Surface &S, f3 unit, float len_d, const uint32_t r[4], uint32_t *prof)
{
    const f3 p = S.p, n = S.n;
    f3 dir;
    f3 reflected = mk(0.0f, 0.0f, 0.0f);
    if (S.kind != K_LAMBERTIAN) {                                  // reflect, :420-422 / :450-452
        const float dn_u = dot(unit, n);
        reflected = unit - (2.0f * dn_u) * n;
    }
    if (S.kind != K_DIELECTRIC) {
        prof_block(prof, RT_PB_RIUS);
        // random_in_unit_sphere, raytracer.zig:574-578 -- once for the lambertian and the metal lanes together
        const f3 cube = mk(u32_to_unit_float(r[0]), u32_to_unit_float(r[1]), u32_to_unit_float(r[2]));
        const f3 cn = cube / sqrtf(dot(cube, cube));
        const float cs = u32_to_unit_float(r[3]);
        bool risky;
        float cb = cbrt_unit_fast(cs, risky);                      // bit-identical to cbrt_f32 on the unit grid
        if (wave_ballot(risky) != 0ull) { prof_block(prof, RT_PB_CBRT_SLOW); cb = cbrt_f32(cs); }   // 1 input in 4096: musl's f64 steps decide
        const f3 rius = mk((2.0f * cn.x - 1.0f) * cb, (2.0f * cn.y - 1.0f) * cb, (2.0f * cn.z - 1.0f) * cb);
        if (S.kind == K_LAMBERTIAN) {                              // Lambertian.scatter, :438-445
            prof_block(prof, RT_PB_LAMBERTIAN);
            const f3 target = (p + n) + rius;
            dir = target - p;
        } else {                                                   // Metal.scatter, :423-433
            prof_block(prof, RT_PB_METAL);
            dir = reflected + S.param * rius;
            if (!(dot(dir, n) > 0)) return false;                  // absorbed: the path returns black
        }
    } else {                                                       // Dielectric.scatter, :469-499
        prof_block(prof, RT_PB_DIELECTRIC);
        const float param = S.param;
        const float ddn = dot(L.d, n);
        f3 on; float ni_over_nt, cos_num;
        if (ddn > 0) { on = -n; ni_over_nt = param; cos_num = param * ddn; }
        else { on = n; ni_over_nt = S.inv_param; cos_num = -ddn; }      // 1.0 / index, :482
        const float cosine = cos_num / len_d;                              // one divide for both branches, :479/:487
        const float dt = dot(unit, on);                            // refract, :453-463
        const float disc = 1.0f - (ni_over_nt * ni_over_nt) * (1.0f - (dt * dt));
        float reflect_prob = 1.0f;
        f3 refracted = mk(0.0f, 0.0f, 0.0f);
        if (disc > 0) {
            const float sq = sqrtf(disc);
            const f3 w = unit - dt * on;                           // uv - n*splat(dt)
            refracted = mk(ni_over_nt * w.x - on.x * sq, ni_over_nt * w.y - on.y * sq, ni_over_nt * w.z - on.z * sq);
            const float r0 = S.r0sq;                               // schlick, :464-468: ((1 - index) / (1 + index))^2
            reflect_prob = r0 + (1.0f - r0) * pow5(1.0f - cosine);
        }
        dir = (u32_to_unit_float(r[0]) < reflect_prob) ? reflected : refracted;   // :492-496
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
// as the CPU checker.  v0 is the triangle's LARGEST corner: rt_upload_scene rotates the vertices so that v0 faces the
// longest edge (oracle: orc_canonical_triangle), which makes the grazing guard below a property of the triangle's
// shape and not of the order its vertices were listed in.
__device__ __forceinline__ void triangle_exact(f3 v0, f3 v1, f3 v2, uint32_t prim, const Lane &L, float t_min,
                                               float &best_t, uint32_t &best)
{
    const f3 e1 = v1 - v0, e2 = v2 - v0;
    const f3 pvec = cross(L.d, e2);
    const float det = dot(e1, pvec);
    if (!(det != 0.0f)) return;
    // grazing guard, part of the test's definition (oracle: orc_triangle_hit): sin(phi) cos(theta) >= 2^-7 bounds the
    // amplification of the barycentrics' rounding error, so what is accepted lies within 7 * 2^-17 |o - v0| of the
    // triangle -- inside the cone the slab test uses (RT_CONE_TRIANGLES)
    if (!(det * det >= ((dot(L.d, L.d) * dot(e1, e1)) * dot(e2, e2)) * 0x1.0p-14f)) return;
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

// Slab test of both (padded) child boxes of a node in FMA form: plane distance = plane*inv - o*inv (noi = -(o*inv)).
// It only CULLS, and it must never cull a primitive whose exact test would accept the ray -- for ANY ray origin.
// Two things can make the exact tests accept a ray that misses the primitive's true shape:
//   * the f32 discriminant of Sphere.hit, b*b - a*c, carries an absolute error of up to ~16 u |oc|^2 |d|^2
//     (u = 2^-24, oc = origin - center): the test accepts rays that pass within sqrt(r^2 + 16 u |oc|^2) of the
//     centre, i.e. up to 2^-10 |oc| outside the sphere when the origin is far away (at |oc| = 10^4 r that is ten
//     radii: the reference's own arithmetic draws a distant sphere as a blob).  Moller-Trumbore's barycentrics
//     are off by a few u |o - v0| across the ray;
//   * the slab arithmetic itself rounds: relative 2^-23 of the distance, plus 2^-24 |o_k| where plane - o cancels.
// So the ray is widened into a CONE of half-angle RT_CONE (per primitive kind) before the slab test: near planes
// are tested with inv_near = 1 / (|d_k| + g), far planes with inv_far = 1 / (|d_k| - g), g = RT_CONE * |d| -- the
// box grows by RT_CONE * distance, which covers both items above at every distance.  An axis that lies within the
// cone's own width of perpendicular to the ray (|d_k| <= g) has no far bound but a second near bound (ray_splat);
// that term costs five more instructions per step and runs only while some lane of the wave holds such a ray (about
// 1 ray in 300 for spheres, 1 in 20 000 for triangles).  The static padding of the boxes (rt_bvh.hpp: 1.25e-3 of the
// size, 1e-5 of the coordinate) covers origins on or inside the primitive and the cancellation term.  Otherwise no
// extra instruction per step: only the operands of the six packed FMAs differ between near and far.
// What makes a step cheap:
// packed FMAs (v_pk_fma_f32: the two boxes' planes of one axis in one issue slot), and no min/max per axis: the
// device node (80 bytes, rt_kernels.h) stores every axis as {hi0, hi1, lo0, lo1, hi0, hi1}, and a lane reads its 16 bytes at +8 when its ray runs
// up that axis and at +0 when it runs down, so the first pair it gets is always the NEAR planes of the two
// boxes and the second the FAR planes (the sign of inv decides which of lo/hi is nearer, and rounding is
// monotonic, so the selected values are exactly the min and max the generic test would compute).
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float RT_CONE_SPHERES = 0x1.08p-10f;      // sqrt(15.5 u) = 0.97 * 2^-10, plus 6 % for the slab rounding, the centre's rounding and second-order terms
// Moller-Trumbore: u and v are off by ~7 u |o - v0| / (|e| sin(phi) cos(theta)) (phi the triangle's corner angle, theta the
// angle between ray and normal), i.e. ~7 u |o - v0| / (sin(phi) cos(theta)) in space.  triangle_exact's grazing guard keeps
// sin(phi) cos(theta) >= 2^-7, so an accepted ray passes within 7 * 2^-17 |o - v0| of the triangle: 2^-13 covers it 2.3x.
// (Round 2's first cone, 2^-14 without the guard, was beyond reach for rays within a third of a degree of a triangle's
// plane: one case in 8 000 of the far-camera fuzz differed from the checker's plain loop in two pixels.)
constexpr float RT_CONE_TRIANGLES = 0x1.0p-13f;
struct RaySplat {
    v2f      nx, ny, nz;   // near planes of axis k: {inv_near.k, -(o.k * inv_near.k)}
    v2f      fx, fy, fz;   // far planes: {inv_far.k, -(o.k * inv_far.k)}; {0, T_MAX} on an axis that lies inside the cone's width of
                           // perpendicular to the ray (|d_k| <= g): the cone then never leaves the slab, there is no far bound ...
    v2f      ex, ey, ez;   // ... but a second LOWER bound, from the far plane: the cone has to open up to the slab first.  {inv, noi}
                           // on such an axis, {0, -T_MAX} otherwise.  Evaluated only while some lane of the wave has such an axis.
    uint32_t ox, oy, oz;   // byte offset of the lane's {near pair, far pair} of each axis inside a node
    bool     perp;         // this ray has an axis of the second kind
};
// The ray operands of the slab test for one ray (see above).  a = dot(d, d).  Cone condition on axis k at parameter t:
// o_k + t (d_k - g) <= hi_k and o_k + t (d_k + g) >= lo_k (the cone's lower and upper edge against the slab).  With
// s = sign(d_k) and |d_k| > g this is the usual pair "t >= (near - o) / (s (|d_k| + g)), t <= (far - o) / (s (|d_k| - g))";
// with |d_k| <= g the second one turns into a lower bound as well, t >= (far - o) / (s (|d_k| - g)) (denominator < 0).
// Common case, x = g / |d_k| < 1/2 on every axis: one reciprocal per axis, inv = 1 / d_k, and the two bounds
// 1 / (|d_k| + g) >= (1 - x) |inv| and 1 / (|d_k| - g) <= (1 + 2 x) |inv| (looser by at most the cone's own width).
// Otherwise (1 ray in 150 for spheres) the lane's steep axes get their exact reciprocals.
__device__ __forceinline__ void ray_splat(RaySplat &R, f3 o, f3 d, float a, float cone)
{
    // g >= cone * |d|: the approximate sqrt is within 1 ulp, the cone constants carry the margin
    const float g = cone * __builtin_amdgcn_sqrtf(a);
    const float dk[3] = { d.x, d.y, d.z }, ok[3] = { o.x, o.y, o.z };
    v2f n[3], f[3];
    float xmax = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float inv = __builtin_amdgcn_rcpf(dk[k]);
        const float x = g * __builtin_fabsf(inv);
        const float rn = __builtin_fmaf(-x, inv, inv), rf = __builtin_fmaf(x + x, inv, inv);
        n[k] = (v2f){ rn, -(ok[k] * rn) };
        f[k] = (v2f){ rf, -(ok[k] * rf) };
        xmax = fmaxf(xmax, x);
    }
    R.ex = R.ey = R.ez = (v2f){ 0.0f, -T_MAX };
    bool perp = false;
    const bool steep = !(xmax < 0.5f);                  // also NaN and infinity (a zero direction component)
    if (wave_ballot(steep) != 0ull) {
        if (steep) {
            v2f e[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t sgn = __float_as_uint(dk[k]) & 0x80000000u;
                const float ad = __builtin_fabsf(dk[k]);
                const float df = ad - g;
                const bool bounded = df > 1e-30f;
                const float rn = __uint_as_float(__float_as_uint(__builtin_amdgcn_rcpf(fmaxf(ad + g, 1e-30f))) ^ sgn);
                const float rf = __uint_as_float(__float_as_uint(__builtin_amdgcn_rcpf(bounded ? df : fminf(df, -1e-30f))) ^ sgn);
                n[k] = (v2f){ rn, -(ok[k] * rn) };
                f[k] = bounded ? (v2f){ rf, -(ok[k] * rf) } : (v2f){ 0.0f, T_MAX };
                e[k] = bounded ? (v2f){ 0.0f, -T_MAX } : (v2f){ rf, -(ok[k] * rf) };
                perp = perp || !bounded;
            }
            R.ex = e[0]; R.ey = e[1]; R.ez = e[2];
        }
    }
    R.nx = n[0]; R.ny = n[1]; R.nz = n[2];
    R.fx = f[0]; R.fy = f[1]; R.fz = f[2];
    R.perp = perp;
    // running up an axis (sign bit clear): near = lo, the pair at +8; running down: near = hi, at +0
    R.ox = ((__float_as_uint(d.x) >> 31) ^ 1u) << 3;
    R.oy = 24u + (((__float_as_uint(d.y) >> 31) ^ 1u) << 3);
    R.oz = 48u + (((__float_as_uint(d.z) >> 31) ^ 1u) << 3);
}
constexpr uint32_t NODE_AXIS_BYTES = 24, NODE_REFS_OFFSET = 72;   // layout of RT_NODE_BYTES = 80 (rt_kernels.h)
struct __attribute__((aligned(8))) PlanePairs { v2f near_, far_; };   // 16 bytes at 8-byte alignment: one ds_read2_b64 / global_load_dwordx4
// The whole test is ONE asm statement: hipcc pads every boundary between an asm statement and the next instruction that
// touches its outputs with an s_nop (it cannot see that nothing in here is a transcendental), which as separate
// one-instruction statements cost eight issue slots per step.  v_pk_fma_f32 with op_sel broadcasts the halves of a
// ray operand pair, {p0, p1} * op.x + op.y, so an axis costs one VGPR pair per operand set; plain hardware min/max
// (fminf/fmaxf would first re-canonicalise the halves taken out of the packed results); a NaN (inf - inf, 0 * inf
// at the edge of the range) is dropped by them, which only loosens the test.
#define RT_PKFMA(dst, planes, op) "v_pk_fma_f32 " dst ", " planes ", " op ", " op " op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
template <bool PERP>
__device__ __forceinline__ void box2_hit_fma(PlanePairs px, PlanePairs py, PlanePairs pz, const RaySplat &R, float t_lo, float t_hi,
                                             float &n0, float &n1, float &f0, float &f1)
{
    // two statements: the halves of the packed results can only be named through the compiler (sub-registers, no copies)
    v2f xe = (v2f){ 0, 0 }, ye = (v2f){ 0, 0 }, ze = (v2f){ 0, 0 };
    if (PERP)
        asm(RT_PKFMA("%[xe]", "%[xf]", "%[rex]") RT_PKFMA("%[ye]", "%[yf]", "%[rey]") RT_PKFMA("%[ze]", "%[zf]", "%[rez]")
            RT_PKFMA("%[xn]", "%[xn]", "%[rnx]") RT_PKFMA("%[yn]", "%[yn]", "%[rny]") RT_PKFMA("%[zn]", "%[zn]", "%[rnz]")
            RT_PKFMA("%[xf]", "%[xf]", "%[rfx]") RT_PKFMA("%[yf]", "%[yf]", "%[rfy]") "v_pk_fma_f32 %[zf], %[zf], %[rfz], %[rfz] op_sel:[0,0,1] op_sel_hi:[1,0,1]"
            : [xe] "=&v"(xe), [ye] "=&v"(ye), [ze] "=&v"(ze),
              [xn] "+v"(px.near_), [yn] "+v"(py.near_), [zn] "+v"(pz.near_),
              [xf] "+v"(px.far_), [yf] "+v"(py.far_), [zf] "+v"(pz.far_)
            : [rnx] "v"(R.nx), [rny] "v"(R.ny), [rnz] "v"(R.nz), [rfx] "v"(R.fx), [rfy] "v"(R.fy), [rfz] "v"(R.fz),
              [rex] "v"(R.ex), [rey] "v"(R.ey), [rez] "v"(R.ez));
    else
        asm(RT_PKFMA("%[xn]", "%[xn]", "%[rnx]") RT_PKFMA("%[yn]", "%[yn]", "%[rny]") RT_PKFMA("%[zn]", "%[zn]", "%[rnz]")
            RT_PKFMA("%[xf]", "%[xf]", "%[rfx]") RT_PKFMA("%[yf]", "%[yf]", "%[rfy]") "v_pk_fma_f32 %[zf], %[zf], %[rfz], %[rfz] op_sel:[0,0,1] op_sel_hi:[1,0,1]"
            : [xn] "+v"(px.near_), [yn] "+v"(py.near_), [zn] "+v"(pz.near_),
              [xf] "+v"(px.far_), [yf] "+v"(py.far_), [zf] "+v"(pz.far_)
            : [rnx] "v"(R.nx), [rny] "v"(R.ny), [rnz] "v"(R.nz), [rfx] "v"(R.fx), [rfy] "v"(R.fy), [rfz] "v"(R.fz));
    asm("v_max_f32 %[n0], %[zn0], %[tlo]\n\t"
        "v_max_f32 %[n1], %[zn1], %[tlo]\n\t"
        "v_min_f32 %[f0], %[zf0], %[thi]\n\t"
        "v_min_f32 %[f1], %[zf1], %[thi]\n\t"
        "v_max3_f32 %[n0], %[xn0], %[yn0], %[n0]\n\t"
        "v_max3_f32 %[n1], %[xn1], %[yn1], %[n1]\n\t"
        "v_min3_f32 %[f0], %[xf0], %[yf0], %[f0]\n\t"
        "v_min3_f32 %[f1], %[xf1], %[yf1], %[f1]"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [f0] "=&v"(f0), [f1] "=&v"(f1)
        : [xn0] "v"(px.near_.x), [xn1] "v"(px.near_.y), [yn0] "v"(py.near_.x), [yn1] "v"(py.near_.y),
          [zn0] "v"(pz.near_.x), [zn1] "v"(pz.near_.y), [xf0] "v"(px.far_.x), [xf1] "v"(px.far_.y),
          [yf0] "v"(py.far_.x), [yf1] "v"(py.far_.y), [zf0] "v"(pz.far_.x), [zf1] "v"(pz.far_.y),
          [tlo] "v"(t_lo), [thi] "v"(t_hi));
    if (PERP)
        asm("v_max3_f32 %[n0], %[xe0], %[ye0], %[n0]\n\t"
            "v_max3_f32 %[n1], %[xe1], %[ye1], %[n1]\n\t"
            "v_max_f32 %[n0], %[ze0], %[n0]\n\t"
            "v_max_f32 %[n1], %[ze1], %[n1]"
            : [n0] "+v"(n0), [n1] "+v"(n1)
            : [xe0] "v"(xe.x), [xe1] "v"(xe.y), [ye0] "v"(ye.x), [ye1] "v"(ye.y), [ze0] "v"(ze.x), [ze1] "v"(ze.y));
}
#undef RT_PKFMA

enum { KIND_BRUTE_SPHERES = 0, KIND_BVH_SPHERES = 1, KIND_BVH_TRIANGLES = 2 };

struct Outputs {
    float              *accum;
    float              *partial;   // chunked mode: [n_chunks][P*3] partial sums, each cell written by one lane
    uint32_t           *raycount;
    uint32_t           *queue;
    uint32_t           *wave_queue;   // LDS: this wave's reserved item range {next, end}
    unsigned long long  my_rays, my_primary;
};

// A lane without a ray moves to its next (pixel, sample): the next sample of its pixel, or -- when the
// pixel is done -- it writes the pixel out and pulls the next pixel from the global queue (one
// wave-aggregated atomic for all lanes of the wave that need one).  Render's loops, raytracer.zig:521-528,
// and trace's, :536-541.  Returns true when a camera ray is to be generated for (L.pix, L.sample).
__device__ __forceinline__ bool advance_direct(const TraceParams &P, Lane &L, bool &alive, Outputs &O, uint32_t lane, uint32_t *prof)
{
    if (L.sample >= L.send) {
        prof_block(prof, RT_PB_DEQUEUE);
        if (L.pix != NOPIX) {                         // (pixel, sample range) finished: this lane is its only writer
            if (P.n_chunks > 1u) {
                float *dst = O.partial + (size_t)L.part * P.plane_stride + 3ull * L.pix;
                dst[0] = L.col.x; dst[1] = L.col.y; dst[2] = L.col.z;
                if (O.raycount) atomicAdd(&O.raycount[L.pix], L.nrays);
            } else {
                float *dst = O.accum + 3ull * L.pix;
                dst[0] += L.col.x; dst[1] += L.col.y; dst[2] += L.col.z;
                if (O.raycount) O.raycount[L.pix] += L.nrays;
            }
            O.my_rays += L.nrays;
            L.pix = NOPIX;
        }
        // Dequeue.  Every wave owns a reserved range of work items [next, end) (kept in LDS, touched only by the
        // wave's leader lane); the lanes that need an item take consecutive ones from it (ballot + popcount
        // rank), and only when the range runs dry does the leader reserve the next P.queue_batch items with ONE
        // atomic on the global head.  (One atomic per item saturates a single counter at this kernel's rate.)
        const unsigned long long m = wave_ballot(1);
        const uint32_t cnt = (uint32_t)__popcll(m);
        const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
        uint32_t q_next = 0, q_avail = 0, q_new = 0;
        if (lane == leader) {
            uint32_t *wq = O.wave_queue;
            q_next = wq[0];
            q_avail = wq[1] - q_next;
            if (cnt > q_avail) {
                q_new = atomicAdd(O.queue, P.queue_batch);
                wq[0] = q_new + (cnt - q_avail);
                wq[1] = q_new + P.queue_batch;
            } else {
                wq[0] = q_next + cnt;
            }
        }
        q_next = __builtin_amdgcn_readfirstlane(q_next);   // leader = first active lane
        q_avail = __builtin_amdgcn_readfirstlane(q_avail);
        q_new = __builtin_amdgcn_readfirstlane(q_new);
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t item = rank < q_avail ? q_next + rank : q_new + (rank - q_avail);
        if (item >= P.n_items) {
            alive = false;
        } else {
            // item -> (tile, chunk, pixel): 8x8 tiles, row-major inside a tile, so a wave starts on one
            // coherent tile; the chunks of a tile are consecutive items
            const uint32_t slot = item >> 6, within = item & 63u;
            const uint32_t tile = slot / P.n_chunks, part = slot - tile * P.n_chunks;
            const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
            const uint32_t i = P.x0 + tx * 8u + (within & 7u);
            const uint32_t jr = P.y0 + ty * 8u + (within >> 3);
            if (i < P.x1 && jr < P.y1) {
                L.pix = jr * P.nx + i;                // raytracer.zig:521-528 output index
                L.i = i;
                L.j = P.q3 ? (P.ny - jr) : (P.ny - 1u - jr);   // raytracer.zig:523
                L.part = part;
                if (P.n_chunks > 1u) {                // chunk k covers global samples [k*C, (k+1)*C)
                    const uint32_t k = P.chunk0 + part;
                    const uint32_t lo = k * P.spp_chunk, hi = lo + P.spp_chunk;
                    L.sample = lo > P.spp_begin ? lo : P.spp_begin;
                    L.send = hi < P.spp_end ? hi : P.spp_end;
                } else {
                    L.sample = P.spp_begin;
                    L.send = P.spp_end;
                }
                L.col = mk(0, 0, 0);
                L.nrays = 0;
            }
        }
    }
    return L.pix != NOPIX && L.sample < L.send;
}

// advance_direct's job with the dequeue done for the whole wave at once and one item ahead (brute-force kernel).
// Some lane finishes an item in nearly every
// shading pass (2-3 of 64), so a dequeue + decode done by the finishing lanes alone would run on every pass with
// the wave almost empty.  Instead every lane keeps its NEXT item decoded in registers; a finishing lane just
// switches to it, and only when a lane finds that slot empty does the wave refill the empty slots of ALL its
// lanes in one go (about every 15th pass).  Called by all lanes that take part in the shading pass.
// Every wave owns a reserved range of items [next, end) in LDS, touched only by its leader lane; the lanes take
// consecutive items from it (ballot + popcount rank), and only when the range runs dry does the leader reserve
// the next P.queue_batch items with ONE atomic on the global head.  (One atomic per item saturates a single
// counter at this kernel's rate.)
__device__ __forceinline__ bool advance_prefetch(const TraceParams &P, Lane &L, bool idle, bool &alive, Outputs &O, uint32_t lane,
                                                 uint32_t *prof)
{
    const bool fin = idle && L.sample >= L.send;
    if (fin && L.pix != NOPIX) {                      // item finished: this lane is its only writer
        if (P.n_chunks > 1u) {
            float *dst = O.partial + (size_t)L.part * P.plane_stride + 3ull * L.pix;
            dst[0] = L.col.x; dst[1] = L.col.y; dst[2] = L.col.z;
            if (O.raycount) atomicAdd(&O.raycount[L.pix], L.nrays);
        } else {
            float *dst = O.accum + 3ull * L.pix;
            dst[0] += L.col.x; dst[1] += L.col.y; dst[2] += L.col.z;
            if (O.raycount) O.raycount[L.pix] += L.nrays;
        }
        O.my_rays += L.nrays;
        L.pix = NOPIX;
    }
    if (wave_ballot(fin && L.nx_pix == NOPIX && !L.dry) != 0ull) {     // wave-uniform: refill every empty slot
        prof_block(prof, RT_PB_DEQUEUE);
        const bool take = L.nx_pix == NOPIX && !L.dry;
        const unsigned long long m = wave_ballot(take);
        const uint32_t cnt = (uint32_t)__popcll(m);
        const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
        uint32_t q_next = 0, q_avail = 0, q_new = 0;
        if (lane == leader) {
            uint32_t *wq = O.wave_queue;
            q_next = wq[0];
            q_avail = wq[1] - q_next;
            if (cnt > q_avail) {
                q_new = atomicAdd(O.queue, P.queue_batch);
                wq[0] = q_new + (cnt - q_avail);
                wq[1] = q_new + P.queue_batch;
            } else {
                wq[0] = q_next + cnt;
            }
        }
        q_next = __builtin_amdgcn_readlane(q_next, leader);    // leader is wave-uniform (from the ballot)
        q_avail = __builtin_amdgcn_readlane(q_avail, leader);
        q_new = __builtin_amdgcn_readlane(q_new, leader);
        if (take) {
            const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            const uint32_t item = rank < q_avail ? q_next + rank : q_new + (rank - q_avail);
            if (item >= P.n_items) {
                L.dry = true;
            } else {
                // item -> (tile, chunk, pixel): 8x8 tiles, row-major inside a tile, so a wave starts on one
                // coherent tile; the chunks of a tile are consecutive items
                const uint32_t slot = item >> 6, within = item & 63u;
                const uint32_t tile = slot / P.n_chunks, part = slot - tile * P.n_chunks;
                const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
                const uint32_t i = P.x0 + tx * 8u + (within & 7u);
                const uint32_t jr = P.y0 + ty * 8u + (within >> 3);
                if (i < P.x1 && jr < P.y1) {          // (tiles may overhang the rect: such an item is empty)
                    L.nx_pix = jr * P.nx + i;         // raytracer.zig:521-528 output index
                    L.nx_i = i;
                    L.nx_j = P.q3 ? (P.ny - jr) : (P.ny - 1u - jr);   // raytracer.zig:523
                    L.nx_part = part;
                }
            }
        }
    }
    if (fin) {
        if (L.nx_pix != NOPIX) {
            L.pix = L.nx_pix; L.i = L.nx_i; L.j = L.nx_j; L.part = L.nx_part;
            L.nx_pix = NOPIX;
            if (P.n_chunks > 1u) {                    // chunk k covers global samples [k*C, (k+1)*C)
                const uint32_t k = P.chunk0 + L.part;
                const uint32_t lo = k * P.spp_chunk, hi = lo + P.spp_chunk;
                L.sample = lo > P.spp_begin ? lo : P.spp_begin;
                L.send = hi < P.spp_end ? hi : P.spp_end;
            } else {
                L.sample = P.spp_begin;
                L.send = P.spp_end;
            }
            L.col = mk(0, 0, 0);
            L.nrays = 0;
        } else if (L.dry) {
            alive = false;                            // queue drained
        }
    }
    return idle && L.pix != NOPIX && L.sample < L.send;
}
// (The BVH kernels keep advance_direct: there the same change left the VALU count where it was and cost 3 % in
// time -- measured, profiles/ notes in DESIGN.md -- while the brute-force kernel gained 15 %.)

// Everything between two world.hit calls of a lane: the rest of color() after the hit test
// (raytracer.zig:555-571) for a lane whose ray is resolved (best/best_t), and the next ray of the lane --
// the scattered ray, or the camera ray of its next sample when the path ended.  The two cases need
// different Philox blocks but never both in one lane, so ONE Philox evaluation serves the whole wave.
// A metal-absorbed lane (1 in 10^4 rays) simply gets its camera ray in the next pass.
template <bool HIST, bool PREFETCH, class SurfaceFn>
__device__ __forceinline__ void shade_and_spawn(const TraceParams &P, Lane &L, bool &alive, bool &has_ray, Outputs &O,
                                                uint32_t lane, float a, float best_t, uint32_t best,
                                                uint32_t *s_hist, SurfaceFn fetch_surface)
{
    bool scatter = false;
    f3 unit = mk(0, 0, 0);
    float len_d = 1.0f;
    Surface S;
    S.kind = K_LAMBERTIAN;
    uint32_t *const prof = HIST ? s_hist + RT_HIST_WORDS : nullptr;   // block counters live behind the depth histogram
    if (has_ray) {
        L.nrays += 1;
        if (HIST) atomicAdd(&s_hist[L.depth < 19u ? L.depth : 19u], 1u);
        len_d = sqrtf(a);                             // vector.zig:10-13 length(direction)
        const bool hit = best != NOHIT;
        if (hit && L.depth <= P.max_depth) {          // raytracer.zig:556-558: the cap is tested after the hit
            prof_block(prof, RT_PB_SURFACE);
            S.p = L.o + best_t * L.d;                 // ray.zig:10-12
            fetch_surface(best, S, prof);
            scatter = true;
        }
        if (!hit || S.kind != K_LAMBERTIAN) { prof_block(prof, RT_PB_UNIT); unit = L.d / len_d; }   // normalize: three true divides
        if (!hit) {                                   // sky, raytracer.zig:569-571
            prof_block(prof, RT_PB_SKY);
            const float tt = 0.5f * (unit.y + 1.0f);
            const f3 sky = mk((1.0f - tt) + tt * 0.5f, (1.0f - tt) + tt * 0.7f, (1.0f - tt) + tt * 1.0f);
            L.col = L.col + L.thr * sky;
        }
        if (!scatter) { L.sample += 1; has_ray = false; }
    }
    bool gen = false;
    if (PREFETCH) gen = advance_prefetch(P, L, !has_ray && alive, alive, O, lane, prof);
    else if (!has_ray && alive) gen = advance_direct(P, L, alive, O, lane, prof);
    if (scatter || gen) {
        prof_block(prof, RT_PB_PHILOX);
        uint32_t r[4];
        philox4x32_10(L.pix, L.sample, scatter ? L.depth + 1u : 0u, 0u, P.key0, P.key1, r);
        if (gen) {
            prof_block(prof, RT_PB_CAMERA);
            camera_ray(P, L, r);
            has_ray = true;
            O.my_primary += 1;
        } else if (!scatter_ray(L, S, unit, len_d, r, prof)) {
            L.sample += 1; has_ray = false;
        }
    }
}

__device__ __forceinline__ void finish(const Outputs &O, unsigned long long *stats, uint32_t *s_hist, bool hist)
{
    atomicAdd(&stats[0], O.my_rays);
    atomicAdd(&stats[1], O.my_primary);
    if (hist) {
        __syncthreads();
        if (threadIdx.x < 20) atomicAdd(&stats[2 + threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
        if (threadIdx.x < 2 * RT_PROF_BLOCKS)
            atomicAdd(&stats[28 + threadIdx.x], (unsigned long long)s_hist[RT_HIST_WORDS + threadIdx.x]);
    }
}

// ---- brute force: small sphere scenes ------------------------------------------------------------------
template <bool HIST>
__global__ __launch_bounds__(BLOCK) void rt_trace_brute_kernel(
    const TraceParams P, const float4 *__restrict__ geom, const float4 *__restrict__ aux,
    const float4 *__restrict__ albedo, float *__restrict__ accum, float *__restrict__ partial,
    uint32_t *__restrict__ raycount, uint32_t *__restrict__ queue, unsigned long long *__restrict__ stats)
{
    __shared__ uint32_t s_hist[RT_HIST_WORDS + 2 * RT_PROF_BLOCKS];
    __shared__ uint32_t s_wq[2 * (BLOCK / 64)];
    if (HIST && threadIdx.x < RT_HIST_WORDS + 2 * RT_PROF_BLOCKS) s_hist[threadIdx.x] = 0;
    if (threadIdx.x < 2 * (BLOCK / 64)) s_wq[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    Lane L;
    L.o = mk(0, 0, 0); L.d = mk(0, 0, 1); L.thr = mk(0, 0, 0); L.col = mk(0, 0, 0);
    L.pix = NOPIX; L.i = 0; L.j = 0; L.sample = 0; L.send = 0; L.part = 0; L.depth = 0; L.nrays = 0;
    L.nx_pix = NOPIX; L.nx_i = 0; L.nx_j = 0; L.nx_part = 0; L.dry = false;
    bool alive = true, has_ray = false;
    Outputs O{ accum, partial, raycount, queue, s_wq + 2 * (threadIdx.x >> 6), 0ull, 0ull };

    float a = 1.0f, best_t = T_MAX;
    uint32_t best = NOHIT;
    for (;;) {
        // resolve the rays of the previous iteration and give every lane its next ray
        shade_and_spawn<HIST, true>(P, L, alive, has_ray, O, lane, a, best_t, best, s_hist, [&](uint32_t prim, Surface &S, uint32_t *prof) {
            const float4 g = geom[prim], ax = aux[prim], al = albedo[2u * prim], al2 = albedo[2u * prim + 1u];
            S.n = sphere_normal(S.p - mk(g.x, g.y, g.z), ax, prof);
            S.kind = __float_as_uint(ax.y); S.param = ax.z; S.albedo = mk(al.x, al.y, al.z); S.r0sq = al.w; S.inv_param = al2.x;
        });
        if (wave_ballot(alive || has_ray) == 0ull) break;             // queue drained and nothing in flight

        // world.hit: closest root over all spheres in list order (raytracer.zig:92-113).  Lanes without a
        // ray run the loop on dummy data.
        a = dot(L.d, L.d);
        best_t = T_MAX;
        best = NOHIT;
        for (uint32_t s = 0; s < P.n_prim; ++s) {
            const float4 g = geom[s];                 // wave-uniform address -> scalar load
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
    }
    finish(O, stats, s_hist, HIST);
}

// ---- BVH traversal: spheres or triangles ----------------------------------------------------------------
// Traversal and shading are decoupled: a lane is either TRAVERSING (node != DONE) or waiting to be shaded.
// The wave keeps taking traversal steps while at least KEEP_TRAVERSING lanes still traverse; below that it
// runs the exact tests, shades the finished lanes and gives them their next ray (bounce or new camera
// ray), and they rejoin the traversal at the root beside the lanes that are still on their way.  Lanes
// therefore do not idle for the wave's slowest ray.

//
// LDSN = true (small scenes, e.g. the book scene): the whole scene is staged in LDS once per workgroup --
// BVH nodes re-strided from 64 to 80 bytes (20 dwords, so the 16-byte reads of 16 lanes at random nodes spread
// over all bank quads instead of 4) and the sphere records -- and stack/queue entries shrink to 16 bits.
// LDSN = false (large scenes, e.g. 285k triangles): nodes and primitives are read from global memory
// through L1/L2.
constexpr int LDS_NODE_STRIDE = RT_NODE_BYTES / 16;    // float4 per node

// PACKED (large triangle scenes): 72-byte nodes without the refs record -- the vector L1 serves ~2 divergent lane
// requests per clock whatever their width (tools/ubench/gather_rate.hip) and this kernel is bound by that rate, so
// the step's fourth load is folded into the other three: child c's 24-bit ref (bit 23 = leaf, else node INDEX) rides
// in the low mantissa byte of its x, y and z planes (lo and hi carry the same byte, so the byte arrives whichever
// way the ray runs; the host rounds the planes outward to the next float with that byte, rt_api.cpp).
// WIDE (rt_tuning.wide_nodes): 144-byte nodes of up to four children (rt_kernels.h RT_WIDE_NODE_BYTES), refs packed as
// above (PACKED is set with it): a step tests four boxes with two of the two-box slab tests, queues the hit leaves,
// descends into the nearest hit inner child and stacks the others.
template <int KIND, int STACK, int CAND, bool HIST, bool LDSN, int NT, bool PACKED = false, bool WIDE = false>
__global__ __launch_bounds__(NT) void rt_trace_bvh_kernel(
    const TraceParams P,
    const float4 *__restrict__ geom, const float4 *__restrict__ tri, const float4 *__restrict__ aux,
    const float4 *__restrict__ albedo, const float4 *__restrict__ bvh, const uint32_t n_nodes,
    float *__restrict__ accum, float *__restrict__ partial, uint32_t *__restrict__ raycount,
    uint32_t *__restrict__ queue, unsigned long long *__restrict__ stats)
{
    typedef typename std::conditional<LDSN, uint16_t, uint32_t>::type idx_t;
    constexpr int BLOCK = NT;
    __shared__ uint32_t s_hist[HIST ? RT_HIST_WORDS + 2 * RT_PROF_BLOCKS : 1];
    __shared__ idx_t s_stack[STACK * BLOCK];          // [entry][thread]: a wave's accesses are conflict-free
    // Candidates (leaves whose box the ray's cone touches) wait for their exact test.
    //   spheres:   per-lane queues [entry][thread]; a batch pops one candidate per lane
    //   triangles: ONE queue per wave, entries {owner lane, triangle}: a batch hands one entry to every lane, whoever
    //              owns it -- the lane fetches the owner's ray with a wave shuffle, tests, and merges the result into
    //              the owner's (t, index) word with an LDS atomic min -- so the expensive test runs with full waves
    //              although a ray meets only three or four leaves on ~30 steps (the per-lane form filled 31 % of it)
    constexpr bool SHARED_Q = KIND == KIND_BVH_TRIANGLES;
    constexpr int  WQ_CAP = 64 + (WIDE ? 1 : 2) * 64; // at most 63 left over + two (WIDE: one) candidates per lane and step
    __shared__ idx_t s_cand[SHARED_Q ? 1 : CAND * BLOCK];
    __shared__ uint32_t s_wcand[SHARED_Q ? (NT / 64) * WQ_CAP : 1];
    __shared__ unsigned long long s_wbest[SHARED_Q ? NT : 1];      // per lane: bits(t) << 32 | triangle: min = closest, ties to the lower index
    extern __shared__ float4 s_scene[];               // LDSN: [n_nodes * 5] nodes, then [n_prim] sphere records
    const float4 *s_geom = s_scene + (LDSN ? n_nodes * LDS_NODE_STRIDE : 0u);
    __shared__ uint32_t s_wq[2 * (NT / 64)];
    if (HIST && threadIdx.x < RT_HIST_WORDS + 2 * RT_PROF_BLOCKS) s_hist[threadIdx.x] = 0;
    if (threadIdx.x < 2 * (NT / 64)) s_wq[threadIdx.x] = 0;
    if (LDSN) {
        // the device nodes are already in traversal format (80 bytes, child refs = byte offsets): straight copy.
        // The 80-byte stride also spreads 16 lanes' reads of random nodes over all bank quads.
        for (uint32_t k = threadIdx.x; k < n_nodes * LDS_NODE_STRIDE; k += NT) s_scene[k] = bvh[k];
        if (KIND == KIND_BVH_SPHERES)
            for (uint32_t k = threadIdx.x; k < P.n_prim; k += NT)
                s_scene[n_nodes * LDS_NODE_STRIDE + k] = geom[k];
    }
    __syncthreads();
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;

    Lane L;
    L.o = mk(0, 0, 0); L.d = mk(0, 0, 1); L.thr = mk(0, 0, 0); L.col = mk(0, 0, 0);
    L.pix = NOPIX; L.i = 0; L.j = 0; L.sample = 0; L.send = 0; L.part = 0; L.depth = 0; L.nrays = 0;
    L.nx_pix = NOPIX; L.nx_i = 0; L.nx_j = 0; L.nx_part = 0; L.dry = false;
    bool alive = true, has_ray = false;
    Outputs O{ accum, partial, raycount, queue, s_wq + 2 * (threadIdx.x >> 6), 0ull, 0ull };

    // traversal state of the lane's current ray
    uint32_t node = DONE, sp = 0, ncand = 0, best = NOHIT;
    uint32_t wcount = 0;                              // SHARED_Q: entries in this wave's queue (wave-uniform)
    uint32_t *const wq = s_wcand + (SHARED_Q ? (tid >> 6) * WQ_CAP : 0);
    constexpr unsigned long long NO_HIT_KEY = ((unsigned long long)0x7f7fffffu << 32) | 0xFFFFFFFFull;   // (T_MAX, NOHIT)
    float best_t = T_MAX, a = 1.0f;
    // the slab test's ray operands as register pairs, made once per ray (ray_splat).  The empty asm keeps the
    // compiler from re-copying t_min (1 v_mov) on every traversal step.
    RaySplat RS;
    RS.nx = RS.ny = RS.nz = RS.fx = RS.fy = RS.fz = RS.ex = RS.ey = RS.ez = (v2f){ 1.0f, 0.0f };
    RS.ox = 8u; RS.oy = NODE_AXIS_BYTES + 8u; RS.oz = 2u * NODE_AXIS_BYTES + 8u;
    RS.perp = false;
    bool wave_perp = false;           // wave-uniform: some lane's current ray has an axis inside its cone
    float t_min_v = P.t_min;
    asm volatile("" : "+v"(t_min_v));

    // HIST builds also count how well the wave is filled in each phase (profiling aid; lane 0 of a wave
    // counts wave-level events, every lane its own)
    unsigned long long c_wsteps = 0, c_lsteps = 0, c_wflush = 0, c_lflush = 0, c_wshade = 0, c_lshade = 0;

    // One batch of exact primitive tests: every lane that holds candidates pops its newest one and tests it
    // (the order of the tests does not matter: the closest hit is the lexicographic minimum of (t, index)).
    // Popping one per lane keeps the batch as full as the number of lanes that hold candidates, instead of
    // looping to the longest queue with ever fewer lanes.
    // SHARED_Q: one batch = the newest min(64, wcount) entries of the wave's queue, one per lane
    auto test_batch = [&]() {
        if (HIST) c_wflush += 1;
        const uint32_t n = wcount < 64u ? wcount : 64u, base = wcount - n;
        const bool mine = lane < n;
        const uint32_t e = mine ? wq[base + lane] : (lane << 26);
        const uint32_t owner = e >> 26, prim = e & (WIDE ? 0x007FFFFFu : 0x03FFFFFFu);    // WIDE entries still carry the ref's leaf bit
        // the owner's ray (every lane takes part in the shuffles: a lane's registers are read by others)
        Lane R;
        R.o = mk(__shfl(L.o.x, (int)owner), __shfl(L.o.y, (int)owner), __shfl(L.o.z, (int)owner));
        R.d = mk(__shfl(L.d.x, (int)owner), __shfl(L.d.y, (int)owner), __shfl(L.d.z, (int)owner));
        if (mine) {
            if (HIST) c_lflush += 1;
            const float4 p0 = tri[3u * prim], p1 = tri[3u * prim + 1u], p2 = tri[3u * prim + 2u];
            float t = T_MAX; uint32_t hit = NOHIT;
            triangle_exact(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), prim, R, P.t_min, t, hit);
            if (hit != NOHIT)       // t > t_min >= 0: the bits of t order like t
                atomicMin(&s_wbest[(tid & ~63u) + owner], ((unsigned long long)__float_as_uint(t) << 32) | prim);
        }
        wcount = base;
        const unsigned long long k = s_wbest[tid];
        best_t = __uint_as_float((uint32_t)(k >> 32)); best = (uint32_t)k;
    };
    auto test_one = [&]() {
        if (HIST) c_wflush += 1;
        if (ncand != 0u) {
            if (HIST) c_lflush += 1;
            ncand -= 1;
            const uint32_t prim = s_cand[ncand * BLOCK + tid];
            if (KIND == KIND_BVH_SPHERES) {
                sphere_exact(LDSN ? s_geom[prim] : geom[prim], prim, L, a, P.t_min, best_t, best);
            } else {
                const float4 p0 = tri[3u * prim], p1 = tri[3u * prim + 1u], p2 = tri[3u * prim + 2u];
                triangle_exact(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), prim, L,
                               P.t_min, best_t, best);
            }
        }
    };

    for (;;) {
        // ---- lanes whose traversal is complete: exact tests, shade, next ray --------------------------
        if (wave_ballot(node == DONE && (has_ray || alive)) != 0ull) {
            // drain the queues of the finished lanes; lanes still traversing join each batch (their best_t tightens)
            if constexpr (SHARED_Q) { while (wcount != 0u) test_batch(); }
            else { while (wave_ballot(node == DONE && ncand != 0u) != 0ull) test_one(); }
            if (HIST) c_wshade += 1;
            if (node == DONE) {
                if (HIST && has_ray) c_lshade += 1;
                shade_and_spawn<HIST, false>(P, L, alive, has_ray, O, lane, a, best_t, best, s_hist, [&](uint32_t prim, Surface &S, uint32_t *prof) {
                    uint32_t mat;
                    if (KIND == KIND_BVH_TRIANGLES) {
                        const float4 p0 = tri[3u * prim], p1 = tri[3u * prim + 1u], p2 = tri[3u * prim + 2u];
                        const f3 v0 = mk(p0.x, p0.y, p0.z);
                        const f3 e1 = mk(p1.x, p1.y, p1.z) - v0, e2 = mk(p2.x, p2.y, p2.z) - v0;
                        const f3 cr = cross(e1, e2);
                        S.n = cr / sqrtf(dot(cr, cr));
                        mat = __float_as_uint(p0.w);
                    } else {
                        const float4 g = LDSN ? s_geom[prim] : geom[prim];
                        mat = prim;
                        S.n = sphere_normal(S.p - mk(g.x, g.y, g.z), aux[mat], prof);
                    }
                    const float4 ax = aux[mat], al = albedo[2u * mat], al2 = albedo[2u * mat + 1u];
                    S.kind = __float_as_uint(ax.y); S.param = ax.z; S.albedo = mk(al.x, al.y, al.z); S.r0sq = al.w; S.inv_param = al2.x;
                });
                if (has_ray) {                        // start the new ray at the root
                    a = dot(L.d, L.d);
                    ray_splat(RS, L.o, L.d, a, KIND == KIND_BVH_SPHERES ? RT_CONE_SPHERES : RT_CONE_TRIANGLES);
                    if (WIDE) {                       // wide axis record {hi x4, lo x4, hi x4}: near quad at +16 (up) or +0 (down), far quad behind it
                        RS.ox = ((__float_as_uint(L.d.x) >> 31) ^ 1u) << 4;
                        RS.oy = 48u + (((__float_as_uint(L.d.y) >> 31) ^ 1u) << 4);
                        RS.oz = 96u + (((__float_as_uint(L.d.z) >> 31) ^ 1u) << 4);
                    }
                    node = 0u; sp = 0; best_t = T_MAX; best = NOHIT;
                    if (SHARED_Q) s_wbest[tid] = NO_HIT_KEY;
                }
            }
        }
        if (wave_ballot(alive || has_ray) == 0ull) break;         // queue drained and nothing in flight
        wave_perp = wave_ballot(RS.perp && node != DONE) != 0ull;

        // ---- traversal steps: every iteration is an inner-node visit for every traversing lane; hit
        //      leaves are only queued.  Two copies of the loop: the slab test's extra term (box2_hit_fma<true>)
        //      runs only in a traversal phase in which some lane needs it.
        // WIDE: a stack entry carries, above its 24-bit ref, an 8-bit code of the box's entry distance (a monotonic,
        // clamped cut of the float's exponent and top three mantissa bits)
        auto dist_code = [](float t) {           // t >= 0
            const int q = (int)(__float_as_uint(t) >> 20) - (111 << 3);      // 0 at 2^-16, 255 just below 2^16
            return (uint32_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        };
        auto traverse = [&](auto perp_tag) {
        constexpr bool PERP = decltype(perp_tag)::value;
        for (;;) {
            if (HIST) c_wsteps += 1;
            if constexpr (WIDE) {
                constexpr uint32_t POP = 0xFFFFFFFEu;             // "take the next entry off the stack" (never survives a step)
                // ---- the lane's node is an inner node: test its four child boxes; ALL hit children, leaves included, are
                //      sorted by entry distance (five compare-exchanges; a miss or a missing child sorts behind everything),
                //      the nearest becomes the lane's node and the others go on the stack farthest first
                if (node < RT_PACKED_LEAF_BIT) {
                    if (HIST) c_lsteps += 1;
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const char *const nbase = (const char *)bvh;
                    const uint32_t noff = __umul24(node, (uint32_t)RT_WIDE_NODE_BYTES);
                    const v4f xn = *(const v4f *)(nbase + (noff + RS.ox)), xf = *(const v4f *)(nbase + (noff + RS.ox + 16u));
                    const v4f yn = *(const v4f *)(nbase + (noff + RS.oy)), yf = *(const v4f *)(nbase + (noff + RS.oy + 16u));
                    const v4f zn = *(const v4f *)(nbase + (noff + RS.oz)), zf = *(const v4f *)(nbase + (noff + RS.oz + 16u));
                    uint32_t ref[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t xy = __builtin_amdgcn_perm(__float_as_uint(yn[k]), __float_as_uint(xn[k]), 0x0c0c0400u);
                        ref[k] = __builtin_amdgcn_perm(__float_as_uint(zn[k]), xy, 0x0c040100u);
                    }
                    float tn[4], tf[4];
                    {
                        const PlanePairs px = { (v2f){ xn.x, xn.y }, (v2f){ xf.x, xf.y } }, py = { (v2f){ yn.x, yn.y }, (v2f){ yf.x, yf.y } },
                                         pz = { (v2f){ zn.x, zn.y }, (v2f){ zf.x, zf.y } };
                        box2_hit_fma<PERP>(px, py, pz, RS, t_min_v, best_t, tn[0], tn[1], tf[0], tf[1]);
                    }
                    {
                        const PlanePairs px = { (v2f){ xn.z, xn.w }, (v2f){ xf.z, xf.w } }, py = { (v2f){ yn.z, yn.w }, (v2f){ yf.z, yf.w } },
                                         pz = { (v2f){ zn.z, zn.w }, (v2f){ zf.z, zf.w } };
                        box2_hit_fma<PERP>(px, py, pz, RS, t_min_v, best_t, tn[2], tn[3], tf[2], tf[3]);
                    }
                    float key[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) key[k] = tn[k] <= tf[k] ? tn[k] : __builtin_inff();     // (entry distances are finite)
#define RT_CE(i, j) { const bool sw = key[j] < key[i]; const float ki = key[i], kj = key[j]; const uint32_t ri = ref[i], rj = ref[j]; \
                      key[i] = sw ? kj : ki; key[j] = sw ? ki : kj; ref[i] = sw ? rj : ri; ref[j] = sw ? ri : rj; }
                    RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
                    if (key[3] < __builtin_inff()) { s_stack[sp * BLOCK + tid] = (idx_t)(ref[3] | (dist_code(key[3]) << 24)); sp += 1; }
                    if (key[2] < __builtin_inff()) { s_stack[sp * BLOCK + tid] = (idx_t)(ref[2] | (dist_code(key[2]) << 24)); sp += 1; }
                    if (key[1] < __builtin_inff()) { s_stack[sp * BLOCK + tid] = (idx_t)(ref[1] | (dist_code(key[1]) << 24)); sp += 1; }
                    node = key[0] < __builtin_inff() ? ref[0] : POP;
                }
                // ---- the lane's node is a leaf (just chosen as the nearest child, or popped in the step before): its
                //      triangle joins the wave's queue (the entry keeps the ref's leaf bit; test_batch drops it)
                const bool at_leaf = node - RT_PACKED_LEAF_BIT < RT_PACKED_LEAF_BIT;          // refs 2^23 .. 2^24 - 1
                {
                    const unsigned long long m = wave_ballot(at_leaf);
                    if (at_leaf) wq[wcount + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (lane << 26) | node;
                    wcount += (uint32_t)__popcll(m);
                }
                // ---- next node off the stack.  An entry whose distance code is ABOVE the code of the closest hit found
                //      since it was pushed lies behind that hit with its whole subtree (or its triangle) and is dropped:
                //      code(a) > code(b) implies a > b, so nothing that could still hold a closer hit is dropped.
                if (at_leaf || node == POP) {
                    node = DONE;
                    if (sp > 0) {
                        const uint32_t best_code = dist_code(best_t);
                        uint32_t e;
                        do { sp -= 1; e = s_stack[sp * BLOCK + tid]; } while ((e >> 24) > best_code && sp > 0);
                        if ((e >> 24) <= best_code) node = e & 0x00FFFFFFu;
                    }
                }
                while (wcount >= 64u) test_batch();
                if ((uint32_t)__popcll(wave_ballot(node != DONE)) < P.keep_traversing) break;
                continue;
            }
            bool push0 = false, push1 = false;
            uint32_t cand0 = 0, cand1 = 0;
            if (node != DONE) {
                if (HIST) c_lsteps += 1;
                float tn0, tn1, tf0, tf1;
                // node = byte offset (PACKED: node index)
                // 32-bit byte offsets from a wave-uniform base: the loads take the base from SGPRs (no 64-bit address math)
                const char *const nbase = LDSN ? (const char *)s_scene : (const char *)bvh;
                const uint32_t noff = PACKED ? __umul24(node, (uint32_t)RT_PACKED_NODE_BYTES) : node;   // node index < 2^23
                const char *nd = nbase + noff;
                const PlanePairs px = *(const PlanePairs *)(nbase + (noff + RS.ox)), py = *(const PlanePairs *)(nbase + (noff + RS.oy)),
                                 pz = *(const PlanePairs *)(nbase + (noff + RS.oz));
                uint32_t c0, c1;
                constexpr uint32_t LEAF = PACKED ? RT_PACKED_LEAF_BIT : LEAF_BIT;
                if constexpr (PACKED) {
                    // {x.b0, y.b0, z.b0} of the near planes of child 0 (.x) and child 1 (.y): two v_perm_b32 each
                    const uint32_t xy0 = __builtin_amdgcn_perm(__float_as_uint(py.near_.x), __float_as_uint(px.near_.x), 0x0c0c0400u);
                    const uint32_t xy1 = __builtin_amdgcn_perm(__float_as_uint(py.near_.y), __float_as_uint(px.near_.y), 0x0c0c0400u);
                    c0 = __builtin_amdgcn_perm(__float_as_uint(pz.near_.x), xy0, 0x0c040100u);
                    c1 = __builtin_amdgcn_perm(__float_as_uint(pz.near_.y), xy1, 0x0c040100u);
                } else {
                    const uint2 refs = *(const uint2 *)(nd + NODE_REFS_OFFSET);
                    c0 = refs.x; c1 = refs.y;
                }
                box2_hit_fma<PERP>(px, py, pz, RS, t_min_v, best_t, tn0, tn1, tf0, tf1);
                const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
                const bool l0 = (c0 & LEAF) != 0u, l1 = (c1 & LEAF) != 0u;
                if constexpr (!SHARED_Q) {
                    if (h0 && l0) { s_cand[ncand * BLOCK + tid] = (idx_t)(c0 & ~LEAF); ncand += 1; }
                    if (h1 && l1) { s_cand[ncand * BLOCK + tid] = (idx_t)(c1 & ~LEAF); ncand += 1; }
                }
                push0 = h0 && l0; push1 = h1 && l1; cand0 = c0 & ~LEAF; cand1 = c1 & ~LEAF;
                const bool v0 = h0 && !l0, v1 = h1 && !l1;
                if (v0 && v1) {                       // descend into the nearer child, stack the other
                    const bool first0 = tn0 <= tn1;
                    s_stack[sp * BLOCK + tid] = (idx_t)(first0 ? c1 : c0);
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
            // run a batch of exact tests when enough lanes hold candidates to fill it (a hit found early also
            // tightens best_t and prunes the rest of that lane's traversal), or when a queue has no room for the
            // two candidates a step can add
            if constexpr (SHARED_Q) {
                // append this step's candidates behind the queue's tail (wave-wide prefix counts: no atomics), then run
                // full batches
                const unsigned long long m0 = wave_ballot(push0), m1 = wave_ballot(push1);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (push0) wq[wcount + (uint32_t)__popcll(m0 & below)] = (lane << 26) | cand0;
                wcount += (uint32_t)__popcll(m0);
                if (push1) wq[wcount + (uint32_t)__popcll(m1 & below)] = (lane << 26) | cand1;
                wcount += (uint32_t)__popcll(m1);
                while (wcount >= 64u) test_batch();
            } else {
                if ((uint32_t)__popcll(wave_ballot(ncand != 0u)) >= P.flush_lanes) test_one();
                while (wave_ballot(ncand > (uint32_t)(CAND - 2)) != 0ull) test_one();
            }
            if ((uint32_t)__popcll(wave_ballot(node != DONE)) < P.keep_traversing) break;
        }
        };
        if (wave_perp) traverse(std::true_type{}); else traverse(std::false_type{});
    }
    finish(O, stats, s_hist, HIST);
    if (HIST) {
        if (lane == 0) { atomicAdd(&stats[22], c_wsteps); atomicAdd(&stats[24], c_wflush); atomicAdd(&stats[26], c_wshade); }
        atomicAdd(&stats[23], c_lsteps); atomicAdd(&stats[25], c_lflush); atomicAdd(&stats[27], c_lshade);
    }
}

// start of a pass: work-queue head and counters to zero.  A kernel rather than two hipMemsetAsync: under stream capture
// (rt_render_device_noalloc) the 4-byte memset node did not run on replay with ROCm 7.2 -- the queue head kept the last
// pass's value and every wave left at once -- and one launch is cheaper than two memsets anyway.
__global__ __launch_bounds__(64) void rt_clear_pass_kernel(uint32_t *queue, unsigned long long *stats, uint32_t n_stats)
{
    if (threadIdx.x == 0) queue[0] = 0u;
    for (uint32_t k = threadIdx.x; k < n_stats; k += 64u) stats[k] = 0ull;
}

// chunked mode: accum += (((p[0] + p[1]) + p[2]) + ...), the chunk partial sums added in sample order
__global__ __launch_bounds__(256) void rt_reduce_chunks_kernel(float *accum, const float *partial, uint32_t n_values,
                                                               uint32_t n_chunks, size_t plane_stride)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_values) return;
    float acc = 0.0f;
    for (uint32_t c = 0; c < n_chunks; ++c) acc += partial[c * plane_stride + k];
    accum[k] += acc;
}

// rt_render_multi on a shared GPU: ray counts of the peers added to the root's
__global__ __launch_bounds__(256) void rt_u32_add_planes_kernel(uint32_t *dst, const uint32_t *planes, uint32_t n,
                                                                uint32_t n_planes, size_t plane_stride)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t acc = dst[k];
    for (uint32_t c = 0; c < n_planes; ++c) acc += planes[c * plane_stride + k];
    dst[k] = acc;
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

// render_and_write's merge of the per-thread images (raytracer.zig:610-617): per-channel integer sum of the
// gamma'd u8 images, then a truncating divide by the image count
__global__ __launch_bounds__(256) void rt_u8_sum_kernel(uint32_t *sum, const uint8_t *rgb8, uint32_t n)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) sum[k] += (uint32_t)rgb8[k];
}
__global__ __launch_bounds__(256) void rt_u8_mean_kernel(const uint32_t *sum, uint32_t n_images, uint8_t *rgb8, uint32_t n)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) rgb8[k] = (uint8_t)(sum[k] / n_images);
}

// production cbrt (fast path + fallback exactly as scatter_ray uses it) on the whole unit grid k * 2^-24
__global__ __launch_bounds__(256) void rt_selftest_cbrt_grid_kernel(float *out, unsigned long long *n_risky)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;      // grid covers exactly 2^24 threads
    const float x = u32_to_unit_float(k << 8);
    bool risky;
    float cb = cbrt_unit_fast(x, risky);
    if (wave_ballot(risky) != 0ull) cb = cbrt_f32(x);
    if (risky) atomicAdd(n_risky, 1ull);
    out[k] = cb;
}

__global__ void rt_selftest_kernel(uint32_t n, const float *a, const float *b, float *out_div, float *out_sqrt,
                                   float *out_cbrt, const uint32_t *ctr0, uint32_t *philox_out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    // even k: the IEEE divide; odd k: div_recip with r = RN(1/b) wherever the kernels would use it (ranges as in
    // sphere_normal / recip_for_div) -- the host compares both against its own a / b
    const float av = __builtin_fabsf(a[k]), bv = __builtin_fabsf(b[k]);
    const bool eligible = (k & 1u) && av >= 0x1.0p-100f && av <= 0x1.0p60f && bv >= 0x1.0p-60f && bv <= 0x1.0p60f &&
                          (__float_as_uint(b[k]) & 0x7fffffu) != 0x7fffffu;
    out_div[k] = eligible ? div_recip(a[k], b[k], 1.0f / b[k]) : a[k] / b[k];
    out_sqrt[k] = sqrtf(a[k]);
    out_cbrt[k] = cbrt_f32(a[k]);
    uint32_t r[4];
    philox4x32_10(ctr0[k], k, 7u, 0u, 0x12345678u, 0x9abcdef0u, r);
    philox_out[4 * k + 0] = r[0]; philox_out[4 * k + 1] = r[1];
    philox_out[4 * k + 2] = r[2]; philox_out[4 * k + 3] = r[3];
}

static long long grid_for(const void *kern, const TraceParams &p, int num_cus, int threads, size_t dyn_lds)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    long long want = ((long long)p.n_items + threads - 1) / threads;    // no more blocks than work
    long long grid = (long long)num_cus * per_cu;
    if (grid > want) grid = want;
    return grid < 1 ? 1 : grid;
}

// LDS-resident scene: nodes at 80 bytes + sphere records must fit beside the per-thread stack/queue
static size_t lds_scene_bytes(const DeviceScene &s, bool spheres)
{
    return (size_t)s.n_nodes * LDS_NODE_STRIDE * 16 + (spheres ? (size_t)s.n_prim * 16 : 0);
}
static bool fits_lds(const DeviceScene &s, bool spheres)
{
    // two 512-thread workgroups per CU (4 waves per SIMD) share 160 KB: 80 KB each, of which the stacks, queues and
    // counters take 24.3 KB (spheres) / 26.3 KB (triangles, whose cap also decides the node format at upload, rt_api.cpp).
    // Round 3: 55 KB instead of 40 for spheres -- scenes of 430..570 spheres trace 9 % faster from LDS than through L1/L2.
    const size_t cap = spheres ? 55 * 1024 : 40 * 1024;
    return s.bvh_depth <= 16 && s.n_nodes < 65536u && s.n_prim < 32768u && lds_scene_bytes(s, spheres) <= cap;
}

template <int KIND, int STACK, int CAND, bool HIST, bool LDSN, int NT, bool PACKED = false, bool WIDE = false>
hipError_t launch_bvh(const TraceParams &p, const DeviceScene &s, float *accum_sum, float *partial, uint32_t *raycount,
                      uint32_t *queue, unsigned long long *stats, int num_cus, hipStream_t stream)
{
    auto kern = rt_trace_bvh_kernel<KIND, STACK, CAND, HIST, LDSN, NT, PACKED, WIDE>;
    const size_t dyn = LDSN ? lds_scene_bytes(s, KIND == KIND_BVH_SPHERES) : 0;
    // (static + dynamic LDS may exceed the 64 KB a kernel gets by default)
    if (LDSN && dyn > 32 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return e;
    }
    const long long grid = grid_for((const void *)kern, p, num_cus, NT, dyn);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), dyn, stream,
                       p, s.geom, s.tri, s.aux, s.albedo, s.bvh, s.n_nodes, accum_sum, partial, raycount, queue, stats);
    return hipGetLastError();
}

template <bool HIST>
hipError_t launch_brute(const TraceParams &p, const DeviceScene &s, float *accum_sum, float *partial, uint32_t *raycount,
                        uint32_t *queue, unsigned long long *stats, int num_cus, hipStream_t stream)
{
    auto kern = rt_trace_brute_kernel<HIST>;
    const long long grid = grid_for((const void *)kern, p, num_cus, BLOCK, 0);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), 0, stream,
                       p, s.geom, s.aux, s.albedo, accum_sum, partial, raycount, queue, stats);
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
                           float *partial, uint32_t *raycount, uint32_t *queue, unsigned long long *stats,
                           bool histogram, int num_cus, hipStream_t stream)
{
#define RT_ARGS p, s, accum_sum, partial, raycount, queue, stats, num_cus, stream
    if (which == RT_K_BRUTE_SPHERES) return histogram ? launch_brute<true>(RT_ARGS) : launch_brute<false>(RT_ARGS);
    if (s.bvh_depth > 32 || !s.bvh) return hipErrorInvalidValue;
    const bool spheres = which == RT_K_BVH_SPHERES;
    const bool lds = fits_lds(s, spheres) && !p.force_global_scene;
#define RT_BVH(KIND)                                                                                       \
    if (lds) return histogram ? launch_bvh<KIND, 16, 8, true, true, 512>(RT_ARGS)                          \
                              : launch_bvh<KIND, 16, 8, false, true, 512>(RT_ARGS);                        \
    if (s.bvh_depth <= 16) return histogram ? launch_bvh<KIND, 16, 8, true, false, 256>(RT_ARGS)           \
                                            : launch_bvh<KIND, 16, 8, false, false, 256>(RT_ARGS);         \
    /* 24 stack + 7 queue words per lane = 31 KB per workgroup: five workgroups fit the CU's 160 KB */     \
    if (s.bvh_depth <= 24) return histogram ? launch_bvh<KIND, 24, 7, true, false, 256>(RT_ARGS)           \
                                            : launch_bvh<KIND, 24, 7, false, false, 256>(RT_ARGS);         \
    return histogram ? launch_bvh<KIND, 32, 8, true, false, 256>(RT_ARGS)                                  \
                     : launch_bvh<KIND, 32, 8, false, false, 256>(RT_ARGS);
    if (spheres) { RT_BVH(KIND_BVH_SPHERES) }
    if (s.wide) {
        if (spheres || s.bvh_depth > 11) return hipErrorInvalidValue;      // 3 stack entries per wide level
        return histogram ? launch_bvh<KIND_BVH_TRIANGLES, RT_WIDE_STACK, 8, true, false, 256, true, true>(RT_ARGS)
                         : launch_bvh<KIND_BVH_TRIANGLES, RT_WIDE_STACK, 8, false, false, 256, true, true>(RT_ARGS);
    }
    if (s.packed) {
        if (s.bvh_depth <= 24) return histogram ? launch_bvh<KIND_BVH_TRIANGLES, 24, 7, true, false, 256, true>(RT_ARGS)
                                                : launch_bvh<KIND_BVH_TRIANGLES, 24, 7, false, false, 256, true>(RT_ARGS);
        return histogram ? launch_bvh<KIND_BVH_TRIANGLES, 32, 8, true, false, 256, true>(RT_ARGS)
                         : launch_bvh<KIND_BVH_TRIANGLES, 32, 8, false, false, 256, true>(RT_ARGS);
    }
    RT_BVH(KIND_BVH_TRIANGLES)
#undef RT_BVH
#undef RT_ARGS
}

hipError_t rt_launch_clear_pass(uint32_t *queue, unsigned long long *stats, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_clear_pass_kernel, dim3(1), dim3(64), 0, stream, queue, stats, (uint32_t)RT_STATS_WORDS);
    return hipGetLastError();
}

hipError_t rt_launch_reduce_chunks(float *accum_sum, const float *partial, uint32_t n_pixels, uint32_t n_chunks,
                                   size_t plane_stride, hipStream_t stream)
{
    const uint32_t n = n_pixels * 3u;
    if (n == 0 || n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_reduce_chunks_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream,
                       accum_sum, partial, n, n_chunks, plane_stride);
    return hipGetLastError();
}

hipError_t rt_launch_u32_add_planes(uint32_t *dst, const uint32_t *planes, uint32_t n, uint32_t n_planes,
                                    size_t plane_stride, hipStream_t stream)
{
    if (n == 0 || n_planes == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_u32_add_planes_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, dst, planes, n, n_planes, plane_stride);
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

hipError_t rt_launch_u8_sum(uint32_t *sum, const uint8_t *rgb8, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_u8_sum_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, sum, rgb8, n);
    return hipGetLastError();
}
hipError_t rt_launch_u8_mean(const uint32_t *sum, uint32_t n_images, uint8_t *rgb8, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_u8_mean_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, sum, n_images, rgb8, n);
    return hipGetLastError();
}

hipError_t rt_launch_selftest_cbrt_grid(float *out, unsigned long long *n_risky, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_selftest_cbrt_grid_kernel, dim3((1u << 24) / 256u), dim3(256), 0, stream, out, n_risky);
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
