/*
 * oracle/rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h for the parity status:
 * "parity unpinned" except the vector helpers, the sfc64 core and Philox).
 *
 * Plain-C restatement of the reference's per-pixel trace path.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/src).  All arithmetic is
 * IEEE binary32 with no contraction: build with -ffp-contract=off and without -ffast-math.
 * Zig's default float mode is strict, so a*b+c there is a rounded multiply then a rounded add.
 */
#include "rt_oracle.h"

/* Counters inside the intersection loops (sphere / box / triangle tests).  The parity build keeps them (they feed
 * tests/golden/opcounts_*.json); the build bench.py TIMES as the CPU baseline (make native: -O3 -march=native
 * -DORC_LEAN_COUNTERS) leaves them out so that the baseline is not slowed by bookkeeping.  Ray counts stay. */
#ifdef ORC_LEAN_COUNTERS
#define ORC_COUNT(x) ((void)0)
#else
#define ORC_COUNT(x) ((x)++)
#endif

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ====================================================================================
 * vector.zig:4-58
 * ==================================================================================== */
static inline orc_v3 v3(float x, float y, float z) { orc_v3 r = { x, y, z }; return r; }
static inline orc_v3 vadd(orc_v3 a, orc_v3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline orc_v3 vsub(orc_v3 a, orc_v3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline orc_v3 vmul(orc_v3 a, orc_v3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline orc_v3 vdiv(orc_v3 a, orc_v3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline orc_v3 vsplat(float s) { return v3(s, s, s); }
static inline orc_v3 vneg(orc_v3 a) { return v3(-a.x, -a.y, -a.z); }

/* vector.zig:6-8 sum = @reduce(.Add, v): LLVM's ordered reduction, ((v0 + v1) + v2) */
static inline float vsum(orc_v3 v) { return (v.x + v.y) + v.z; }
/* vector.zig:22-24 */
float orc_dot(orc_v3 a, orc_v3 b) { return vsum(vmul(a, b)); }
/* vector.zig:10-17 */
float orc_length(orc_v3 a) { return sqrtf(vsum(vmul(a, a))); }
/* vector.zig:18-20: v / splat(length(v)) -- three true divides */
orc_v3 orc_normalize(orc_v3 a) { return vdiv(a, vsplat(orc_length(a))); }
/* vector.zig:26-33 */
orc_v3 orc_cross(orc_v3 a, orc_v3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vector.zig:40-56 (ties go to the later axis) */
uint32_t orc_argmax(orc_v3 v)
{
    if (v.x > v.y) return (v.x > v.z) ? 0u : 2u;
    return (v.y > v.z) ? 1u : 2u;
}
static inline float vget(orc_v3 v, uint32_t ax) { return ax == 0 ? v.x : (ax == 1 ? v.y : v.z); }

/* ====================================================================================
 * RNG.  std.Random.Sfc64 (Zig std 0.14 lib/std/Random/Sfc64.zig; not in /root/reference):
 * call sites raytracer.zig:23,519,732.
 * ==================================================================================== */
uint64_t orc_sfc64_next(orc_sfc64 *s)
{
    uint64_t tmp = s->a + s->b + s->counter;
    s->counter += 1;
    s->a = s->b ^ (s->b >> 11);
    s->b = s->c + (s->c << 3);
    s->c = ((s->c << 24) | (s->c >> 40)) + tmp;
    return tmp;
}
void orc_sfc64_init(orc_sfc64 *s, uint64_t seed)
{
    s->a = seed; s->b = seed; s->c = seed; s->counter = 1;
    for (int i = 0; i < 12; i++) (void)orc_sfc64_next(s);
}
static inline int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
/* std.Random.float(f32) (Zig std 0.14 lib/std/Random.zig): 23 mantissa bits from the low end
 * of one u64, exponent from the leading-zero count of the same word. */
float orc_sfc64_float(orc_sfc64 *s)
{
    uint64_t r = orc_sfc64_next(s);
    uint32_t lz = (uint32_t)clz64(r);
    if (lz >= 41) {
        lz = 41 + (uint32_t)clz64(orc_sfc64_next(s));
        if (lz == 41 + 64) {
            /* r.int(u32) | 0b111: low four bytes of one more draw */
            uint32_t w = (uint32_t)orc_sfc64_next(s) | 7u;
            lz += (uint32_t)__builtin_clz(w);
        }
    }
    uint32_t mant = (uint32_t)r & 0x7FFFFFu;
    uint32_t bits = ((126u - lz) << 23) | mant;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* Philox4x32-10 (Salmon et al., Random123): the build-defined counter RNG. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
float orc_u32_to_unit_float(uint32_t x) { return (float)(x >> 8) * 0x1.0p-24f; }

/* One generator object for both modes.  Counter mode: the stream of a sample is addressed by
 * (pixel, global sample index, block); block 0 holds the two jitter draws (raytracer.zig:537-538),
 * block d+1 the draws of the scatter at recursion depth d (4 for lambertian/metal, 1 for
 * dielectric).  Draw k of a block is word k of Philox(ctr = {pixel, sample, block, 0}, key = seed). */
typedef struct {
    int       counter_mode;
    orc_sfc64 sfc;
    uint32_t  key[2];
    uint32_t  pixel, sample, block, idx;
    uint32_t  buf[4];
    int       have;
} rng_t;

static void rng_set_block(rng_t *g, uint32_t block)
{
    if (g->counter_mode) { g->block = block; g->idx = 0; g->have = 0; }
}
static float rng_float(rng_t *g)
{
    if (!g->counter_mode) return orc_sfc64_float(&g->sfc);
    if (!g->have) {
        uint32_t ctr[4] = { g->pixel, g->sample, g->block, 0u };
        orc_philox4x32_10(ctr, g->key, g->buf);
        g->have = 1;
    }
    if (g->idx >= 4) { fprintf(stderr, "rt_oracle: more than 4 draws in a block\n"); abort(); }
    return orc_u32_to_unit_float(g->buf[g->idx++]);
}

/* ====================================================================================
 * std.math.cbrt(f32) == musl cbrtf (FreeBSD msun s_cbrtf.c); call site raytracer.zig:577.
 * f64 arithmetic inside, two Newton steps, one final rounding to f32.
 * ==================================================================================== */
float orc_cbrtf(float x)
{
    const uint32_t B1 = 709958130u, B2 = 642849266u;
    uint32_t u; memcpy(&u, &x, 4);
    uint32_t hx = u & 0x7fffffffu;
    if (hx >= 0x7f800000u) return x + x;
    if (hx < 0x00800000u) {
        if (hx == 0) return x;
        float xs = x * 0x1p24f;
        memcpy(&u, &xs, 4);
        hx = u & 0x7fffffffu;
        hx = hx / 3 + B2;
    } else {
        hx = hx / 3 + B1;
    }
    u &= 0x80000000u; u |= hx;
    float tf; memcpy(&tf, &u, 4);
    double T = tf, r;
    r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    return (float)T;
}
void orc_cbrtf_array(const float *x, float *out, size_t n) { for (size_t k = 0; k < n; ++k) out[k] = orc_cbrtf(x[k]); }
/* std.math.pow(f32, x, 5) (raytracer.zig:467): Zig's pow squares-and-multiplies on the frexp
 * significand (a1 = m; m2 = m*m; m4 = m2*m2; a1 *= m4) -- scaling by powers of two is exact, so
 * away from overflow/underflow this equals x * ((x*x)*(x*x)) with a rounding after each product. */
float orc_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x * x4; }

/* ====================================================================================
 * ray.zig:10-12
 * ==================================================================================== */
static inline orc_v3 point_at(orc_v3 ro, orc_v3 rd, float t) { return vadd(ro, vmul(vsplat(t), rd)); }

/* ====================================================================================
 * camera.zig
 * ==================================================================================== */
void orc_camera_basic(orc_camera *cam)
{   /* camera.zig:10-13 */
    cam->lower_left = v3(-2.0f, -1.0f, -1.0f);
    cam->horizontal = v3(4.0f, 0.0f, 0.0f);
    cam->vertical = v3(0.0f, 2.0f, 0.0f);
    cam->origin = v3(0.0f, 0.0f, 0.0f);
    cam->subtract_origin = 0;
}
void orc_camera_set_position(orc_camera *cam, orc_v3 lookfrom, orc_v3 lookat, orc_v3 vup,
                             float vfov, float aspect)
{   /* camera.zig:25-37.  std.math.pi is a comptime_float: vfov * pi is folded to f32 * f32(pi). */
    cam->origin = lookfrom;
    const float theta = vfov * 3.14159265358979323846f / 180.0f;
    const float half_height = tanf(theta / 2);
    const float half_width = aspect * half_height;
    const orc_v3 w = orc_normalize(vsub(lookfrom, lookat));
    const orc_v3 u = orc_normalize(orc_cross(vup, w));
    const orc_v3 v = orc_cross(w, u);
    cam->lower_left = vsub(vsub(vsub(cam->origin, vmul(vsplat(half_width), u)),
                                vmul(vsplat(half_height), v)), w);
    cam->horizontal = vmul(vsplat(2 * half_width), u);
    cam->vertical = vmul(vsplat(2 * half_height), v);
    cam->subtract_origin = 1;
}
void orc_camera_get_ray(const orc_camera *cam, float s, float t, orc_v3 *ro, orc_v3 *rd)
{   /* camera.zig:15-18 (basic) / :38-41 (movable): left-to-right sums */
    orc_v3 d = vadd(vadd(cam->lower_left, vmul(vsplat(s), cam->horizontal)),
                    vmul(vsplat(t), cam->vertical));
    if (cam->subtract_origin) d = vsub(d, cam->origin);
    *ro = cam->origin;
    *rd = d;
}

/* ====================================================================================
 * geometry: raytracer.zig:31-115
 * ==================================================================================== */
typedef struct {
    float    t;
    orc_v3   p, normal;
    uint32_t prim;      /* index of the primitive hit (sphere index, or n_sph + triangle index) */
    uint32_t material;  /* index into the material table */
} hit_rec;

/* Sphere.hit, raytracer.zig:42-78 */
int orc_sphere_hit(const orc_sphere *s, orc_v3 ro, orc_v3 rd, float t_min, float t_max,
                   float *t, orc_v3 *p, orc_v3 *normal)
{
    const orc_v3 oc = vsub(ro, s->center);
    const float a = orc_dot(rd, rd);
    const float b = orc_dot(oc, rd);
    const float c = orc_dot(oc, oc) - s->radius * s->radius;
    const float discriminant = b * b - a * c;
    if (discriminant > 0) {
        const float temp = (-b - sqrtf(b * b - a * c)) / a;
        if ((temp < t_max) && (temp > t_min)) {
            const orc_v3 point = point_at(ro, rd, temp);
            *t = temp; *p = point;
            *normal = vdiv(vsub(point, s->center), vsplat(s->radius));
            return 1;
        }
        const float temp_n = (-b + sqrtf(b * b - a * c)) / a;
        if ((temp_n < t_max) && (temp_n > t_min)) {
            const orc_v3 point = point_at(ro, rd, temp_n);
            *t = temp_n; *p = point;
            *normal = vdiv(vsub(point, s->center), vsplat(s->radius));
            return 1;
        }
    }
    return 0;
}

/* Canonical vertex order of a triangle (part of the build-defined triangle test, applied by orc_world_create and by the
 * product's rt_upload_scene): the vertices are rotated cyclically -- winding and the side the flat normal points to are
 * kept -- so that v0 is the vertex OPPOSITE THE LONGEST EDGE, i.e. the corner with the largest angle.  The grazing guard
 * below bounds sin(corner angle at v0) * cos(theta); taken at whatever vertex the caller happened to list first it made
 * a needle triangle whose v0 is the sharp apex invisible from every direction and visible again when its vertices were
 * rotated.  With the largest corner (>= 60 degrees) at v0 the guard depends on the triangle's shape only:
 * sin >= 0.866 unless the largest angle exceeds 120 degrees; a triangle whose largest angle exceeds 179.55 degrees
 * (sin < 2^-7) is rejected from every direction by definition -- it is a line segment to 2^-7.
 * Squared edge lengths in f32 with the ordered dot; among equally long edges the one facing the lexicographically
 * smallest vertex (x, then y, then z) wins, so the result does not depend on the order the caller listed them in. */
static int v3_lex_less(orc_v3 a, orc_v3 b)
{
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    return a.z < b.z;
}
void orc_canonical_triangle(orc_triangle *tr)
{
    const orc_v3 v[3] = { tr->v0, tr->v1, tr->v2 };
    const orc_v3 a = vsub(v[1], v[0]), b = vsub(v[2], v[1]), c = vsub(v[0], v[2]);
    const float opp[3] = { orc_dot(b, b), orc_dot(c, c), orc_dot(a, a) };   /* squared edge facing v0, v1, v2 */
    int best = 0;
    for (int i = 1; i < 3; i++)
        if (opp[i] > opp[best] || (opp[i] == opp[best] && v3_lex_less(v[i], v[best]))) best = i;
    tr->v0 = v[best]; tr->v1 = v[(best + 1) % 3]; tr->v2 = v[(best + 2) % 3];
}

/* Build-defined ray-triangle test (Moller-Trumbore, two-sided).  The reference has none
 * (SURVEY 8(a) T1).  Flat normal = normalize(cross(v1-v0, v2-v0)): outward for the
 * par_shapes winding, inward when the owning sphere's radius is negative.
 * Grazing guard (part of the definition): det = |d| |e1| |e2| sin(phi) cos(theta) (phi the corner angle at v0, theta the
 * angle between ray and normal), and the barycentrics' rounding error, ~7 ulps of |o - v0|, is divided by that product:
 * without a floor on it the test accepts rays that pass arbitrarily far from the triangle when the origin is far away,
 * and no culling structure could then be conservative.  A ray with sin(phi) cos(theta) < 2^-7 (0.45 degrees off the
 * triangle's plane for a right-angled corner) misses by definition; what is accepted then lies within 7 * 2^-17 |o - v0|
 * of the triangle, inside the cone the cull tests use (2^-13). */
int orc_triangle_hit(const orc_triangle *tr, orc_v3 ro, orc_v3 rd, float t_min, float t_max,
                     float *t, orc_v3 *p, orc_v3 *normal)
{
    const orc_v3 e1 = vsub(tr->v1, tr->v0);
    const orc_v3 e2 = vsub(tr->v2, tr->v0);
    const orc_v3 pvec = orc_cross(rd, e2);
    const float det = orc_dot(e1, pvec);
    if (!(det != 0.0f)) return 0;               /* parallel (or NaN) */
    if (!(det * det >= ((orc_dot(rd, rd) * orc_dot(e1, e1)) * orc_dot(e2, e2)) * 0x1.0p-14f)) return 0;   /* grazing */
    const float inv_det = 1.0f / det;
    const orc_v3 tvec = vsub(ro, tr->v0);
    const float u = orc_dot(tvec, pvec) * inv_det;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    const orc_v3 qvec = orc_cross(tvec, e1);
    const float v = orc_dot(rd, qvec) * inv_det;
    if (!(v >= 0.0f && (u + v) <= 1.0f)) return 0;
    const float tt = orc_dot(e2, qvec) * inv_det;
    if ((tt < t_max) && (tt > t_min)) {
        *t = tt;
        *p = point_at(ro, rd, tt);
        *normal = orc_normalize(orc_cross(e1, e2));
        return 1;
    }
    return 0;
}

/* ---- reference BVH: raytracer.zig:117-395 ------------------------------------------ */
typedef struct { float min, max; } interval;
typedef struct { interval x, y, z; } aabb;

/* AxisAlignedBBox.make, raytracer.zig:126-144 */
static aabb aabb_make(orc_v3 a, orc_v3 b)
{
    aabb r;
    if (a.x <= b.x) { r.x.min = a.x; r.x.max = b.x; } else { r.x.min = b.x; r.x.max = a.x; }
    if (a.y <= b.y) { r.y.min = a.y; r.y.max = b.y; } else { r.y.min = b.y; r.y.max = a.y; }
    if (a.z <= b.z) { r.z.min = a.z; r.z.max = b.z; } else { r.z.min = b.z; r.z.max = a.z; }
    return r;
}
/* AxisAlignedBBox.hit, raytracer.zig:154-182 -- axes 0 and 1 only (quirk Q1, `for (0..2)`) */
static int aabb_hit(const aabb *bb, orc_v3 ro, orc_v3 rd, float t_min, float t_max)
{
    interval ray_t = { t_min, t_max };
    for (int axis = 0; axis < 2; axis++) {
        const interval ax = axis == 0 ? bb->x : bb->y;
        const float d = axis == 0 ? rd.x : rd.y;
        const float o = axis == 0 ? ro.x : ro.y;
        const float adinv = 1.0f / d;
        const float t0 = (ax.min - o) * adinv;
        const float t1 = (ax.max - o) * adinv;
        if (t0 < t1) {
            if (t0 > ray_t.min) ray_t.min = t0;
            if (t1 < ray_t.max) ray_t.max = t1;
        } else {
            if (t1 > ray_t.min) ray_t.min = t1;
            if (t0 < ray_t.max) ray_t.max = t0;
        }
        if (ray_t.max <= ray_t.min) return 0;
    }
    return 1;
}

typedef struct ref_node {
    int is_leaf;
    aabb bbox;
    struct ref_node *child1, *child2;   /* branch: raytracer.zig:204-207 */
    uint32_t *inds; uint32_t n_inds;    /* leaf:   raytracer.zig:237-240 */
} ref_node;

/* std.sort.heap (Zig std 0.14 lib/std/sort.zig heapContext/siftDown), call site raytracer.zig:331.
 * Unstable; restated so that equal keys land where Zig's would (unverifiable here). */
typedef struct { uint32_t ind; orc_sphere obj; } obj_ind;
static int obj_less(uint32_t ax, const obj_ind *a, const obj_ind *b)
{   /* Sphere.lessThan, raytracer.zig:79-85 */
    return (vget(a->obj.center, ax) - a->obj.radius) < (vget(b->obj.center, ax) - b->obj.radius);
}
static void obj_swap(obj_ind *a, obj_ind *b) { obj_ind t = *a; *a = *b; *b = t; }
static void sift_down(obj_ind *it, size_t a, size_t target, size_t b, uint32_t ax)
{
    size_t cur = target;
    for (;;) {
        size_t child = (cur - a) * 2 + a + 1;
        if (!(child < b)) break;
        size_t next_child = child + 1;
        if (next_child < b && obj_less(ax, &it[child], &it[next_child])) child = next_child;
        if (obj_less(ax, &it[child], &it[cur])) break;
        obj_swap(&it[child], &it[cur]);
        cur = child;
    }
}
static void heap_sort(obj_ind *it, size_t n, uint32_t ax)
{
    size_t a = 0, b = n;
    size_t i = a + (b - a) / 2;
    while (i > a) { i -= 1; sift_down(it, a, i, b, ax); }
    i = b;
    while (i > a) { i -= 1; obj_swap(&it[a], &it[i]); sift_down(it, a, a, i, ax); }
}

/* build_BVH, raytracer.zig:279-376 (debug prints omitted) */
static ref_node *build_bvh(const orc_sphere *parent, const uint32_t *obj_inds, uint32_t len,
                           uint32_t depth, uint32_t *node_count)
{
    *node_count += 1;                                           /* :280 */
    obj_ind *objects = (obj_ind *)malloc(sizeof(obj_ind) * (len ? len : 1));
    for (uint32_t i = 0; i < len; i++) { objects[i].ind = obj_inds[i]; objects[i].obj = parent[obj_inds[i]]; }

    orc_v3 min_corner = v3(99999.0f, 99999.0f, 99999.0f);       /* :287-298 */
    orc_v3 max_corner = v3(-99999.0f, -99999.0f, -99999.0f);
    for (uint32_t k = 0; k < len; k++) {
        const orc_sphere *o = &objects[k].obj;
        if ((o->center.x - o->radius) < min_corner.x) min_corner.x = o->center.x - o->radius;
        if ((o->center.x + o->radius) > max_corner.x) max_corner.x = o->center.x + o->radius;
        if ((o->center.y - o->radius) < min_corner.y) min_corner.y = o->center.y - o->radius;
        if ((o->center.y + o->radius) > max_corner.y) max_corner.y = o->center.y + o->radius;
        if ((o->center.z - o->radius) < min_corner.z) min_corner.z = o->center.z - o->radius;
        if ((o->center.z + o->radius) > max_corner.z) max_corner.z = o->center.z + o->radius;
    }
    const aabb new_bbox = aabb_make(min_corner, max_corner);    /* :300 */
    ref_node *node = (ref_node *)calloc(1, sizeof(ref_node));
    node->bbox = new_bbox;
    if ((depth == 0) || (len <= 3)) {                           /* :309-326 */
        node->is_leaf = 1;
        node->inds = (uint32_t *)malloc(sizeof(uint32_t) * (len ? len : 1));
        memcpy(node->inds, obj_inds, sizeof(uint32_t) * len);
        node->n_inds = len;
        free(objects);
        return node;
    }
    const orc_v3 dimensions = vsub(max_corner, min_corner);     /* :329-331 */
    const uint32_t biggest_ax = orc_argmax(dimensions);
    heap_sort(objects, len, biggest_ax);
    uint32_t *sorted_inds = (uint32_t *)malloc(sizeof(uint32_t) * len);
    for (uint32_t k = 0; k < len; k++) sorted_inds[k] = objects[k].ind;
    const float thresh = vget(dimensions, biggest_ax) / 2 + vget(min_corner, biggest_ax);  /* :337 */
    uint32_t thresh_idx = 1;                                    /* :339-346 */
    for (uint32_t k = 0; k < len; k++) {
        const float point = vget(objects[k].obj.center, biggest_ax) + objects[k].obj.radius;
        if (point > thresh) break;
        thresh_idx += 1;
    }
    free(objects);
    if (thresh_idx > len - 1) thresh_idx = len - 1;             /* :359 */
    const uint32_t split = thresh_idx;                          /* :364-366 */
    node->child1 = build_bvh(parent, sorted_inds, split, depth - 1, node_count);             /* :371 */
    node->child2 = build_bvh(parent, sorted_inds + split, len - split, depth - 1, node_count); /* :372 */
    free(sorted_inds);
    return node;
}
static void free_bvh(ref_node *n)
{
    if (!n) return;
    free_bvh(n->child1); free_bvh(n->child2);
    free(n->inds); free(n);
}

/* ---- conservative median-split BVH over triangles (oracle speed-up only; closest hit is
 * the lexicographic minimum of (t, index), i.e. exactly the list answer) -------------- */
typedef struct { float lo[3], hi[3]; uint32_t left, right, first, count; } tri_node;

struct orc_world {
    orc_sphere   *sph;  uint32_t n_sph;
    orc_triangle *tri;  uint32_t n_tri;
    orc_sphere   *mat;  uint32_t n_mat;   /* material table for triangles */
    ref_node     *bvh;  uint32_t bvh_nodes;
    tri_node     *tnodes; uint32_t n_tnodes; uint32_t *tri_order;
};

static void tri_bounds(const orc_triangle *t, float lo[3], float hi[3])
{
    const float xs[3] = { t->v0.x, t->v1.x, t->v2.x }, ys[3] = { t->v0.y, t->v1.y, t->v2.y },
                zs[3] = { t->v0.z, t->v1.z, t->v2.z };
    lo[0] = fminf(xs[0], fminf(xs[1], xs[2])); hi[0] = fmaxf(xs[0], fmaxf(xs[1], xs[2]));
    lo[1] = fminf(ys[0], fminf(ys[1], ys[2])); hi[1] = fmaxf(ys[0], fmaxf(ys[1], ys[2]));
    lo[2] = fminf(zs[0], fminf(zs[1], zs[2])); hi[2] = fmaxf(zs[0], fmaxf(zs[1], zs[2]));
}
typedef struct { const orc_triangle *tri; int axis; } tri_sort_ctx;
static tri_sort_ctx g_tsc;  /* qsort has no context argument; world creation is single-threaded */
static int tri_cmp(const void *pa, const void *pb)
{
    const orc_triangle *a = &g_tsc.tri[*(const uint32_t *)pa], *b = &g_tsc.tri[*(const uint32_t *)pb];
    float ca, cb;
    if (g_tsc.axis == 0) { ca = a->v0.x + a->v1.x + a->v2.x; cb = b->v0.x + b->v1.x + b->v2.x; }
    else if (g_tsc.axis == 1) { ca = a->v0.y + a->v1.y + a->v2.y; cb = b->v0.y + b->v1.y + b->v2.y; }
    else { ca = a->v0.z + a->v1.z + a->v2.z; cb = b->v0.z + b->v1.z + b->v2.z; }
    return (ca > cb) - (ca < cb);
}
static uint32_t tri_build(orc_world *w, uint32_t first, uint32_t count)
{
    uint32_t id = w->n_tnodes++;
    tri_node *n = &w->tnodes[id];
    for (int k = 0; k < 3; k++) { n->lo[k] = FLT_MAX; n->hi[k] = -FLT_MAX; }
    for (uint32_t i = 0; i < count; i++) {
        float lo[3], hi[3];
        tri_bounds(&w->tri[w->tri_order[first + i]], lo, hi);
        for (int k = 0; k < 3; k++) { n->lo[k] = fminf(n->lo[k], lo[k]); n->hi[k] = fmaxf(n->hi[k], hi[k]); }
    }
    /* pad so that the slab test below can never cull a triangle its exact test accepts */
    for (int k = 0; k < 3; k++) {
        float pad = 1e-4f * fmaxf(1.0f, fmaxf(fabsf(n->lo[k]), fabsf(n->hi[k])));
        n->lo[k] -= pad; n->hi[k] += pad;
    }
    n->first = first; n->count = count; n->left = n->right = 0;
    if (count <= 4) return id;
    int axis = 0; float ext = n->hi[0] - n->lo[0];
    if (n->hi[1] - n->lo[1] > ext) { axis = 1; ext = n->hi[1] - n->lo[1]; }
    if (n->hi[2] - n->lo[2] > ext) { axis = 2; }
    g_tsc.tri = w->tri; g_tsc.axis = axis;
    qsort(w->tri_order + first, count, sizeof(uint32_t), tri_cmp);
    uint32_t half = count / 2;
    uint32_t l = tri_build(w, first, half);
    uint32_t r = tri_build(w, first + half, count - half);
    n = &w->tnodes[id];
    n->left = l; n->right = r; n->count = 0;
    return id;
}
/* Cull test of the checker's triangle BVH.  It must never reject a box that holds a triangle orc_triangle_hit would
 * accept, for any ray origin: Moller-Trumbore's barycentrics are off by a few ulps of |o - v0| ACROSS the ray, and
 * the slab arithmetic rounds relative to the distance.  So the ray is widened into a cone of half-angle 2^-13
 * (f64 arithmetic, plain formulas; speed does not matter here): at parameter t the box is grown by g t, g = 2^-13 |d|
 * (orc_triangle_hit's grazing guard bounds what the test can accept to 7 * 2^-17 |o - v0| around the triangle).
 * Axis k: o_k + t (d_k + g) >= lo_k and o_k + t (d_k - g) <= hi_k.  The nodes' static padding (tri_build) covers
 * origins next to a triangle. */
static int tri_box_hit(const tri_node *n, orc_v3 ro, orc_v3 rd, float t_min, float t_max)
{
    const double o[3] = { ro.x, ro.y, ro.z }, d[3] = { rd.x, rd.y, rd.z };
    const double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double g = len * (1.0 / 8192.0);
    double lo = t_min, hi = t_max;
    if (!(len > 0.0) || !(len < 1e300)) return 1;          /* degenerate ray: no culling (the exact tests reject it) */
    for (int k = 0; k < 3; k++) {
        /* upper edge of the cone, slope d_k + g, must reach lo_k; lower edge, slope d_k - g, must stay below hi_k */
        const double up = d[k] + g, dn = d[k] - g;
        const double a = (double)n->lo[k] - o[k], b = (double)n->hi[k] - o[k];
        /* t * up >= a */
        if (up > 0) { if (a / up > lo) lo = a / up; }
        else if (up < 0) { if (a / up < hi) hi = a / up; }
        else if (a > 0) return 0;
        /* t * dn <= b */
        if (dn > 0) { if (b / dn < hi) hi = b / dn; }
        else if (dn < 0) { if (b / dn > lo) lo = b / dn; }
        else if (b < 0) return 0;
    }
    return !(hi < lo - 1e-9 * fabs(lo));
}

orc_world *orc_world_create(const orc_sphere *sph, uint32_t n_sph,
                            const orc_triangle *tri, uint32_t n_tri,
                            const orc_sphere *materials, uint32_t n_mat)
{
    orc_world *w = (orc_world *)calloc(1, sizeof(orc_world));
    if (!w) return NULL;
    if (n_sph) {
        w->sph = (orc_sphere *)malloc(sizeof(orc_sphere) * n_sph);
        memcpy(w->sph, sph, sizeof(orc_sphere) * n_sph);
        w->n_sph = n_sph;
        /* raytracer.zig:693-703: identity index list, build_BVH(depth = 5) */
        uint32_t *inds = (uint32_t *)malloc(sizeof(uint32_t) * n_sph);
        for (uint32_t i = 0; i < n_sph; i++) inds[i] = i;
        w->bvh = build_bvh(w->sph, inds, n_sph, 5, &w->bvh_nodes);
        free(inds);
    }
    if (n_mat) {
        w->mat = (orc_sphere *)malloc(sizeof(orc_sphere) * n_mat);
        memcpy(w->mat, materials, sizeof(orc_sphere) * n_mat);
        w->n_mat = n_mat;
    }
    if (n_tri) {
        w->tri = (orc_triangle *)malloc(sizeof(orc_triangle) * n_tri);
        memcpy(w->tri, tri, sizeof(orc_triangle) * n_tri);
        w->n_tri = n_tri;
        for (uint32_t i = 0; i < n_tri; i++) orc_canonical_triangle(&w->tri[i]);
        for (uint32_t i = 0; i < n_tri; i++)
            if (w->tri[i].material >= n_mat) { orc_world_destroy(w); return NULL; }
        w->tri_order = (uint32_t *)malloc(sizeof(uint32_t) * n_tri);
        for (uint32_t i = 0; i < n_tri; i++) w->tri_order[i] = i;
        w->tnodes = (tri_node *)malloc(sizeof(tri_node) * (2 * (size_t)n_tri + 1));
        tri_build(w, 0, n_tri);
    }
    return w;
}
void orc_world_destroy(orc_world *w)
{
    if (!w) return;
    free_bvh(w->bvh);
    free(w->sph); free(w->tri); free(w->mat); free(w->tnodes); free(w->tri_order);
    free(w);
}
uint32_t orc_world_bvh_nodes(const orc_world *w) { return w->bvh_nodes; }

/* ---- closest hit ------------------------------------------------------------------- */
typedef struct {
    const orc_world *w;
    uint32_t traversal;
    orc_counters *cnt;
} hit_ctx;

static int sphere_hit_rec(const hit_ctx *c, uint32_t idx, orc_v3 ro, orc_v3 rd, float t_min, float t_max, hit_rec *rec)
{
    ORC_COUNT(c->cnt->sphere_tests);
    if (orc_sphere_hit(&c->w->sph[idx], ro, rd, t_min, t_max, &rec->t, &rec->p, &rec->normal)) {
        rec->prim = idx; rec->material = idx;
        return 1;
    }
    return 0;
}
/* hittableList.hit, raytracer.zig:92-113 */
static int list_hit(const hit_ctx *c, orc_v3 ro, orc_v3 rd, float t_min, float t_max, hit_rec *out)
{
    hit_rec temp_rec; int hit_anything = 0; float closest_so_far = t_max;
    for (uint32_t i = 0; i < c->w->n_sph; i++) {
        hit_rec rec;
        if (sphere_hit_rec(c, i, ro, rd, t_min, closest_so_far, &rec)) {
            hit_anything = 1; temp_rec = rec; closest_so_far = rec.t;
        }
    }
    if (hit_anything) *out = temp_rec;
    return hit_anything;
}
/* TreeNode.hit / BranchNode.hit / LeafNode.hit, raytracer.zig:191-267 */
static int node_hit(const hit_ctx *c, const ref_node *n, orc_v3 ro, orc_v3 rd, float t_min, float t_max, hit_rec *out)
{
    ORC_COUNT(c->cnt->aabb_tests);
    if (!aabb_hit(&n->bbox, ro, rd, t_min, t_max)) return 0;
    if (n->is_leaf) {
        hit_rec temp_rec; int hit_anything = 0; float closest_so_far = t_max;
        for (uint32_t k = 0; k < n->n_inds; k++) {
            hit_rec rec;
            if (sphere_hit_rec(c, n->inds[k], ro, rd, t_min, closest_so_far, &rec)) {
                hit_anything = 1; temp_rec = rec; closest_so_far = rec.t;
            }
        }
        if (hit_anything) *out = temp_rec;
        return hit_anything;
    }
    hit_rec rec1, rec2;
    const int h1 = node_hit(c, n->child1, ro, rd, t_min, t_max, &rec1);   /* both children, same t_max */
    const int h2 = node_hit(c, n->child2, ro, rd, t_min, t_max, &rec2);
    if (h1 && h2) { *out = (rec1.t < rec2.t) ? rec1 : rec2; return 1; }
    if (h1) { *out = rec1; return 1; }
    if (h2) { *out = rec2; return 1; }
    return 0;
}
static void tri_traverse(const hit_ctx *c, orc_v3 ro, orc_v3 rd, float t_min, hit_rec *best, int *have)
{
    const orc_world *w = c->w;
    uint32_t stack[64]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const tri_node *n = &w->tnodes[stack[--sp]];
        /* allow ties through: a later-visited, lower-index triangle at equal t must still be seen */
        float tmax = *have ? best->t : FLT_MAX;
        ORC_COUNT(c->cnt->aabb_tests);
        if (!tri_box_hit(n, ro, rd, t_min, tmax)) continue;
        if (n->count) {
            for (uint32_t k = 0; k < n->count; k++) {
                uint32_t ti = w->tri_order[n->first + k];
                hit_rec rec;
                ORC_COUNT(c->cnt->tri_tests);
                if (orc_triangle_hit(&w->tri[ti], ro, rd, t_min, FLT_MAX, &rec.t, &rec.p, &rec.normal)) {
                    rec.prim = w->n_sph + ti; rec.material = w->tri[ti].material;
                    if (!*have || rec.t < best->t || (rec.t == best->t && rec.prim < best->prim)) {
                        *best = rec; *have = 1;
                    }
                }
            }
        } else {
            if (sp + 2 > 64) { fprintf(stderr, "rt_oracle: triangle BVH stack overflow\n"); abort(); }
            stack[sp++] = n->left; stack[sp++] = n->right;
        }
    }
}
/* world.hit: HittableBVH.hit raytracer.zig:382-393 or the list; triangles (build-defined) are
 * tested after the spheres and win only with a strictly smaller t. */
static int world_hit(const hit_ctx *c, orc_v3 ro, orc_v3 rd, float t_min, float t_max, hit_rec *out)
{
    int have = 0;
    hit_rec best;
    if (c->w->n_sph) {
        if (c->traversal == ORC_TRAV_BVH_REF) have = node_hit(c, c->w->bvh, ro, rd, t_min, t_max, &best);
        else have = list_hit(c, ro, rd, t_min, t_max, &best);
    }
    if (c->w->n_tri && c->traversal == ORC_TRAV_BRUTE) {
        for (uint32_t ti = 0; ti < c->w->n_tri; ti++) {       /* every triangle, index order: the definition */
            hit_rec rec;
            ORC_COUNT(c->cnt->tri_tests);
            if (orc_triangle_hit(&c->w->tri[ti], ro, rd, t_min, FLT_MAX, &rec.t, &rec.p, &rec.normal)) {
                rec.prim = c->w->n_sph + ti; rec.material = c->w->tri[ti].material;
                if (!have || rec.t < best.t) { best = rec; have = 1; }
            }
        }
    } else if (c->w->n_tri) tri_traverse(c, ro, rd, t_min, &best, &have);
    if (have) *out = best;
    return have;
}

/* ====================================================================================
 * materials: raytracer.zig:397-500
 * ==================================================================================== */
/* Metal.reflect / Dielectric.reflect, raytracer.zig:420-422, :450-452: splat(2)*splat(dot)*n */
orc_v3 orc_reflect(orc_v3 v, orc_v3 n)
{
    return vsub(v, vmul(vmul(vsplat(2.0f), vsplat(orc_dot(v, n))), n));
}
/* Dielectric.refract, raytracer.zig:453-463 */
int orc_refract(orc_v3 v, orc_v3 n, float ni_nt, orc_v3 *out)
{
    const orc_v3 uv = orc_normalize(v);
    const float dt = orc_dot(uv, n);
    const float discriminant = 1.0f - ni_nt * ni_nt * (1.0f - (dt * dt));
    if (discriminant > 0) {
        *out = vsub(vmul(vsplat(ni_nt), vsub(uv, vmul(n, vsplat(dt)))), vmul(n, vsplat(sqrtf(discriminant))));
        return 1;
    }
    return 0;
}
/* Dielectric.schlick, raytracer.zig:464-468 */
float orc_schlick(float cosine, float idx)
{
    float r0 = (1 - idx) / (1 + idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * orc_pow5(1 - cosine);
}
/* random_in_unit_sphere, raytracer.zig:574-578: draws x, y, z then the cbrt argument */
static orc_v3 random_in_unit_sphere(rng_t *g)
{
    const float x = rng_float(g), y = rng_float(g), z = rng_float(g);
    const orc_v3 unit_cube = v3(x, y, z);
    const orc_v3 unit_sphere = vsub(vmul(vsplat(2.0f), orc_normalize(unit_cube)), vsplat(1.0f));
    return vmul(unit_sphere, vsplat(orc_cbrtf(rng_float(g))));
}
/* Material.scatter and the three scatter functions, raytracer.zig:407-414, :423-433, :438-445,
 * :469-499.  Returns 0 when the ray is absorbed. */
static int scatter(const orc_sphere *m, orc_v3 rd, const hit_rec *rec, rng_t *g,
                   orc_v3 *atten, orc_v3 *so, orc_v3 *sd, orc_counters *cnt)
{
    switch (m->kind) {
    case ORC_METAL: {
        cnt->scatter_metal++;
        const orc_v3 reflected = orc_reflect(orc_normalize(rd), rec->normal);
        const orc_v3 dir = vadd(reflected, vmul(vsplat(m->param), random_in_unit_sphere(g)));
        if (orc_dot(dir, rec->normal) > 0) { *atten = m->albedo; *so = rec->p; *sd = dir; return 1; }
        cnt->absorbed++;
        return 0;
    }
    case ORC_LAMBERTIAN: {
        cnt->scatter_lambertian++;
        const orc_v3 target = vadd(vadd(rec->p, rec->normal), random_in_unit_sphere(g));
        *atten = m->albedo; *so = rec->p; *sd = vsub(target, rec->p);
        return 1;
    }
    default: {
        cnt->scatter_dielectric++;
        orc_v3 outward_normal; float ni_over_nt, cosine, reflect_prob;
        const orc_v3 reflected = orc_reflect(orc_normalize(rd), rec->normal);
        if (orc_dot(rd, rec->normal) > 0) {
            outward_normal = vneg(rec->normal);
            ni_over_nt = m->param;
            cosine = m->param * orc_dot(rd, rec->normal) / orc_length(rd);
        } else {
            outward_normal = rec->normal;
            ni_over_nt = 1.0f / m->param;
            cosine = -orc_dot(rd, rec->normal) / orc_length(rd);
        }
        orc_v3 refracted = v3(0, 0, 0);
        const int has_refr = orc_refract(rd, outward_normal, ni_over_nt, &refracted);
        reflect_prob = has_refr ? orc_schlick(cosine, m->param) : 1.0f;
        *atten = m->albedo; *so = rec->p;
        if (rng_float(g) < reflect_prob) *sd = reflected; else *sd = refracted;
        return 1;
    }
    }
}

/* sky, raytracer.zig:569-571 */
orc_v3 orc_sky(orc_v3 dir)
{
    const orc_v3 unit_dir = orc_normalize(dir);
    const float t = 0.5f * (unit_dir.y + 1.0f);
    return vadd(vmul(vsplat(1.0f - t), vsplat(1.0f)), vmul(vsplat(t), v3(0.5f, 0.7f, 1.0f)));
}

typedef struct {
    hit_ctx hc;
    float t_min;
    uint32_t max_depth;
    uint32_t rays_this_pixel;
} trace_ctx;

/* color, raytracer.zig:554-572 */
static orc_v3 color(trace_ctx *tc, orc_v3 ro, orc_v3 rd, uint32_t depth, rng_t *g)
{
    orc_counters *cnt = tc->hc.cnt;
    hit_rec rec;
    cnt->rays++; tc->rays_this_pixel++;
    if (depth < 40) cnt->rays_by_depth[depth]++;
    if (world_hit(&tc->hc, ro, rd, tc->t_min, FLT_MAX, &rec)) {
        cnt->hits++;
        if (depth > tc->max_depth) { cnt->depth_exceeded++; return vsplat(0.0f); }
        const orc_world *w = tc->hc.w;
        const orc_sphere *m = rec.prim < w->n_sph ? &w->sph[rec.material] : &w->mat[rec.material];
        orc_v3 atten, so, sd;
        rng_set_block(g, depth + 1);
        if (scatter(m, rd, &rec, g, &atten, &so, &sd, cnt)) {
            const orc_v3 c = color(tc, so, sd, depth + 1, g);
            return vmul(atten, c);
        }
        return vsplat(0.0f);
    }
    cnt->misses++;
    return orc_sky(rd);
}

/* one sample of trace's loop body, raytracer.zig:537-540 */
static orc_v3 trace_sample(trace_ctx *tc, const orc_camera *cam, uint32_t i, uint32_t j,
                           uint32_t nx, uint32_t ny, rng_t *g)
{
    rng_set_block(g, 0);
    const float u = ((float)i + rng_float(g)) / (float)nx;
    const float v = ((float)j + rng_float(g)) / (float)ny;
    orc_v3 ro, rd;
    orc_camera_get_ray(cam, u, v, &ro, &rd);
    tc->hc.cnt->primary_rays++;
    return color(tc, ro, rd, 0, g);
}

static void quantise(orc_v3 mean, uint8_t *rgb)
{   /* trace's sqrt gamma (raytracer.zig:545-549) then Render's 255.99 truncation (:527) */
    const float c[3] = { mean.x, mean.y, mean.z };
    for (int k = 0; k < 3; k++) {
        float q = 255.99f * sqrtf(c[k]);
        int iv = (int)q;              /* truncation toward zero */
        if (iv < 0) iv = 0;           /* @intFromFloat is undefined out of range; clamp (documented) */
        if (iv > 255) iv = 255;
        rgb[k] = (uint8_t)iv;
    }
}

static void counters_add(orc_counters *d, const orc_counters *s)
{
    uint64_t *dp = (uint64_t *)d; const uint64_t *sp = (const uint64_t *)s;
    for (size_t k = 0; k < sizeof(orc_counters) / sizeof(uint64_t); k++) dp[k] += sp[k];
}

/* ---- counter-RNG render ---------------------------------------------------------------- */
typedef struct {
    const orc_world *w; const orc_camera *cam; const orc_config *cfg;
    const uint32_t *pixels; uint32_t n;
    float *accum_mean; uint8_t *rgb8; uint32_t *raycount;
    uint32_t tid, nthreads;
    orc_counters cnt;
} job_t;

static void *counter_worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    const orc_config *cfg = jb->cfg;
    trace_ctx tc;
    tc.hc.w = jb->w; tc.hc.traversal = cfg->traversal; tc.hc.cnt = &jb->cnt;
    tc.t_min = cfg->t_min; tc.max_depth = cfg->max_depth;
    rng_t g; memset(&g, 0, sizeof g);
    g.counter_mode = 1;
    g.key[0] = (uint32_t)cfg->seed; g.key[1] = (uint32_t)(cfg->seed >> 32);
    for (uint32_t k = jb->tid; k < jb->n; k += jb->nthreads) {
        const uint32_t pix = jb->pixels ? jb->pixels[k] : k;
        const uint32_t jr = pix / cfg->nx, i = pix % cfg->nx;           /* out index jr*nx+i, raytracer.zig:521-528 */
        const uint32_t j = cfg->row_offset_q3 ? (cfg->ny - jr) : (cfg->ny - 1 - jr);   /* :523 */
        g.pixel = pix;
        tc.rays_this_pixel = 0;
        orc_v3 col = v3(0, 0, 0);                                        /* trace: raytracer.zig:535-541 */
        if (cfg->spp_chunk == 0) {
            for (uint32_t s = cfg->spp_begin; s < cfg->spp_end; s++) {
                g.sample = s;
                col = vadd(col, trace_sample(&tc, jb->cam, i, j, cfg->nx, cfg->ny, &g));
            }
        } else {
            uint32_t s = cfg->spp_begin;
            while (s < cfg->spp_end) {
                uint32_t e = (s / cfg->spp_chunk + 1) * cfg->spp_chunk;
                if (e > cfg->spp_end) e = cfg->spp_end;
                orc_v3 part = v3(0, 0, 0);
                for (; s < e; s++) {
                    g.sample = s;
                    part = vadd(part, trace_sample(&tc, jb->cam, i, j, cfg->nx, cfg->ny, &g));
                }
                col = vadd(col, part);
            }
        }
        const orc_v3 mean = vdiv(col, vsplat((float)cfg->spp_total));   /* :542 */
        if (jb->accum_mean) { jb->accum_mean[3 * k] = mean.x; jb->accum_mean[3 * k + 1] = mean.y; jb->accum_mean[3 * k + 2] = mean.z; }
        if (jb->rgb8) quantise(mean, jb->rgb8 + 3 * (size_t)k);
        if (jb->raycount) jb->raycount[k] = tc.rays_this_pixel;
    }
    return NULL;
}

int orc_render_counter(const orc_world *w, const orc_camera *cam, const orc_config *cfg,
                       const uint32_t *pixels, uint32_t n_pixels,
                       float *accum_mean, uint8_t *rgb8, uint32_t *raycount,
                       orc_counters *counters)
{
    if (!w || !cam || !cfg || cfg->nx == 0 || cfg->ny == 0 || cfg->spp_end < cfg->spp_begin || cfg->spp_total == 0)
        return -1;
    const uint32_t n = pixels ? n_pixels : cfg->nx * cfg->ny;
    if (pixels) for (uint32_t k = 0; k < n; k++) if (pixels[k] >= cfg->nx * cfg->ny) return -1;
    uint32_t nt = cfg->threads ? cfg->threads : 1;
    if (nt > 256) nt = 256;
    job_t *jobs = (job_t *)calloc(nt, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc(nt, sizeof(pthread_t));
    for (uint32_t t = 0; t < nt; t++) {
        jobs[t].w = w; jobs[t].cam = cam; jobs[t].cfg = cfg; jobs[t].pixels = pixels; jobs[t].n = n;
        jobs[t].accum_mean = accum_mean; jobs[t].rgb8 = rgb8; jobs[t].raycount = raycount;
        jobs[t].tid = t; jobs[t].nthreads = nt;
        if (nt > 1) pthread_create(&th[t], NULL, counter_worker, &jobs[t]);
    }
    if (nt == 1) counter_worker(&jobs[0]);
    else for (uint32_t t = 0; t < nt; t++) pthread_join(th[t], NULL);
    if (counters) { memset(counters, 0, sizeof *counters); for (uint32_t t = 0; t < nt; t++) counters_add(counters, &jobs[t].cnt); }
    free(jobs); free(th);
    return 0;
}

/* ---- reference-style render: Render + render_and_write, raytracer.zig:517-532, :580-618 ---- */
typedef struct {
    const orc_world *w; const orc_camera *cam; const orc_config *cfg;
    uint32_t spp; uint64_t seed; uint8_t *img;
    orc_counters cnt;
} ref_job_t;

static void *reference_worker(void *arg)
{   /* Render, raytracer.zig:517-532: one sequential sfc64 stream for the whole image */
    ref_job_t *jb = (ref_job_t *)arg;
    const orc_config *cfg = jb->cfg;
    trace_ctx tc;
    tc.hc.w = jb->w; tc.hc.traversal = cfg->traversal; tc.hc.cnt = &jb->cnt;
    tc.t_min = cfg->t_min; tc.max_depth = cfg->max_depth;
    rng_t g; memset(&g, 0, sizeof g);
    g.counter_mode = 0;
    orc_sfc64_init(&g.sfc, jb->seed);
    for (uint32_t jr = 0; jr < cfg->ny; jr++) {
        for (uint32_t i = 0; i < cfg->nx; i++) {
            const uint32_t j = cfg->row_offset_q3 ? (cfg->ny - jr) : (cfg->ny - 1 - jr);
            orc_v3 col = v3(0, 0, 0);
            for (uint32_t s = 0; s < jb->spp; s++)
                col = vadd(col, trace_sample(&tc, jb->cam, i, j, cfg->nx, cfg->ny, &g));
            const orc_v3 mean = vdiv(col, vsplat((float)jb->spp));
            quantise(mean, jb->img + 3 * ((size_t)jr * cfg->nx + i));
        }
    }
    return NULL;
}

int orc_render_reference_style(const orc_world *w, const orc_camera *cam, const orc_config *cfg,
                               uint32_t n_threads, uint32_t spp_thread,
                               uint8_t *rgb8, orc_counters *counters, double *seconds)
{
    if (!w || !cam || !cfg || n_threads == 0 || spp_thread == 0 || cfg->nx == 0 || cfg->ny == 0) return -1;
    const size_t npix = (size_t)cfg->nx * cfg->ny;
    ref_job_t *jobs = (ref_job_t *)calloc(n_threads, sizeof(ref_job_t));
    pthread_t *th = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
    uint8_t *images = (uint8_t *)malloc(npix * 3 * n_threads);
    if (!jobs || !th || !images) { free(jobs); free(th); free(images); return -2; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);                                   /* :584 */
    for (uint32_t t = 0; t < n_threads; t++) {
        jobs[t].w = w; jobs[t].cam = cam; jobs[t].cfg = cfg; jobs[t].spp = spp_thread;
        jobs[t].seed = (uint64_t)t * 123u;                                 /* :600 */
        jobs[t].img = images + npix * 3 * t;
        pthread_create(&th[t], NULL, reference_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);   /* :605-607 */
    clock_gettime(CLOCK_MONOTONIC, &t1);                                   /* :608 */
    if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (rgb8) {
        for (size_t idx = 0; idx < npix * 3; idx++) {                      /* :610-617 integer mean */
            uint32_t acc = 0;
            for (uint32_t t = 0; t < n_threads; t++) acc += images[npix * 3 * t + idx];
            rgb8[idx] = (uint8_t)(acc / n_threads);
        }
    }
    if (counters) { memset(counters, 0, sizeof *counters); for (uint32_t t = 0; t < n_threads; t++) counters_add(counters, &jobs[t].cnt); }
    free(jobs); free(th); free(images);
    return 0;
}

/* ====================================================================================
 * scenes
 * ==================================================================================== */
static orc_v3 rius_sfc(orc_sfc64 *s)
{   /* random_in_unit_sphere on the sequential stream */
    rng_t g; memset(&g, 0, sizeof g); g.counter_mode = 0; g.sfc = *s;
    orc_v3 r = random_in_unit_sphere(&g);
    *s = g.sfc;
    return r;
}
/* Random_Scene, raytracer.zig:644-689 */
int orc_random_scene(uint32_t half_size, uint64_t seed, orc_sphere *out, uint32_t cap)
{
    orc_sfc64 rng; orc_sfc64_init(&rng, seed);
    uint32_t n = 0;
#define PUSH(S) do { if (n >= cap) return -1; out[n++] = (S); } while (0)
    for (uint32_t A = 0; A < 2 * half_size; A++) {
        for (uint32_t B = 0; B < 2 * half_size; B++) {
            const float a = (float)A - (float)half_size;
            const float b = (float)B - (float)half_size;
            const uint8_t choose_mat = (uint8_t)(orc_sfc64_float(&rng) * 100);                  /* :652 */
            const float cx = a + 0.9f * orc_sfc64_float(&rng);                                   /* :653 */
            const float cz = b + 0.9f * orc_sfc64_float(&rng);
            const orc_v3 center = v3(cx, 0.2f, cz);
            if (orc_length(vsub(center, v3(4.0f, 0.2f, 0.0f))) > 0.9f && (orc_sfc64_float(&rng) > 0.4f)) {  /* :655 */
                orc_sphere s; memset(&s, 0, sizeof s);
                s.center = center; s.radius = 0.2f;
                if (choose_mat <= 80) {                                                          /* :657-672 */
                    const orc_v3 rnd_vec = rius_sfc(&rng);
                    const orc_v3 normed_vec = orc_normalize(vmul(rnd_vec, rnd_vec));
                    const float xi = orc_sfc64_float(&rng);
                    const float scale = sqrtf(xi * xi) * 0.5f + 0.5f;    /* sqrt(pow(f32, xi, 2)) * 0.5 + 0.5 */
                    s.albedo = vmul(normed_vec, vsplat(scale));
                    if (choose_mat <= 50) { s.kind = ORC_LAMBERTIAN; s.param = 0.0f; }
                    else { s.kind = ORC_METAL; s.param = 0.3f * orc_sfc64_float(&rng); }
                } else if (choose_mat <= 90) {                                                   /* :673-674 */
                    s.kind = ORC_DIELECTRIC; s.albedo = vsplat(0.95f);
                    s.param = 1.5f + 8.5f * orc_sfc64_float(&rng);
                } else {                                                                         /* :675-680 */
                    const float rnd_index = orc_sfc64_float(&rng);
                    orc_sphere inside = s;
                    inside.radius = -0.15f; inside.kind = ORC_DIELECTRIC; inside.albedo = vsplat(0.95f);
                    inside.param = 1.5f + 3.5f * rnd_index;
                    PUSH(inside);
                    s.kind = ORC_DIELECTRIC; s.albedo = vsplat(0.95f); s.param = 1.5f + 3.5f * rnd_index;
                }
                PUSH(s);                                                                         /* :682 */
            }
        }
    }
    {   /* :686-689 */
        orc_sphere g0 = { v3(0.0f, -1000.0f, 0.0f), 1000.0f, ORC_LAMBERTIAN, v3(0.5f, 0.5f, 0.4f), 0.0f };
        orc_sphere g1 = { v3(0.0f, 1.0f, 0.0f), 1.0f, ORC_DIELECTRIC, v3(1.0f, 1.0f, 1.0f), 1.5f };
        orc_sphere g2 = { v3(4.0f, 1.0f, 0.0f), 1.0f, ORC_METAL, v3(0.7f, 0.6f, 0.5f), 0.0f };
        orc_sphere g3 = { v3(-4.0f, 1.0f, 0.0f), 1.0f, ORC_LAMBERTIAN, v3(0.4f, 0.2f, 0.1f), 0.0f };
        PUSH(g0); PUSH(g1); PUSH(g2); PUSH(g3);
    }
#undef PUSH
    return (int)n;
}
/* book_segments.zig:146-149 geometry, :162 attenuation 0.5 => lambertian(0.5) */
int orc_two_sphere_scene(orc_sphere *out, uint32_t cap)
{
    if (cap < 2) return -1;
    orc_sphere s0 = { v3(0.0f, 0.0f, -1.0f), 0.5f, ORC_LAMBERTIAN, v3(0.5f, 0.5f, 0.5f), 0.0f };
    orc_sphere s1 = { v3(0.0f, -100.5f, -1.0f), 100.0f, ORC_LAMBERTIAN, v3(0.5f, 0.5f, 0.5f), 0.0f };
    out[0] = s0; out[1] = s1;
    return 2;
}

/* Parametric sphere after par_shapes.h (prideout/par, wrapped by zmesh; not in /root/reference):
 * vertex(stack, slice): phi = stack/stacks*pi, theta = slice/slices*2pi,
 * (cos(theta) sin(phi), sin(theta) sin(phi), cos(phi)); two triangles per (stack, slice) quad with
 * par_shapes' index order; triangles with a repeated corner position (the pole fans) are dropped,
 * as par_shapes_remove_degenerate does.  main.zig:128-138: (150,150) for radius > 50, (20,20) for
 * radius < 1, else (30,30); scaled by the (signed) radius; placed at the centre (main.zig:143). */
static orc_v3 param_vertex(int stack, int slice, int stacks, int slices)
{
    const float PI = 3.14159265358979323846f;
    const float phi = ((float)stack / (float)stacks) * PI;
    const float theta = ((float)slice / (float)slices) * 2 * PI;
    return v3(cosf(theta) * sinf(phi), sinf(theta) * sinf(phi), cosf(phi));
}
/* par_shapes_remove_degenerate keeps a triangle iff |cross(b-a, c-a)|^2 >= (2*min_area)^2 on the
 * unit sphere.  par_shapes' own epsilon is not verifiable here; min_area = 5e-8 removes exactly the
 * 2*slices zero-area pole-fan triangles for every tessellation the reference asks for (the
 * smallest real triangle at 150x150 has area 9e-6, the south-pole slivers 4e-11). */
static int degenerate(orc_v3 a, orc_v3 b, orc_v3 c)
{
    const orc_v3 cp = orc_cross(vsub(b, a), vsub(c, a));
    return !(orc_dot(cp, cp) >= 1e-14f);
}
int orc_tessellate_spheres(const orc_sphere *sph, uint32_t n, int slices_in, int stacks_in,
                           orc_triangle *out, uint32_t cap)
{
    uint32_t nt = 0;
    for (uint32_t si = 0; si < n; si++) {
        int slices = slices_in, stacks = stacks_in;
        const float radius = sph[si].radius;
        if (slices <= 0 || stacks <= 0) {
            if (radius > 50) slices = stacks = 150;
            else if (radius < 1.0f) slices = stacks = 20;
            else slices = stacks = 30;
        }
        for (int stack = 0; stack < stacks; stack++) {
            for (int slice = 0; slice < slices; slice++) {
                const int next = slice + 1;
                /* faces (v+slice+slices+1, v+next, v+slice) and (v+slice+slices+1, v+next+slices+1, v+next) */
                const orc_v3 A = param_vertex(stack + 1, slice, stacks, slices);
                const orc_v3 B = param_vertex(stack, next, stacks, slices);
                const orc_v3 C = param_vertex(stack, slice, stacks, slices);
                const orc_v3 D = param_vertex(stack + 1, next, stacks, slices);
                const orc_v3 tris[2][3] = { { A, B, C }, { A, D, B } };
                for (int k = 0; k < 2; k++) {
                    const orc_v3 p0 = tris[k][0], p1 = tris[k][1], p2 = tris[k][2];
                    if (degenerate(p0, p1, p2)) continue;
                    if (nt >= cap) return -1;
                    orc_triangle t;
                    t.v0 = vadd(vmul(p0, vsplat(radius)), sph[si].center);
                    t.v1 = vadd(vmul(p1, vsplat(radius)), sph[si].center);
                    t.v2 = vadd(vmul(p2, vsplat(radius)), sph[si].center);
                    t.material = si;
                    out[nt++] = t;
                }
            }
        }
    }
    return (int)nt;
}

/* image.zig:18-37 */
int orc_write_ppm(const char *fname, const uint8_t *rgb8, uint32_t nx, uint32_t ny)
{
    FILE *f = fopen(fname, "w");
    if (!f) return -1;
    fprintf(f, "P3\n%u %u\n255\n", nx, ny);
    for (size_t k = 0; k < (size_t)nx * ny; k++)
        fprintf(f, "%u %u %u\n", rgb8[3 * k], rgb8[3 * k + 1], rgb8[3 * k + 2]);
    fclose(f);
    return 0;
}
