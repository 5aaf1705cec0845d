/*
 * oracle/rt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the per-pixel trace path of Copper280z/raytrace
 * (src/raytracer.zig, src/vector.zig, src/ray.zig, src/camera.zig, src/image.zig).
 * It is the CHECKER for the HIP path: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  Nothing under raytrace_amd/ links,
 * imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" for everything but the vector helpers.  The
 * reference is Zig and there is no Zig toolchain in this image, so the reference
 * itself cannot be run; its own tests pin only vector.zig:60-102 (4 known-answer
 * tests, reproduced in tests/).  The sfc64 core is pinned against numpy's SFC64
 * bit generator; Zig std's Random.float(f32), std.math.cbrt/pow and std.sort.heap
 * are restated from their published algorithms (Zig std 0.14, musl cbrtf) and are
 * not verifiable here.  The counter RNG (Philox4x32-10, Random123) is build-defined
 * and pinned against the published Random123 known-answer vectors.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material tags: the C-side encoding of raytracer.zig:402-406 */
enum { ORC_LAMBERTIAN = 0, ORC_METAL = 1, ORC_DIELECTRIC = 2 };

typedef struct { float x, y, z; } orc_v3;

/* raytracer.zig:38-41 Sphere + :402-449 Material flattened */
typedef struct {
    orc_v3   center;
    float    radius;    /* may be negative (hollow glass, raytracer.zig:677) */
    uint32_t kind;      /* ORC_* */
    orc_v3   albedo;    /* albedo (lambertian, metal) or color (dielectric) */
    float    param;     /* roughness (metal) or index (dielectric); 0 for lambertian */
} orc_sphere;

/* build-defined triangle primitive (the reference has no ray-triangle code) */
typedef struct {
    orc_v3   v0, v1, v2;
    uint32_t material;  /* index into a material table of orc_sphere-style records */
} orc_triangle;

/* camera.zig:9-14 / :20-24 -- four vectors; subtract_origin = 1 for MovableCamera (camera.zig:39) */
typedef struct {
    orc_v3   lower_left, horizontal, vertical, origin;
    uint32_t subtract_origin;
} orc_camera;

/* traversal modes */
enum { ORC_TRAV_LIST = 0,    /* hittableList semantics, raytracer.zig:88-115 (normative); triangles: the checker's own
                                conservative BVH (a speed-up that returns the list answer) */
       ORC_TRAV_BVH_REF = 1, /* build_BVH + 2-axis AABB, raytracer.zig:117-395 (quirks Q1, Q2) */
       ORC_TRAV_BRUTE = 2    /* spheres as LIST; triangles as a plain loop over every triangle in index order: no culling
                                structure at all (small meshes; pins the speed-up above) */ };

typedef struct {
    uint32_t nx, ny;
    uint32_t spp_begin, spp_end, spp_total; /* global sample indices [begin,end); mean divides by spp_total */
    uint32_t max_depth;                     /* reference: 16 (raytracer.zig:556) */
    float    t_min;                         /* reference: 0.001 (raytracer.zig:555) */
    uint64_t seed;
    uint32_t row_offset_q3;                 /* 1: j = ny - jr (raytracer.zig:523); 0: j = ny-1-jr */
    uint32_t traversal;                     /* ORC_TRAV_* (spheres only) */
    uint32_t spp_chunk;                     /* 0: one running sum; >0: partial sums of this many samples, added in order */
    uint32_t threads;                       /* worker threads for the pixel loop (0/1 = serial) */
} orc_config;

/* op counters used for the algorithmic-flops model */
typedef struct {
    uint64_t rays;            /* world.hit evaluations */
    uint64_t primary_rays;
    uint64_t sphere_tests;    /* Sphere.hit calls */
    uint64_t aabb_tests;      /* AxisAlignedBBox.hit calls */
    uint64_t tri_tests;
    uint64_t root_evals;      /* discriminant > 0 */
    uint64_t hits, misses;
    uint64_t scatter_lambertian, scatter_metal, scatter_dielectric;
    uint64_t absorbed;        /* metal scattered below surface */
    uint64_t depth_exceeded;
    uint64_t rays_by_depth[40];
} orc_counters;

/* ---- RNG ------------------------------------------------------------------------- */
typedef struct { uint64_t a, b, c, counter; } orc_sfc64;
void     orc_sfc64_init(orc_sfc64 *s, uint64_t seed);          /* std.Random.Sfc64.init */
uint64_t orc_sfc64_next(orc_sfc64 *s);
float    orc_sfc64_float(orc_sfc64 *s);                        /* std.Random.float(f32) */
void     orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float    orc_u32_to_unit_float(uint32_t x);                    /* (x >> 8) * 2^-24 */

/* ---- math helpers exposed for known-answer tests ---------------------------------- */
float  orc_dot(orc_v3 a, orc_v3 b);
float  orc_length(orc_v3 a);
orc_v3 orc_normalize(orc_v3 a);
orc_v3 orc_cross(orc_v3 a, orc_v3 b);
uint32_t orc_argmax(orc_v3 a);
float  orc_cbrtf(float x);                                      /* musl cbrtf == Zig std.math.cbrt(f32) */
void   orc_cbrtf_array(const float *x, float *out, size_t n);       /* orc_cbrtf over an array (test aid) */
float  orc_pow5(float x);
float  orc_schlick(float cosine, float idx);
orc_v3 orc_reflect(orc_v3 v, orc_v3 n);
int    orc_refract(orc_v3 v, orc_v3 n, float ni_nt, orc_v3 *out);
/* Sphere.hit, raytracer.zig:42-78: returns 1 on hit and fills t, p, normal */
int    orc_sphere_hit(const orc_sphere *s, orc_v3 ro, orc_v3 rd, float t_min, float t_max,
                      float *t, orc_v3 *p, orc_v3 *normal);
/* rotates the vertices so that v0 faces the longest edge (worlds do this to every triangle at creation) */
void   orc_canonical_triangle(orc_triangle *tr);
/* the test on the triangle AS GIVEN (no rotation) */
int    orc_triangle_hit(const orc_triangle *tr, orc_v3 ro, orc_v3 rd, float t_min, float t_max,
                        float *t, orc_v3 *p, orc_v3 *normal);
orc_v3 orc_sky(orc_v3 dir);                                     /* raytracer.zig:569-571 */

/* ---- cameras ---------------------------------------------------------------------- */
void orc_camera_basic(orc_camera *cam);                         /* camera.zig:9-14 */
void orc_camera_set_position(orc_camera *cam, orc_v3 lookfrom, orc_v3 lookat, orc_v3 vup,
                             float vfov, float aspect);         /* camera.zig:25-37 */
void orc_camera_get_ray(const orc_camera *cam, float s, float t, orc_v3 *ro, orc_v3 *rd);

/* ---- scenes ----------------------------------------------------------------------- */
/* Random_Scene, raytracer.zig:644-689 (sphere list only). Returns sphere count (<= cap) or -1. */
int orc_random_scene(uint32_t half_size, uint64_t seed, orc_sphere *out, uint32_t cap);
/* book_segments.zig:146-149 + 0.5 attenuation (:162) */
int orc_two_sphere_scene(orc_sphere *out, uint32_t cap);
/* parametric-sphere tessellation after par_shapes (zmesh) and main.zig:128-138.
 * Returns triangle count written (<= cap) or -1; slices/stacks <= 0 selects the
 * reference's per-radius choice. */
int orc_tessellate_spheres(const orc_sphere *sph, uint32_t n, int slices, int stacks,
                           orc_triangle *out, uint32_t cap);

/* ---- world handle ----------------------------------------------------------------- */
typedef struct orc_world orc_world;
orc_world *orc_world_create(const orc_sphere *sph, uint32_t n_sph,
                            const orc_triangle *tri, uint32_t n_tri,
                            const orc_sphere *materials, uint32_t n_mat);
void       orc_world_destroy(orc_world *w);
uint32_t   orc_world_bvh_nodes(const orc_world *w);

/* ---- rendering -------------------------------------------------------------------- */
/* Counter-RNG render (the mode GPU parity is defined in).
 *   pixels/n_pixels: optional list of output pixel indices (jr*nx+i) to render; NULL = whole image.
 *   accum_mean: [n][3] linear mean radiance (sum over [spp_begin,spp_end) / spp_total)
 *   rgb8:       [n][3] trunc(255.99*sqrt(mean)), raytracer.zig:527,545-549
 *   raycount:   [n] world.hit evaluations per pixel
 * where n = n_pixels or nx*ny.  Any output may be NULL. */
int orc_render_counter(const orc_world *w, const orc_camera *cam, const orc_config *cfg,
                       const uint32_t *pixels, uint32_t n_pixels,
                       float *accum_mean, uint8_t *rgb8, uint32_t *raycount,
                       orc_counters *counters);

/* Reference-style render: restates render_and_write (raytracer.zig:580-618): n_threads threads,
 * each renders the whole image with spp_thread samples from its own sequential sfc64 stream
 * seeded i*123, then the per-pixel integer mean of the u8 images.  Returns wall seconds
 * spent between spawn and join (raytracer.zig:584-608) through *seconds. */
int orc_render_reference_style(const orc_world *w, const orc_camera *cam, const orc_config *cfg,
                               uint32_t n_threads, uint32_t spp_thread,
                               uint8_t *rgb8, orc_counters *counters, double *seconds);

/* image.zig:18-37 ASCII P3 writer */
int orc_write_ppm(const char *fname, const uint8_t *rgb8, uint32_t nx, uint32_t ny);

#ifdef __cplusplus
}
#endif
#endif
