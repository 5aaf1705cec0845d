// Host-side mirror of src/raytracer.zig: the scene types, Random_Scene, and a render_and_write with
// the reference's call shape whose body hands the per-pixel trace path to the HIP backend through
// the C ABI (include/rt_hip.h).  No tracing arithmetic runs on the host: Render/trace/color
// (raytracer.zig:517-572) live in raytrace_amd/csrc/rt_kernels.hip.
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "camera.hpp"
#include "image.hpp"
#include "material.hpp"
#include "ray.hpp"
#include "vector.hpp"

namespace rt {

struct shape_s { size_t ny = 0; size_t nx = 0; };   // raytracer.zig:26-29 (ny first)

struct Sphere {                                     // raytracer.zig:38-41
    vec_type center;
    Material material;
    float radius = 0;                               // may be negative (hollow glass, :677)
};

// std.Random.Sfc64 + Random.float(f32) (Zig std 0.14; used at raytracer.zig:23,652-679).
struct RndGen {
    uint64_t a = 0, b = 0, c = 0, counter = 1;
    static RndGen init(uint64_t seed) {
        RndGen g; g.a = g.b = g.c = seed; g.counter = 1;
        for (int i = 0; i < 12; ++i) g.next();
        return g;
    }
    uint64_t next() {
        const uint64_t tmp = a + b + counter;
        counter += 1;
        a = b ^ (b >> 11);
        b = c + (c << 3);
        c = ((c << 24) | (c >> 40)) + tmp;
        return tmp;
    }
    static unsigned clz64(uint64_t x) { return x ? (unsigned)__builtin_clzll(x) : 64u; }
    float float_f32() {   // Random.float(f32): 23 mantissa bits + exponent from the leading-zero count
        const uint64_t r = next();
        unsigned lz = clz64(r);
        if (lz >= 41) {
            lz = 41 + clz64(next());
            if (lz == 41 + 64) lz += (unsigned)__builtin_clz((uint32_t)next() | 7u);
        }
        const uint32_t bits = ((126u - lz) << 23) | ((uint32_t)r & 0x7FFFFFu);
        float f; std::memcpy(&f, &bits, 4);
        return f;
    }
};

// std.math.cbrt(f32) (musl cbrtf): f64 Newton steps on a bit-hack seed.  Kept identical to the
// device version in csrc/rt_device.h.
inline float cbrt_f32(float x) {
    const uint32_t B1 = 709958130u, B2 = 642849266u;
    uint32_t u; std::memcpy(&u, &x, 4);
    uint32_t hx = u & 0x7fffffffu;
    if (hx >= 0x7f800000u) return x + x;
    if (hx < 0x00800000u) {
        if (hx == 0) return x;
        const float xs = x * 0x1p24f;
        std::memcpy(&u, &xs, 4);
        hx = (u & 0x7fffffffu) / 3 + B2;
    } else {
        hx = hx / 3 + B1;
    }
    u = (u & 0x80000000u) | hx;
    float tf; std::memcpy(&tf, &u, 4);
    double T = tf, r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    return (float)T;
}

// random_in_unit_sphere, raytracer.zig:574-578 (host copy: only the scene generator needs it here)
inline vec_type random_in_unit_sphere(RndGen &rng) {
    const float x = rng.float_f32(), y = rng.float_f32(), z = rng.float_f32();
    const vec_type unit_sphere = vec3f::Splat(2.0f) * vec3f::normalize(vec_type{ x, y, z }) - vec3f::Splat(1.0f);
    return unit_sphere * vec3f::Splat(cbrt_f32(rng.float_f32()));
}

// `Hittable` (raytracer.zig:712): the reference pairs the flat sphere slice with a CPU BVH.  The HIP
// path builds its own acceleration data at upload, so the host-side world is the object list.
struct Hittable {
    std::vector<Sphere> objects;
};

// Random_Scene, raytracer.zig:644-708.  `rng` is consumed exactly as the reference consumes it.
inline Hittable Random_Scene(size_t half_size, RndGen &rng) {
    Hittable world;
    auto &list = world.objects;
    for (size_t A = 0; A < 2 * half_size; ++A) {
        for (size_t B = 0; B < 2 * half_size; ++B) {
            const float a = (float)A - (float)half_size;
            const float b = (float)B - (float)half_size;
            const uint8_t choose_mat = (uint8_t)(rng.float_f32() * 100);
            const float cx = a + 0.9f * rng.float_f32();
            const float cz = b + 0.9f * rng.float_f32();
            const vec_type center{ cx, 0.2f, cz };
            if (vec3f::length(center - vec_type{ 4.0f, 0.2f, 0.0f }) > 0.9f && (rng.float_f32() > 0.4f)) {
                if (choose_mat <= 80) {                    // lambertian 0..50, metal 51..80
                    const vec_type rnd_vec = random_in_unit_sphere(rng);
                    const vec_type normed_vec = vec3f::normalize(rnd_vec * rnd_vec);
                    const float xi = rng.float_f32();
                    const vec_type albedo = normed_vec * vec3f::Splat(std::sqrt(xi * xi) * 0.5f + 0.5f);
                    if (choose_mat <= 50) list.push_back({ center, Material::lambertian(albedo), 0.2f });
                    else list.push_back({ center, Material::metal(albedo, 0.3f * rng.float_f32()), 0.2f });
                } else if (choose_mat <= 90) {             // glass
                    list.push_back({ center, Material::dielectric(vec3f::Splat(0.95f), 1.5f + 8.5f * rng.float_f32()), 0.2f });
                } else {                                   // hollow glass: inner (negative radius) first
                    const float rnd_index = rng.float_f32();
                    list.push_back({ center, Material::dielectric(vec3f::Splat(0.95f), 1.5f + 3.5f * rnd_index), -0.15f });
                    list.push_back({ center, Material::dielectric(vec3f::Splat(0.95f), 1.5f + 3.5f * rnd_index), 0.2f });
                }
            }
        }
    }
    list.push_back({ { 0.0f, -1000.0f, 0.0f }, Material::lambertian({ 0.5f, 0.5f, 0.4f }), 1000.0f });
    list.push_back({ { 0.0f, 1.0f, 0.0f }, Material::dielectric({ 1.0f, 1.0f, 1.0f }, 1.5f), 1.0f });
    list.push_back({ { 4.0f, 1.0f, 0.0f }, Material::metal({ 0.7f, 0.6f, 0.5f }, 0.0f), 1.0f });
    list.push_back({ { -4.0f, 1.0f, 0.0f }, Material::lambertian({ 0.4f, 0.2f, 0.1f }), 1.0f });
    return world;
}

// book_segments.zig:146-149 + :162 -- the "single sphere + ground" diffuse scene.
inline Hittable Two_Sphere_Scene() {
    Hittable world;
    world.objects.push_back({ { 0.0f, 0.0f, -1.0f }, Material::lambertian({ 0.5f, 0.5f, 0.5f }), 0.5f });
    world.objects.push_back({ { 0.0f, -100.5f, -1.0f }, Material::lambertian({ 0.5f, 0.5f, 0.5f }), 100.0f });
    return world;
}

// SoA repack of world.objects for the C ABI (what the Zig shim does).
struct SphereSoA {
    std::vector<float> cx, cy, cz, radius, r, g, b, param;
    std::vector<uint32_t> kind;
    explicit SphereSoA(const std::vector<Sphere> &objs) {
        for (const Sphere &s : objs) {
            cx.push_back(s.center.x); cy.push_back(s.center.y); cz.push_back(s.center.z);
            radius.push_back(s.radius); kind.push_back((uint32_t)s.material.tag);
            r.push_back(s.material.rgb.x); g.push_back(s.material.rgb.y); b.push_back(s.material.rgb.z);
            param.push_back(s.material.param);
        }
    }
    rt_spheres view() const {
        return { (uint32_t)cx.size(), cx.data(), cy.data(), cz.data(), radius.data(), kind.data(),
                 r.data(), g.data(), b.data(), param.data() };
    }
};

// Parametric-sphere tessellation of the scene's spheres (main.zig:115-154; par_shapes via zmesh).
struct Triangle { vec_type v0, v1, v2; uint32_t material; };
inline vec_type parametric_sphere_vertex(int stack, int slice, int stacks, int slices) {
    const float PI = 3.14159265358979323846f;
    const float phi = ((float)stack / (float)stacks) * PI;
    const float theta = ((float)slice / (float)slices) * 2 * PI;
    return { std::cos(theta) * std::sin(phi), std::sin(theta) * std::sin(phi), std::cos(phi) };
}
inline std::vector<Triangle> Tessellate_Spheres(const std::vector<Sphere> &objs, int slices_in = 0, int stacks_in = 0) {
    std::vector<Triangle> out;
    for (size_t si = 0; si < objs.size(); ++si) {
        int slices = slices_in, stacks = stacks_in;
        const float radius = objs[si].radius;
        if (slices <= 0 || stacks <= 0) {                     // main.zig:128-134
            if (radius > 50) slices = stacks = 150;
            else if (radius < 1.0f) slices = stacks = 20;
            else slices = stacks = 30;
        }
        for (int stack = 0; stack < stacks; ++stack) {
            for (int slice = 0; slice < slices; ++slice) {
                const vec_type A = parametric_sphere_vertex(stack + 1, slice, stacks, slices);
                const vec_type B = parametric_sphere_vertex(stack, slice + 1, stacks, slices);
                const vec_type C = parametric_sphere_vertex(stack, slice, stacks, slices);
                const vec_type D = parametric_sphere_vertex(stack + 1, slice + 1, stacks, slices);
                const vec_type quad[2][3] = { { A, B, C }, { A, D, B } };
                for (const auto &t : quad) {
                    const vec_type cp = vec3f::cross(t[1] - t[0], t[2] - t[0]);
                    if (!(vec3f::dot(cp, cp) >= 1e-14f)) continue;   // pole-fan degenerates dropped
                    const vec_type R = vec3f::Splat(radius);         // mesh.scale(radius...), main.zig:137
                    out.push_back({ t[0] * R + objs[si].center, t[1] * R + objs[si].center,
                                    t[2] * R + objs[si].center, (uint32_t)si });
                }
            }
        }
    }
    return out;
}

// The reference's implicit camera input (raytracer.zig:710).
inline MovableCamera cam;

// render_and_write, raytracer.zig:580-633, same argument meaning; `ctx` replaces the allocator
// argument (device memory lives in the ctx).  The reference renders num_samples on EACH of
// getCpuCount() threads; here num_samples is the total per pixel and `seed` keys the counter RNG.
inline int render_and_write(rt_ctx *ctx, const std::string &fname, uint16_t num_samples, const Hittable &world,
                            shape_s shape, img_io::ImgType filetype, uint64_t seed = 1, rt_stats *stats = nullptr) {
    SphereSoA soa(world.objects);
    rt_spheres view = soa.view();
    int rc = rt_upload_scene(ctx, &view, nullptr, nullptr);
    if (rc != RT_OK) return rc;
    rt_config cfg{};
    cfg.nx = (uint32_t)shape.nx; cfg.ny = (uint32_t)shape.ny;
    cfg.spp_begin = 0; cfg.spp_end = num_samples; cfg.spp_total = num_samples;
    cfg.max_depth = 16; cfg.t_min = 0.001f; cfg.seed = seed; cfg.row_offset_q3 = 1; cfg.spp_chunk = RT_SPP_CHUNK_AUTO;
    const rt_camera c = cam.to_abi();
    if (filetype != img_io::ImgType::ppm) return rt_render(ctx, &c, &cfg, nullptr, nullptr, nullptr, stats);
    return rt_render_and_write(ctx, fname.c_str(), &c, &cfg, stats);
}

// ---- the GUI's "Trace" button without the window (main.zig:307-318, 342-345) ---------------------------------------
// The reference's start_trace sets the camera from the GUI's fly camera -- SetPosition(cam_pos, cam_forward,
// cam_pos + (0,1,0), 45, 1200/675): `lookat` receives the FORWARD VECTOR and v_up a point above the camera (quirk Q9,
// kept) -- and then blocks the UI thread inside render_and_write until Scene.ppm is written.  TraceJob is the same call
// in polling form (the C++ twin of zig/hip_backend.zig's TraceJob): begin() = the body of start_trace up to the render,
// poll() = one pass, called once per frame where compute.onCompute was meant to go (main.zig:620-626) -- rgb8 then holds
// the picture so far --, finish() = the file.
struct TraceJob {
    rt_ctx              *ctx = nullptr;
    std::vector<uint8_t> rgb8;
    shape_s              shape{ 675, 1200 };                    // main.zig:312-313
    uint32_t             spp_done = 0;
    bool                 active = false;
    rt_stats             stats{};

    int begin(rt_ctx *c, const Hittable &world, vec_type cam_pos, vec_type cam_forward, uint16_t num_samples, uint32_t n_passes,
              shape_s shp = shape_s{ 675, 1200 }) {
        ctx = c; shape = shp; spp_done = 0; active = false;
        const vec_type cam_up{ cam_pos.x, cam_pos.y + 1.0f, cam_pos.z };                             // main.zig:310
        cam.SetPosition(cam_pos, cam_forward, cam_up, 45.0f, (float)shape.nx / (float)shape.ny);      // main.zig:314-315
        SphereSoA soa(world.objects);
        const rt_spheres view = soa.view();
        int rc = rt_upload_scene(ctx, &view, nullptr, nullptr);
        if (rc != RT_OK) return rc;
        rt_config cfg{};
        cfg.nx = (uint32_t)shape.nx; cfg.ny = (uint32_t)shape.ny;
        cfg.spp_begin = 0; cfg.spp_end = num_samples; cfg.spp_total = num_samples;
        cfg.max_depth = 16; cfg.t_min = 0.001f; cfg.seed = 1; cfg.row_offset_q3 = 1; cfg.spp_chunk = RT_SPP_CHUNK_AUTO;   // (as the Zig twin: short passes need the small items)
        const rt_camera abi = cam.to_abi();
        rgb8.assign(shape.nx * shape.ny * 3, 0);
        rc = rt_progressive_begin(ctx, &abi, &cfg, n_passes);
        active = rc == RT_OK;
        return rc;
    }
    // one pass; RT_PROGRESS_MORE while passes remain, RT_OK after the last one, an error code otherwise
    int poll() {
        if (!active) return RT_E_STATE;
        const int rc = rt_progressive_step(ctx, rgb8.data(), nullptr, &spp_done, &stats);
        active = rc == RT_PROGRESS_MORE;
        return rc;
    }
    int finish(const std::string &fname) const {                                                     // main.zig:317 "Scene.ppm"
        return rt_write_ppm(fname.c_str(), rgb8.data(), (uint32_t)shape.nx, (uint32_t)shape.ny);
    }
};

// start_trace (main.zig:307-318) as the reference has it: blocking, one file at the end.
inline int start_trace(rt_ctx *ctx, const Hittable &world, vec_type cam_pos, vec_type cam_forward, uint16_t num_samples,
                       const std::string &fname = "Scene.ppm", shape_s shape = shape_s{ 675, 1200 }) {
    TraceJob job;
    int rc = job.begin(ctx, world, cam_pos, cam_forward, num_samples, 1, shape);
    while (rc == RT_OK && job.active) { rc = job.poll(); if (rc == RT_PROGRESS_MORE) rc = RT_OK; }
    return rc == RT_OK ? job.finish(fname) : rc;
}

}  // namespace rt
