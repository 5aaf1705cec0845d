// Host-side mirror of the Material types.  src/material.zig is an empty stub (material.zig:1-21);
// the live types are raytracer.zig:402-449.  The HIP path consumes them flattened into
// (kind, rgb, param) -- see rt_spheres in include/rt_hip.h.  The scatter arithmetic itself runs on
// the GPU (raytrace_amd/csrc/rt_kernels.hip); these are the data carriers of the drop-in API.
#pragma once
#include "../../include/rt_hip.h"
#include "vector.hpp"

namespace rt {

struct Metal      { vec_type albedo; float roughness = 0; };   // raytracer.zig:417-419
struct Lambertian { vec_type albedo; };                        // raytracer.zig:436-437
struct Dielectric { vec_type color; float index = 1; };        // raytracer.zig:447-449

struct Material {                                              // raytracer.zig:402-406 union(enum)
    enum class Tag : uint32_t { lambertian = RT_LAMBERTIAN, metal = RT_METAL, dielectric = RT_DIELECTRIC };
    Tag tag = Tag::lambertian;
    vec_type rgb;      // albedo | color
    float param = 0;   // roughness | index

    static Material metal(vec_type albedo, float roughness) { return { Tag::metal, albedo, roughness }; }
    static Material lambertian(vec_type albedo) { return { Tag::lambertian, albedo, 0.0f }; }
    static Material dielectric(vec_type color, float index) { return { Tag::dielectric, color, index }; }
    Material() = default;
    Material(Tag t, vec_type c, float p) : tag(t), rgb(c), param(p) {}
    Material(const Metal &m) : tag(Tag::metal), rgb(m.albedo), param(m.roughness) {}
    Material(const Lambertian &m) : tag(Tag::lambertian), rgb(m.albedo), param(0) {}
    Material(const Dielectric &m) : tag(Tag::dielectric), rgb(m.color), param(m.index) {}
};

}  // namespace rt
