// Host-side mirror of src/vector.zig:4-58 (Vector(vec_type) helpers over @Vector(3, f32)).
// Strict binary32: build with -ffp-contract=off.  Operation order follows the reference:
// sum is the ordered reduction ((x + y) + z); normalize is three true divides.
#pragma once
#include <cmath>
#include <cstdint>

namespace rt {

struct vec_type {   // raytracer.zig:12  `pub const vec_type = @Vector(3, f32)`
    float x = 0, y = 0, z = 0;
    constexpr vec_type() = default;
    constexpr vec_type(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](unsigned i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec_type operator+(vec_type a, vec_type b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec_type operator-(vec_type a, vec_type b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec_type operator*(vec_type a, vec_type b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline vec_type operator/(vec_type a, vec_type b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
inline vec_type operator-(vec_type a) { return { -a.x, -a.y, -a.z }; }

struct vec3f {   // raytracer.zig:13  `const vec3f = vector.Vector(vec_type)`
    static float sum(vec_type v) { return (v.x + v.y) + v.z; }                 // vector.zig:6-8
    static float length_squared(vec_type v) { return sum(v * v); }             // vector.zig:14-17
    static float length(vec_type v) { return std::sqrt(length_squared(v)); }   // vector.zig:10-13
    static vec_type Splat(float s) { return { s, s, s }; }                     // vector.zig:34-36
    static vec_type normalize(vec_type v) { return v / Splat(length(v)); }     // vector.zig:18-20
    static float dot(vec_type a, vec_type b) { return sum(a * b); }            // vector.zig:22-24
    static vec_type cross(vec_type a, vec_type b) {                            // vector.zig:26-33
        return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    }
    static vec_type elem_sqrt(vec_type v) { return { std::sqrt(v.x), std::sqrt(v.y), std::sqrt(v.z) }; }  // :37-39
    static unsigned argmax(vec_type v) {                                       // vector.zig:40-56
        if (v.x > v.y) return v.x > v.z ? 0u : 2u;
        return v.y > v.z ? 1u : 2u;
    }
};

inline vec_type f32x3(float e0, float e1, float e2) { return { e0, e1, e2 }; }   // raytracer.zig:17-19

}  // namespace rt
