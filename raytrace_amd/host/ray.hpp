// Host-side mirror of src/ray.zig:4-15.
#pragma once
#include "vector.hpp"

namespace rt {

struct Ray {                       // ray.zig:7-13
    vec_type origin;
    vec_type direction;            // never normalised on the path
    vec_type point_at_parameter(float t) const { return origin + vec3f::Splat(t) * direction; }   // ray.zig:10-12
};

}  // namespace rt
