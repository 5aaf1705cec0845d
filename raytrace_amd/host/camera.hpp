// Host-side mirror of src/camera.zig:5-44.  The cameras keep the reference's field names and
// methods; to_abi() packs them into the C-ABI rt_camera that crosses into the HIP path, replacing
// the implicit global `pub var cam` (raytracer.zig:710).
#pragma once
#include "../../include/rt_hip.h"
#include "ray.hpp"

namespace rt {

struct BasicCamera {               // camera.zig:9-19
    vec_type lower_left{ -2.0f, -1.0f, -1.0f };
    vec_type horizontal{ 4.0f, 0.0f, 0.0f };
    vec_type vertical{ 0.0f, 2.0f, 0.0f };
    vec_type origin{ 0.0f, 0.0f, 0.0f };

    Ray get_ray(float u, float v) const {   // camera.zig:15-18
        return Ray{ origin, lower_left + vec3f::Splat(u) * horizontal + vec3f::Splat(v) * vertical };
    }
    rt_camera to_abi() const { return pack(lower_left, horizontal, vertical, origin, 0u); }

    static rt_camera pack(vec_type ll, vec_type h, vec_type v, vec_type o, uint32_t sub) {
        rt_camera c;
        c.lower_left[0] = ll.x; c.lower_left[1] = ll.y; c.lower_left[2] = ll.z;
        c.horizontal[0] = h.x;  c.horizontal[1] = h.y;  c.horizontal[2] = h.z;
        c.vertical[0] = v.x;    c.vertical[1] = v.y;    c.vertical[2] = v.z;
        c.origin[0] = o.x;      c.origin[1] = o.y;      c.origin[2] = o.z;
        c.subtract_origin = sub;
        return c;
    }
};

struct MovableCamera {             // camera.zig:20-42
    vec_type lower_left{ -2.0f, -1.0f, -1.0f };
    vec_type horizontal{ 4.0f, 0.0f, 0.0f };
    vec_type vertical{ 0.0f, 2.0f, 0.0f };
    vec_type origin{ 0.0f, 0.0f, 0.0f };

    void SetPosition(vec_type lookfrom, vec_type lookat, vec_type v_up, float vfov, float aspect) {   // :25-37
        origin = lookfrom;
        const float theta = vfov * 3.14159265358979323846f / 180.0f;
        const float half_height = std::tan(theta / 2);
        const float half_width = aspect * half_height;
        const vec_type w = vec3f::normalize(lookfrom - lookat);
        const vec_type u = vec3f::normalize(vec3f::cross(v_up, w));
        const vec_type v = vec3f::cross(w, u);
        lower_left = origin - vec3f::Splat(half_width) * u - vec3f::Splat(half_height) * v - w;
        horizontal = vec3f::Splat(2 * half_width) * u;
        vertical = vec3f::Splat(2 * half_height) * v;
    }
    Ray get_ray(float s, float t) const {   // camera.zig:38-41
        return Ray{ origin, lower_left + vec3f::Splat(s) * horizontal + vec3f::Splat(t) * vertical - origin };
    }
    rt_camera to_abi() const { return BasicCamera::pack(lower_left, horizontal, vertical, origin, 1u); }
};

}  // namespace rt
