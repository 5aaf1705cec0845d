// Device-side arithmetic primitives of the trace path (gfx950).
//
// Everything that feeds a branch on the path (discriminants, t comparisons, dot(dir,n) > 0, the
// Schlick compare) must round exactly like the CPU checker, because path tracing is chaotic: one
// flipped decision changes a whole sample.  So: strict binary32, no contraction (-ffp-contract=off),
// IEEE-correct divide and square root (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt;
// NOT __fsqrt_rn/__fdiv_rn, which are the native approximations in this toolchain), f32 denormals
// kept (gfx950 default).  Operation order follows the reference (file:line under
// /root/reference/src).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }   // three true divides
__device__ __forceinline__ f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }

// vector.zig:6-8,22-24: ordered reduction ((x + y) + z) of the element products
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// vector.zig:26-33
__device__ __forceinline__ f3 cross(f3 a, f3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Philox4x32-10 (Random123): the counter RNG shared bit-for-bit with the CPU checker.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniform in [0,1): top 24 bits, exact conversion and exact power-of-two scale
__device__ __forceinline__ float u32_to_unit_float(uint32_t x) { return (float)(x >> 8) * 0x1.0p-24f; }

// std.math.cbrt(f32) == musl cbrtf (call site raytracer.zig:577).  f64 arithmetic is IEEE on both
// sides, so this is bit-identical to the host version.
__device__ __forceinline__ float cbrt_f32(float x)
{
    const uint32_t B1 = 709958130u, B2 = 642849266u;
    uint32_t u = __float_as_uint(x);
    uint32_t hx = u & 0x7fffffffu;
    if (hx >= 0x7f800000u) return x + x;
    if (hx < 0x00800000u) {
        if (hx == 0) return x;
        u = __float_as_uint(x * 0x1p24f);
        hx = (u & 0x7fffffffu) / 3 + B2;
    } else {
        hx = hx / 3 + B1;
    }
    u = (u & 0x80000000u) | hx;
    double T = (double)__uint_as_float(u);
    double r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    return (float)T;
}
// cbrt on the only inputs the path feeds it: x = k * 2^-24, 0 <= k < 2^24 (u32_to_unit_float).  f32 only:
// musl's bit-hack start, two of its Halley steps in f32 (error ~2^-22), then one Newton step whose residual
// y^3 - x is computed exactly with FMAs, so y - d carries ~43 good bits before the one final rounding.
// cbrt_f32 rounds a value that is itself within 2^-46 of cbrt(x); the two can only round differently when the
// true value lies within ~2^-19 ulp of a rounding midpoint.  `risky` flags everything within 2^-13 ulp of one
// (1 input in 4096) and the caller re-evaluates those with cbrt_f32.  The result -- not this argument -- is what
// the parity contract rests on: tests/test_gpu_parity.py compares ALL 2^24 inputs bit for bit with the oracle.
__device__ __forceinline__ float cbrt_unit_fast(float x, bool &risky)
{
    float y = __uint_as_float(__float_as_uint(x) / 3u + 709958130u);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const float r = (y * y) * y;
        y = (y * __builtin_fmaf(2.0f, x, r)) * __builtin_amdgcn_rcpf(__builtin_fmaf(2.0f, r, x));
    }
    const float h = y * y, hl = __builtin_fmaf(y, y, -h);          // y^2 = h + hl exactly
    const float p = h * y, pl = __builtin_fmaf(h, y, -p);          // h*y = p + pl exactly
    const float res = (p - x) + __builtin_fmaf(hl, y, pl);         // y^3 - x
    const float d = res * __builtin_amdgcn_rcpf(3.0f * h);
    const float out = y - d;
    const float dropped = (y - out) - d;                           // what the rounding of y - d lost
    const float ulp = __uint_as_float((__float_as_uint(out) & 0x7f800000u) - (23u << 23));
    const bool zero = x == 0.0f;
    risky = !zero && !(__builtin_fabsf(dropped) < ulp * (0.5f - 0x1.0p-13f));
    return zero ? 0.0f : out;
}
// x / y, correctly rounded, from r = RN(1/y) computed once elsewhere (Markstein): q = RN(x*r) is within one ulp,
// x - q*y is exact in one FMA, and the FMA correction lands on RN(x/y).  Valid when nothing under/overflows
// (callers keep |x| >= 2^-100 and |x/y| far below 2^127) and y's significand is not all ones (the caller checks
// that where r is made).  x = -0 would give +0: callers send zeros down the IEEE path.
__device__ __forceinline__ float div_recip(float x, float y, float r)
{
    const float q = x * r;
    const float rem = __builtin_fmaf(-q, y, x);
    return __builtin_fmaf(rem, r, q);
}
// std.math.pow(f32, x, 5) as Zig evaluates it: x * ((x*x)*(x*x)) (raytracer.zig:467)
__device__ __forceinline__ float pow5(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x * x4; }

}  // namespace rtd
