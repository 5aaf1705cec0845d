// Host check of the division the kernels use where a reciprocal is made once (rt_device.h div_recip, rt_api.cpp
// recip_for_div): q = RN(x*r), rem = x - q*y (one FMA, exact), result = fma(rem, r, q) must equal the IEEE quotient
// bit for bit whenever r = RN(1/y) and the operands are in the range the callers guarantee.  Random operands plus
// the structured cases of the trace path: (i + k*2^-24) / nx and (p - c) / radius with |p - c| ~ |radius|.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

static float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static uint64_t s = 88172645463325252ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

static float div_recip(float x, float y, float r)
{
    const float q = x * r;
    const float rem = std::fmaf(-q, y, x);
    return std::fmaf(rem, r, q);
}
static bool eligible_denominator(float y)
{
    const float ay = std::fabs(y);
    return ay >= 0x1.0p-60f && ay <= 0x1.0p60f && (f2u(y) & 0x7fffffu) != 0x7fffffu;
}

int main()
{
    long bad = 0, tested = 0;
    for (long k = 0; k < 20000000; ++k) {                      // random significands, |x/y| within 2^+-30
        const uint64_t r0 = rnd();
        uint32_t my = (uint32_t)(r0 & 0x7fffff);
        const uint32_t mx = (uint32_t)((r0 >> 23) & 0x7fffff);
        const int ey = (int)((r0 >> 46) % 81) - 40, ex = ey + (int)((r0 >> 53) % 61) - 30;
        if ((k & 15) == 0) my = (k & 16) ? 0x7ffffe : (uint32_t)((r0 >> 40) & 0xff);
        float y = u2f(((uint32_t)(ey + 127) << 23) | my), x = u2f(((uint32_t)(ex + 127) << 23) | mx);
        if (r0 >> 63) x = -x;
        if ((r0 >> 62) & 1) y = -y;
        if (!eligible_denominator(y) || std::fabs(x) < 0x1.0p-100f) continue;
        ++tested;
        if (f2u(div_recip(x, y, 1.0f / y)) != f2u(x / y)) ++bad;
    }
    const float widths[] = { 1, 2, 3, 7, 32, 96, 225, 400, 1080, 1920, 2160, 3840, 4097, 65535, 1000003, 16777214 };
    for (float nx : widths) {                                  // camera: u = (i + xi) / nx
        if (!eligible_denominator(nx)) { ++bad; continue; }
        const float r = 1.0f / nx;
        for (long k = 0; k < 500000; ++k) {
            const uint64_t r0 = rnd();
            const float i = (float)(uint32_t)(r0 % (uint64_t)(nx + 1.0f)), xi = (float)((r0 >> 32) & 0xffffff) * 0x1.0p-24f;
            const float x = i + xi;
            ++tested;
            if (f2u(div_recip(x, nx, r)) != f2u(x / nx)) ++bad;
        }
    }
    const float radii[] = { 0.2f, -0.15f, 0.5f, 1.0f, 100.0f, 1000.0f, 0.3333333f, -7.77f, 1e-3f, 12345.678f };
    for (float rad : radii) {                                  // normal: (p - c) / radius, |p - c| <= ~|radius|
        const float r = 1.0f / rad;
        for (long k = 0; k < 500000; ++k) {
            const uint64_t r0 = rnd();
            float x = std::fabs(rad) * ((float)(r0 & 0xffffff) * 0x1.0p-24f) * (1.0f + (float)((r0 >> 24) & 0xff) * 0x1.0p-20f);
            if (r0 >> 63) x = -x;
            if (std::fabs(x) < 0x1.0p-100f) continue;
            ++tested;
            if (f2u(div_recip(x, rad, r)) != f2u(x / rad)) ++bad;
        }
    }
    if (bad) { std::printf("FAIL: %ld of %ld quotients differ from IEEE\n", bad, tested); return 1; }
    std::printf("ok: %ld quotients equal the IEEE divide bit for bit\n", tested);
    return 0;
}
