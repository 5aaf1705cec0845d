// Microbenchmark: issue rate of plain vs packed f32 VALU ops on gfx950 at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    float2_ ss = {s, s};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) { // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 2) { // 4 independent v_pk_mul_f32 (8 lane-mults)
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss));
        } else if (MODE == 3) { // 4 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss));
        } else if (MODE == 4) { // 8 v_mul with an SGPR operand
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
        } else if (MODE == 5) { // 8 v_sqrt_f32
            asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 6) { // 8 v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 7) { // 8 v_cmp + v_cndmask pairs (4 pairs)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %0, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %2, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %4, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %6, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
        } else if (MODE == 8) { // 4 v_mad_u64_u32 (64-bit product + 64-bit add)
            unsigned long long r0, r1, r2, r3;
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0\n"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(s) : "vcc");
            a0 = (float)(unsigned)(r0 >> 32 ^ r0); a1 = (float)(unsigned)(r1 >> 32 ^ r1); a2 = (float)(unsigned)(r2 >> 32 ^ r2); a3 = (float)(unsigned)(r3 >> 32 ^ r3);
        } else if (MODE == 9) { // 8 v_mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                         "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 10) { // 8 v_fma_f64 (4 independent, 2 each)
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3, ds = s;
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                         "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(ds));
            a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
        } else if (MODE == 11) { // 8 v_mad_u64_u32, accumulate form (4 independent chains, 2 each)
                        unsigned long long r0 = i, r1 = i + 1, r2 = i + 2, r3 = i + 3;
            unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3), us = __float_as_uint(s);
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, %0\n v_mad_u64_u32 %1, vcc, %5, %8, %1\n v_mad_u64_u32 %2, vcc, %6, %8, %2\n v_mad_u64_u32 %3, vcc, %7, %8, %3\n"
                         "v_mad_u64_u32 %0, vcc, %4, %8, %0\n v_mad_u64_u32 %1, vcc, %5, %8, %1\n v_mad_u64_u32 %2, vcc, %6, %8, %2\n v_mad_u64_u32 %3, vcc, %7, %8, %3\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(us) : "vcc");
            if (r0 + r1 + r2 + r3 == 12345ull) a4 += 1.0f;
        } else if (MODE == 12) { // 8 v_rcp_f64 (4 chains, 2 each)
            double d0 = a0 + 2.0, d1 = a1 + 2.0, d2 = a2 + 2.0, d3 = a3 + 2.0;
            asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                         "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            if (d0 + d1 + d2 + d3 == 12345.0) a4 += 1.0f;
        } else if (MODE == 13) { // 8 v_mul_f64
            double d0 = a0 + 2.0, d1 = a1 + 2.0, d2 = a2 + 2.0, d3 = a3 + 2.0, ds = s;
            asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(ds));
            if (d0 + d1 + d2 + d3 == 12345.0) a4 += 1.0f;
        } else if (MODE == 14) { // 8 v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 15) { // 8 v_add_f64
            double d0 = a0 + 2.0, d1 = a1 + 2.0, d2 = a2 + 2.0, d3 = a3 + 2.0, ds = s;
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(ds));
            if (d0 + d1 + d2 + d3 == 12345.0) a4 += 1.0f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char *name, int wavesPerSimd, int instrPerIter, int laneOpsPerIter, float *d)
{
    const int iters = 50000;
    const int blocks = 256 * wavesPerSimd;   // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1000, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * instrPerIter * wavesPerSimd;
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    double tlops = (double)iters * laneOpsPerIter * 64.0 * blocks * 4 / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d: %.3f ms, %.3f ns per wave-instr per SIMD (%.2f cyc @2.4GHz), %.1f T lane-ops/s\n",
           name, wavesPerSimd, ms, ns_per_instr, ns_per_instr * 2.4, tlops);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {4, 8}) {
        run<0>("v_mul_f32", w, 8, 8, d);
        run<1>("v_fma_f32", w, 8, 8, d);
        run<2>("v_pk_mul_f32", w, 4, 8, d);
        run<3>("v_pk_fma_f32", w, 4, 8, d);
        run<4>("v_mul_f32 (sgpr src)", w, 8, 8, d);
        run<5>("v_sqrt_f32", w, 8, 8, d);
        run<6>("v_mul_lo_u32", w, 8, 8, d);
        run<7>("v_cmp+v_cndmask", w, 8, 8, d);
        run<8>("v_mad_u64_u32 (+8 cvt/xor)", w, 4, 4, d);
        run<9>("v_mul_hi_u32", w, 8, 8, d);
        run<10>("v_fma_f64 (+cvts)", w, 8, 8, d);
        run<11>("v_mad_u64_u32 (+few)", w, 8, 8, d);
        run<12>("v_rcp_f64 (+few)", w, 8, 8, d);
        run<13>("v_mul_f64 (+few)", w, 8, 8, d);
        run<14>("v_rcp_f32", w, 8, 8, d);
        run<15>("v_add_f64 (+few)", w, 8, 8, d);
    }
    return 0;
}
