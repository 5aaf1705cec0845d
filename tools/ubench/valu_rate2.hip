// Microbenchmark 2: what sets the issue cost of a VALU instruction on gfx950 -- encoding size (VOP2 = 4 bytes, VOP3 = 8),
// number of VGPR source operands, or the operation?  8 waves per SIMD, independent instructions, as valu_rate.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s, float t)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, ss = {s, s}, tt = {t, t};
    for (int i = 0; i < iters; ++i) {
#define A8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(t)
        if (MODE == 0)   // VOP2 v_mul_f32, 2 VGPR sources
            asm volatile("v_mul_f32_e32 %0, %0, %8\n v_mul_f32_e32 %1, %1, %8\n v_mul_f32_e32 %2, %2, %8\n v_mul_f32_e32 %3, %3, %8\n"
                         "v_mul_f32_e32 %4, %4, %8\n v_mul_f32_e32 %5, %5, %8\n v_mul_f32_e32 %6, %6, %8\n v_mul_f32_e32 %7, %7, %8" A8);
        else if (MODE == 1)   // the same multiply in the 8-byte VOP3 encoding
            asm volatile("v_mul_f32_e64 %0, %0, %8\n v_mul_f32_e64 %1, %1, %8\n v_mul_f32_e64 %2, %2, %8\n v_mul_f32_e64 %3, %3, %8\n"
                         "v_mul_f32_e64 %4, %4, %8\n v_mul_f32_e64 %5, %5, %8\n v_mul_f32_e64 %6, %6, %8\n v_mul_f32_e64 %7, %7, %8" A8);
        else if (MODE == 2)   // VOP2 v_add_f32
            asm volatile("v_add_f32_e32 %0, %0, %8\n v_add_f32_e32 %1, %1, %8\n v_add_f32_e32 %2, %2, %8\n v_add_f32_e32 %3, %3, %8\n"
                         "v_add_f32_e32 %4, %4, %8\n v_add_f32_e32 %5, %5, %8\n v_add_f32_e32 %6, %6, %8\n v_add_f32_e32 %7, %7, %8" A8);
        else if (MODE == 3)   // VOP2 v_fmac_f32: 4-byte encoding, three VGPR reads (dst is a source)
            asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                         "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9" A8);
        else if (MODE == 4)   // VOP3 v_fma_f32, three distinct VGPR sources
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" A8);
        else if (MODE == 5)   // VOP3 v_fma_f32 with an inline constant as third source (two VGPR reads)
            asm volatile("v_fma_f32 %0, %0, %8, 1.0\n v_fma_f32 %1, %1, %8, 1.0\n v_fma_f32 %2, %2, %8, 1.0\n v_fma_f32 %3, %3, %8, 1.0\n"
                         "v_fma_f32 %4, %4, %8, 1.0\n v_fma_f32 %5, %5, %8, 1.0\n v_fma_f32 %6, %6, %8, 1.0\n v_fma_f32 %7, %7, %8, 1.0" A8);
        else if (MODE == 6)   // v_max3_f32 (VOP3, three reads)
            asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9" A8);
        else if (MODE == 7)   // v_mov_b32
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                         "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" A8);
        else if (MODE == 8)   // v_xor_b32 (VOP2 integer)
            asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                         "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" A8);
        else if (MODE == 9)   // v_cndmask_b32 VOP2 (vcc), no compare
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" A8 : "vcc");
        else if (MODE == 10)  // v_cmp_lt_f32 to vcc (VOPC)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" A8 : "vcc");
        else if (MODE == 11)  // v_pk_mul_f32 (4 instructions = 8 lane-multiplies)
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss), "v"(tt));
        else if (MODE == 12)  // v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss), "v"(tt));
        else if (MODE == 13)  // v_pk_fma_f32, three distinct sources
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss), "v"(tt));
        else if (MODE == 14)  // VOP2 v_mul_f32 with both sources the same bank-distance apart (src0 = src1)
            asm volatile("v_mul_f32_e32 %0, %0, %0\n v_mul_f32_e32 %1, %1, %1\n v_mul_f32_e32 %2, %2, %2\n v_mul_f32_e32 %3, %3, %3\n"
                         "v_mul_f32_e32 %4, %4, %4\n v_mul_f32_e32 %5, %5, %5\n v_mul_f32_e32 %6, %6, %6\n v_mul_f32_e32 %7, %7, %7" A8);
        else if (MODE == 15)  // v_mul_lo_u32 + v_mul_hi_u32 mix (Philox's pair)
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" A8);
        else if (MODE == 16)  // s_and_saveexec-style scalar work between VALU: 4 v_mul + 4 s_and_b64
            asm volatile("v_mul_f32_e32 %0, %0, %8\n s_and_b64 s[20:21], s[20:21], exec\n v_mul_f32_e32 %1, %1, %8\n s_and_b64 s[22:23], s[22:23], exec\n"
                         "v_mul_f32_e32 %2, %2, %8\n s_and_b64 s[20:21], s[20:21], exec\n v_mul_f32_e32 %3, %3, %8\n s_and_b64 s[22:23], s[22:23], exec" A8 : "s20", "s21", "s22", "s23", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char *name, int wavesPerSimd, int instrPerIter, float *d)
{
    const int iters = 40000;
    const int blocks = 256 * wavesPerSimd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1000, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * instrPerIter * wavesPerSimd);
    printf("%-44s waves/SIMD %d: %.3f ns per wave-instr per SIMD = %.2f cycles @2.4 GHz\n", name, wavesPerSimd, ns, ns * 2.4);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {4, 8}) {
        run<0>("v_mul_f32 VOP2 (4 B, 2 reads)", w, 8, d);
        run<1>("v_mul_f32 VOP3 (8 B, 2 reads)", w, 8, d);
        run<2>("v_add_f32 VOP2", w, 8, d);
        run<3>("v_fmac_f32 VOP2 (4 B, 3 reads)", w, 8, d);
        run<4>("v_fma_f32 VOP3 (8 B, 3 reads)", w, 8, d);
        run<5>("v_fma_f32 VOP3 (8 B, 2 reads + const)", w, 8, d);
        run<6>("v_max3_f32 VOP3 (3 reads)", w, 8, d);
        run<7>("v_mov_b32", w, 8, d);
        run<8>("v_xor_b32 VOP2", w, 8, d);
        run<9>("v_cndmask_b32 VOP2", w, 8, d);
        run<10>("v_cmp_lt_f32 VOPC", w, 8, d);
        run<11>("v_pk_mul_f32", w, 4, d);
        run<12>("v_pk_add_f32", w, 4, d);
        run<13>("v_pk_fma_f32 (3 distinct sources)", w, 4, d);
        run<14>("v_mul_f32 VOP2 x*x", w, 8, d);
        run<15>("v_mul_lo_u32 / v_mul_hi_u32", w, 8, d);
        run<16>("v_mul_f32 + s_and_b64 alternating", w, 8, d);
    }
    return 0;
}
