// Microbenchmark 3: what scalar instructions, s_nop padding and exec-mask branches cost a SIMD that is otherwise busy with
// VALU work (gfx950; 4 and 8 waves per SIMD).  Every mode issues the same 8 independent v_fma_f32 per iteration plus extras.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s, float t)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
#define A8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(t)
#define F(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
        if (MODE == 0)        // 8 VALU
            asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) A8);
        else if (MODE == 1)   // 8 VALU + 4 s_add_u32 (plain SALU, no exec)
            asm volatile(F(0) "s_add_u32 s20, s20, 1\n" F(1) F(2) "s_add_u32 s21, s21, 1\n" F(3) F(4) "s_add_u32 s22, s22, 1\n" F(5) F(6) "s_add_u32 s23, s23, 1\n" F(7)
                         A8 : "s20", "s21", "s22", "s23", "scc");
        else if (MODE == 2)   // 8 VALU + 8 SALU
            asm volatile(F(0) "s_add_u32 s20, s20, 1\n" F(1) "s_add_u32 s21, s21, 1\n" F(2) "s_add_u32 s22, s22, 1\n" F(3) "s_add_u32 s23, s23, 1\n"
                         F(4) "s_add_u32 s20, s20, 1\n" F(5) "s_add_u32 s21, s21, 1\n" F(6) "s_add_u32 s22, s22, 1\n" F(7) "s_add_u32 s23, s23, 1\n"
                         A8 : "s20", "s21", "s22", "s23", "scc");
        else if (MODE == 3)   // 8 VALU + 8 SALU in one run behind the VALU
            asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
                         "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         A8 : "s20", "s21", "s22", "s23", "scc");
        else if (MODE == 4)   // 8 VALU + 4 x s_nop 1
            asm volatile(F(0) "s_nop 1\n" F(1) F(2) "s_nop 1\n" F(3) F(4) "s_nop 1\n" F(5) F(6) "s_nop 1\n" F(7) A8);
        else if (MODE == 5)   // 8 VALU + 2 x (s_and_saveexec_b64 / s_or_b64 exec) with an all-ones mask: the if/endif pair of a divergent branch
            asm volatile("s_mov_b64 s[22:23], -1\n"
                         F(0) "s_and_saveexec_b64 s[20:21], s[22:23]\n" F(1) F(2) "s_or_b64 exec, exec, s[20:21]\n" F(3)
                         F(4) "s_and_saveexec_b64 s[20:21], s[22:23]\n" F(5) F(6) "s_or_b64 exec, exec, s[20:21]\n" F(7)
                         A8 : "s20", "s21", "s22", "s23", "scc");
        else if (MODE == 6)   // 8 VALU + 2 x (v_cmp to vcc, s_and_saveexec vcc, s_cbranch_execz (not taken), ..., s_or exec): a full if-block
            asm volatile(F(0) "v_cmp_lt_f32 vcc, %8, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 0\n" F(1) F(2) "s_or_b64 exec, exec, s[20:21]\n" F(3)
                         F(4) "v_cmp_lt_f32 vcc, %8, %4\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 0\n" F(5) F(6) "s_or_b64 exec, exec, s[20:21]\n" F(7)
                         A8 : "s20", "s21", "vcc", "scc");
        else if (MODE == 7)   // 8 VALU + 4 v_cmp/s_nop 1/v_cndmask triples (the compiler's select)
            asm volatile(F(0) "v_cmp_lt_f32 vcc, %8, %0\n s_nop 1\n v_cndmask_b32 %1, %1, %9, vcc\n" F(1) F(2) "v_cmp_lt_f32 vcc, %8, %2\n s_nop 1\n v_cndmask_b32 %3, %3, %9, vcc\n" F(3)
                         F(4) "v_cmp_lt_f32 vcc, %8, %4\n s_nop 1\n v_cndmask_b32 %5, %5, %9, vcc\n" F(5) F(6) "v_cmp_lt_f32 vcc, %8, %6\n s_nop 1\n v_cndmask_b32 %7, %7, %9, vcc\n" F(7)
                         A8 : "vcc");
        else if (MODE == 8)   // the same selects without the s_nop (hardware interlock)
            asm volatile(F(0) "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %1, %1, %9, vcc\n" F(1) F(2) "v_cmp_lt_f32 vcc, %8, %2\n v_cndmask_b32 %3, %3, %9, vcc\n" F(3)
                         F(4) "v_cmp_lt_f32 vcc, %8, %4\n v_cndmask_b32 %5, %5, %9, vcc\n" F(5) F(6) "v_cmp_lt_f32 vcc, %8, %6\n v_cndmask_b32 %7, %7, %9, vcc\n" F(7)
                         A8 : "vcc");
        else if (MODE == 9)   // 8 VALU + 4 s_waitcnt lgkmcnt(0) with nothing outstanding
            asm volatile(F(0) "s_waitcnt lgkmcnt(0)\n" F(1) F(2) "s_waitcnt lgkmcnt(0)\n" F(3) F(4) "s_waitcnt lgkmcnt(0)\n" F(5) F(6) "s_waitcnt lgkmcnt(0)\n" F(7) A8);
        else if (MODE == 10)  // 8 VALU + 4 ballots: v_cmp to an SGPR pair + s_bcnt1 (the kernels' wave votes)
            asm volatile(F(0) "v_cmp_lt_f32 s[20:21], %8, %0\n s_bcnt1_i32_b64 s22, s[20:21]\n" F(1) F(2) "v_cmp_lt_f32 s[20:21], %8, %2\n s_bcnt1_i32_b64 s22, s[20:21]\n" F(3)
                         F(4) "v_cmp_lt_f32 s[20:21], %8, %4\n s_bcnt1_i32_b64 s22, s[20:21]\n" F(5) F(6) "v_cmp_lt_f32 s[20:21], %8, %6\n s_bcnt1_i32_b64 s22, s[20:21]\n" F(7)
                         A8 : "s20", "s21", "s22", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char *name, int wavesPerSimd, float base_ns, float *d, float *ret = nullptr)
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
    const double ns_iter = ms * 1e6 / ((double)iters * wavesPerSimd);       // SIMD time per wave-iteration
    if (ret) *ret = (float)ns_iter;
    printf("%-64s waves/SIMD %d: %.2f ns = %.1f cycles per iteration per SIMD  (extras: +%.1f cycles)\n", name, wavesPerSimd, ns_iter, ns_iter * 2.4,
           (ns_iter - base_ns) * 2.4);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {4, 8}) {
        float b = 0;
        run<0>("8 v_fma_f32", w, 0, d, &b);
        run<1>("8 v_fma_f32 + 4 s_add_u32 interleaved", w, b, d);
        run<2>("8 v_fma_f32 + 8 s_add_u32 interleaved", w, b, d);
        run<3>("8 v_fma_f32 + 8 s_add_u32 in one run", w, b, d);
        run<4>("8 v_fma_f32 + 4 s_nop 1", w, b, d);
        run<5>("8 v_fma_f32 + 2 (s_and_saveexec, s_or exec)", w, b, d);
        run<6>("8 v_fma_f32 + 2 (v_cmp, s_and_saveexec, s_cbranch_execz, s_or exec)", w, b, d);
        run<7>("8 v_fma_f32 + 4 (v_cmp, s_nop 1, v_cndmask)", w, b, d);
        run<8>("8 v_fma_f32 + 4 (v_cmp, v_cndmask)", w, b, d);
        run<9>("8 v_fma_f32 + 4 s_waitcnt lgkmcnt(0)", w, b, d);
        run<10>("8 v_fma_f32 + 4 (v_cmp to sgpr pair, s_bcnt1)", w, b, d);
    }
    return 0;
}
