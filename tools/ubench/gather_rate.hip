// How fast does a CU gather 16 bytes per lane from divergent addresses?  (dev aid; DESIGN.md, triangle kernel's bound)
// Every lane loads float4 at pseudo-random 16-byte-aligned offsets inside a window of W bytes; 16 waves per CU.
//   throughput mode: 4 independent loads in flight per lane, addresses from an LCG
//   chain mode:      the next address depends on the loaded value (one load in flight per lane)
// Build: hipcc --offload-arch=gfx950 -O2 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// the same with 4 or 8 bytes per lane (is the L1's limit requests or bytes?); addresses stay 16-byte aligned
template <class V>
__global__ __launch_bounds__(256) void gather_narrow(const float4 *buf, uint32_t mask, uint32_t iters, float *out, unsigned long long *cycles)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; ++i) {
        const uint32_t a = s, b = s * 1664525u + 1013904223u, c = b * 1664525u + 1013904223u, d = c * 1664525u + 1013904223u;
        const V v0 = *(const V *)&buf[(a >> 4) & mask], v1 = *(const V *)&buf[(b >> 4) & mask],
                v2 = *(const V *)&buf[(c >> 4) & mask], v3 = *(const V *)&buf[(d >> 4) & mask];
        acc += v0.x + v1.x + v2.x + v3.x;
        s = d * 1664525u + 1013904223u;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 123.456f) out[0] = acc;
    if (threadIdx.x == 0) atomicMax(cycles, t1 - t0);
}

template <bool CHAIN, bool COHERENT>
__global__ __launch_bounds__(256) void gather(const float4 *buf, uint32_t mask, uint32_t iters, float *out, unsigned long long *cycles)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    if (COHERENT) s = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u + 12345u;   // one address per wave (broadcast)
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; ++i) {
        if (CHAIN) {
            const float4 v = buf[s & mask];
            acc += v.y;
            s = s * 1664525u + 1013904223u + __float_as_uint(v.x);
        } else {
            const uint32_t a = s, b = s * 1664525u + 1013904223u, c = b * 1664525u + 1013904223u, d = c * 1664525u + 1013904223u;
            const float4 v0 = buf[(a >> 4) & mask], v1 = buf[(b >> 4) & mask], v2 = buf[(c >> 4) & mask], v3 = buf[(d >> 4) & mask];
            acc += v0.x + v1.y + v2.z + v3.w;
            s = d * 1664525u + 1013904223u;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 123.456f) out[0] = acc;
    if (threadIdx.x == 0) atomicMax(cycles, t1 - t0);
}

int main()
{
    hipDeviceProp_t p;
    if (hipSetDevice(0) != hipSuccess || hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int ncu = p.multiProcessorCount;
    const size_t bytes = 64u << 20;
    float4 *buf; float *out; unsigned long long *cyc;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
    std::vector<float> h(bytes / 4);
    for (size_t k = 0; k < h.size(); ++k) h[k] = (float)(k * 2654435761u >> 8);
    hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice);
    const uint32_t iters = 2000;
    const size_t windows[] = { 8u << 10, 256u << 10, 2u << 20, 24u << 20, 64u << 20 };
    printf("%s, %d CUs, 16 waves per CU, float4 per lane\n", p.name, ncu);
    for (int mode = 0; mode < 5; ++mode) {
        for (size_t w : windows) {
            const uint32_t mask = (uint32_t)(w / 16) - 1u;    // (24 MB is not a power of two: mask of 16 MB + wrap is close enough)
            uint32_t m = 1; while ((size_t)(m + 1) * 16 <= w) m = m * 2 + 1; 
            const uint32_t use = (w & (w - 1)) ? m : mask;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(cyc, 0, 8);
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((gather<false, false>), dim3(ncu * 4), dim3(256), 0, 0, buf, use, iters, out, cyc);
                if (mode == 1) hipLaunchKernelGGL((gather<true, false>), dim3(ncu * 4), dim3(256), 0, 0, buf, use, iters, out, cyc);
                if (mode == 2) hipLaunchKernelGGL((gather<false, true>), dim3(ncu * 4), dim3(256), 0, 0, buf, use, iters, out, cyc);
                if (mode == 3) hipLaunchKernelGGL((gather_narrow<float2>), dim3(ncu * 4), dim3(256), 0, 0, buf, use, iters, out, cyc);
                if (mode == 4) hipLaunchKernelGGL((gather_narrow<float1>), dim3(ncu * 4), dim3(256), 0, 0, buf, use, iters, out, cyc);
                hipEventRecord(e1);
                if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double loads_per_cu = 16.0 * iters * (mode == 1 ? 1 : 4);          // wave-instructions per CU
            printf("%-34s window %8zu KB: %8.3f ms, %6.1f clk per wave-load per CU, %5.2f lanes/clk/CU, %7.1f GB/s chip (useful)\n",
                   mode == 0 ? "divergent, 4 loads in flight" : mode == 1 ? "divergent, dependent chain" : mode == 2 ? "wave-uniform address, 4 in flight" : mode == 3 ? "divergent, 8 B per lane" : "divergent, 4 B per lane",
                   w >> 10, ms, (double)c / loads_per_cu, loads_per_cu * 64 / (double)c, loads_per_cu * ncu * 64 * 16 / (ms * 1e6));
        }
    }
    return 0;
}
