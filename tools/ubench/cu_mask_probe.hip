// Which bits of a HIP CU mask (hipExtStreamCreateWithCUMask) belong to one XCD on this GPU?  Launches a census
// kernel on streams with candidate masks and prints the set of (XCC, SE, SH, CU) the workgroups ran on.
// Build: hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe     (dev aid; DESIGN.md "one XCD")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void census(unsigned *out)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the block resident for a while so that the grid spreads over every CU the mask allows
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000ull) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char *name, const std::vector<uint32_t> &mask)
{
    hipStream_t st = nullptr;
    printf("%s ...\n", name); fflush(stdout);
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int blocks = 2048;
    unsigned *d = nullptr;
    if (hipMalloc(&d, blocks * 8) != hipSuccess) { printf("malloc failed\n"); return; }
    hipLaunchKernelGGL(census, dim3(blocks), dim3(64), 0, st, d);
    if (hipStreamSynchronize(st) != hipSuccess) { printf("sync failed: %s\n", hipGetErrorString(hipGetLastError())); return; }
    std::vector<unsigned> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> cus;      // xcc -> {se, sh, cu}
    for (int b = 0; b < blocks; ++b) cus[h[2 * b]].insert((h[2 * b + 1] >> 8) & 0xff);
    printf("%s:", name);
    size_t total = 0;
    for (auto &kv : cus) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  -> %zu CUs on %zu XCDs\n", total, cus.size());
    hipFree(d); hipStreamDestroy(st);
}

int main()
{
    hipDeviceProp_t p;
    if (hipSetDevice(0) != hipSuccess || hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    printf("%s: %d CUs\n", p.name, ncu);
    std::vector<uint32_t> all(words, 0xffffffffu);
    run("all", all);
    std::vector<uint32_t> first32(words, 0); first32[0] = 0xffffffffu;
    run("bits 0..31", first32);
    for (int x = 0; x < 8; x += 7) {
        std::vector<uint32_t> m(words, 0);
        for (int i = x; i < ncu; i += 8) m[i / 32] |= 1u << (i % 32);
        char nm[64]; snprintf(nm, sizeof nm, "bits i %% 8 == %d", x);
        run(nm, m);
    }
    std::vector<uint32_t> blk(words, 0);
    for (int i = 32; i < 64; ++i) blk[i / 32] |= 1u << (i % 32);
    run("bits 32..63", blk);
    return 0;
}
