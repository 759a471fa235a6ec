// Which XCD does bit i of a stream's CU mask (hipExtStreamCreateWithCUMask) select on this part?  For every candidate rule the probe
// creates a masked stream, launches 512 small workgroups and tallies HW_REG_XCC_ID.
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcd_mask_probe tools/xcd_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_where(unsigned *xcc_hist, unsigned *cu_seen) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        atomicAdd(&xcc_hist[xcc & 15u], 1u);
        const unsigned cu = (hwid >> 8) & 15u, sh = (hwid >> 12) & 1u, se = (hwid >> 13) & 7u;
        atomicOr(&cu_seen[(xcc & 15u) * 8 + se], 1u << (cu + 16 * sh));
    }
    // keep the workgroup alive a little so that the dispatcher has to spread the grid
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 300) {}
}
static void run(const char *what, const std::vector<uint32_t> &mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%-40s stream creation failed\n", what); return; }
    unsigned *d; hipMalloc(&d, (16 + 128) * 4); hipMemsetAsync(d, 0, (16 + 128) * 4, s);
    hipLaunchKernelGGL(k_where, dim3(2048), dim3(256), 0, s, d, d + 16);
    hipStreamSynchronize(s);
    unsigned h[16 + 128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s workgroups per XCC:", what);
    for (int i = 0; i < 8; ++i) printf(" %4u", h[i]);
    int cus = 0; for (int i = 0; i < 128; ++i) cus += __builtin_popcount(h[16 + i]);
    printf("   distinct CUs seen: %d\n", cus);
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    printf("%s: %d CUs\n", p.name, ncu);
    { std::vector<uint32_t> m(words, 0xFFFFFFFFu); run("all bits", m); }
    for (int x = 0; x < 8; x += 3) {  // rule A: bit i -> XCD i % 8
        std::vector<uint32_t> m(words, 0u);
        for (int i = 0; i < ncu; ++i) if (i % 8 == x) m[i / 32] |= 1u << (i % 32);
        char b[64]; snprintf(b, sizeof b, "bits with i %% 8 == %d", x); run(b, m);
    }
    for (int x = 0; x < 8; x += 3) {  // rule B: bit i -> XCD i / (ncu / 8)
        std::vector<uint32_t> m(words, 0u);
        for (int i = 0; i < ncu; ++i) if (i / (ncu / 8) == x) m[i / 32] |= 1u << (i % 32);
        char b[64]; snprintf(b, sizeof b, "bits with i / %d == %d", ncu / 8, x); run(b, m);
    }
    { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < ncu; ++i) if (i % 8 < 4) m[i / 32] |= 1u << (i % 32); run("bits with i % 8 < 4", m); }
    { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < ncu / 2; ++i) m[i / 32] |= 1u << (i % 32); run("the lower half of the bits", m); }
    return 0;
}
