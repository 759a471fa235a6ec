// ratio of clock64() (what the sort stamps count) to wall_clock64() (constant-rate counter) on this box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *o, int spin) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { o[0] = c1 - c0; o[1] = w1 - w0; o[2] = (unsigned long long)x; }
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 32);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, device clock rate attr %d kHz\n", rate, clk);
    for (int spin : {100000, 1000000, 10000000}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, spin);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("spin %d: clock64 %llu, wall_clock64 %llu -> clock64 ticks at %.1f MHz; %.2f ticks per fma-iteration\n", spin, h[0], h[1],
               (double)h[0] / ((double)h[1] / (rate * 1e3)) / 1e6, (double)h[0] / spin);
    }
    return 0;
}
