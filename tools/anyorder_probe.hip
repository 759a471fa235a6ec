// does hipExtAnyOrderLaunch let a kernel start beside its predecessor on the SAME stream (gfx950)?  A spins ~200 us on 8 workgroups; B
// (any-order) and C (plain) stamp the 100 MHz clock.  B starting ~0 us after A's start = works; ~200 us = the flag is ignored.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_spin(unsigned long long ticks, unsigned long long *t) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
__global__ void k_stamp(unsigned long long *t, int i) {
    if (threadIdx.x == 0 && blockIdx.x == 0) t[i] = wall_clock64();
}
int main() {
    unsigned long long *t;
    hipMalloc(&t, 64);
    hipMemset(t, 0, 64);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, t, 7);
        hipStreamSynchronize(s);
        hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, s, 20000ull, t);
        hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, t, 2);
        hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, t, 3);
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, t, 4);
        hipStreamSynchronize(s);
        unsigned long long h[8];
        hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("A start 0, A end %.1f us, any-order B at %.1f us, any-order B2 at %.1f us, ordered C at %.1f us\n", (h[1] - h[0]) * 0.01, ((double)h[2] - (double)h[0]) * 0.01,
               ((double)h[3] - (double)h[0]) * 0.01, ((double)h[4] - (double)h[0]) * 0.01);
    }
    return 0;
}
