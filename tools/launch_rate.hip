// host-side launch rate probe: how long does hipLaunchKernelGGL take per call on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_null(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
struct Big { char b[256]; };
__global__ void k_big(Big b, int *p) { if (p && threadIdx.x == 9999) *p = b.b[3]; }
int wide_main();
int main() {
    wide_main();
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 3; ++rep) {
        for (int n : {10, 100, 1000}) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s, (int *)nullptr);
            auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(s);
            auto t2 = std::chrono::steady_clock::now();
            printf("null x%d: enqueue %.2f us/launch, total %.2f us/launch\n", n, std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
        }
        Big b{};
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b, (int *)nullptr);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        printf("big  x1000: enqueue %.2f us/launch, total %.2f us/launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 1000,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / 1000);
    }
    // graph of 100 null kernels
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s, (int *)nullptr);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        hipGraphLaunch(ge, s);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        printf("graph(100 null): launch %.2f us, total %.2f us (%.2f us/node)\n", std::chrono::duration<double, std::micro>(t1 - t0).count(),
               std::chrono::duration<double, std::micro>(t2 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t0).count() / 100);
    }
    return 0;
}
// (appended) dispatch cost of wide grids of big workgroups that exit at once
__global__ __launch_bounds__(1024) void k_exit(const int *flag) { if (flag[0] == 12345) ((volatile int *)flag)[1] = 1; }
__global__ __launch_bounds__(1024) void k_exit_lds(const int *flag) {
    __shared__ int big[15000];
    if (flag[0] == 12345) { big[threadIdx.x] = 1; ((volatile int *)flag)[1] = big[5]; }
}
int wide_main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    for (int grid : {64, 256, 1024, 2160, 8192}) {
        for (int lds = 0; lds < 2; ++lds) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 200; ++i) {
                if (lds) hipLaunchKernelGGL(k_exit_lds, dim3(grid), dim3(1024), 0, s, (const int *)d);
                else hipLaunchKernelGGL(k_exit, dim3(grid), dim3(1024), 0, s, (const int *)d);
            }
            hipStreamSynchronize(s);
            auto t2 = std::chrono::steady_clock::now();
            printf("exit kernel grid %5d x 1024 thr, lds %d: %.2f us/launch\n", grid, lds * 60000, std::chrono::duration<double, std::micro>(t2 - t0).count() / 200);
        }
    }
    return 0;
}
