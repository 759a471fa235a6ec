// bw_probe.hip — variants of the VoI membership pass, to see what bounds k_voi_split on MI355X.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) 8 x float2 per lane, f64 test, ballots (the current kernel's O path)
__global__ __launch_bounds__(256) void v_f2x8_f64(const float2* __restrict__ xy, uint32_t n, double xc, double yc, double r2, unsigned long long* m, uint32_t* ci) {
    const uint32_t lane = threadIdx.x & 63u, wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nch = n / 512;
    for (uint32_t c = wid; c < nch; c += nw) {
        float2 p[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) p[t] = xy[c * 512 + t * 64 + lane];
        unsigned long long my = 0; uint32_t cv = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const double dx = (double)p[t].x - xc, dy = (double)p[t].y - yc;
            const bool in = (__float_as_uint(p[t].x) != 0xFFC0DEADu) && (dx * dx + dy * dy < r2);
            const unsigned long long b = __ballot(in);
            if ((int)lane == t) my = b;
            cv += __popcll(b);
        }
        if (lane < 8) m[(size_t)c * 8 + lane] = my;
        if (lane == 0) ci[c] = cv;
    }
}
// (b) same with f32 arithmetic (not exact; isolates the f64 cost)
__global__ __launch_bounds__(256) void v_f2x8_f32(const float2* __restrict__ xy, uint32_t n, float xc, float yc, float r2, unsigned long long* m, uint32_t* ci) {
    const uint32_t lane = threadIdx.x & 63u, wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nch = n / 512;
    for (uint32_t c = wid; c < nch; c += nw) {
        float2 p[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) p[t] = xy[c * 512 + t * 64 + lane];
        unsigned long long my = 0; uint32_t cv = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float dx = p[t].x - xc, dy = p[t].y - yc;
            const bool in = dx * dx + dy * dy < r2;
            const unsigned long long b = __ballot(in);
            if ((int)lane == t) my = b;
            cv += __popcll(b);
        }
        if (lane < 8) m[(size_t)c * 8 + lane] = my;
        if (lane == 0) ci[c] = cv;
    }
}
// (c) float4 x U per lane (2 points per lane), f64, ballots WITHOUT interleave (masks in lane order)
template <int U>
__global__ __launch_bounds__(256) void v_f4_f64(const float4* __restrict__ xy4, uint32_t n, double xc, double yc, double r2, unsigned long long* m, uint32_t* ci) {
    const uint32_t lane = threadIdx.x & 63u, wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const uint32_t per = 128 * U;  // points per wave-iteration
    const uint32_t nch = n / per;
    for (uint32_t c = wid; c < nch; c += nw) {
        float4 q[U];
#pragma unroll
        for (int g = 0; g < U; ++g) q[g] = xy4[(size_t)c * (per / 2) + g * 64 + lane];
        unsigned long long my = 0; uint32_t cv = 0;
#pragma unroll
        for (int g = 0; g < U; ++g) {
            const double dx0 = (double)q[g].x - xc, dy0 = (double)q[g].y - yc, dx1 = (double)q[g].z - xc, dy1 = (double)q[g].w - yc;
            const bool in0 = (__float_as_uint(q[g].x) != 0xFFC0DEADu) && (dx0 * dx0 + dy0 * dy0 < r2);
            const bool in1 = (__float_as_uint(q[g].z) != 0xFFC0DEADu) && (dx1 * dx1 + dy1 * dy1 < r2);
            const unsigned long long b0 = __ballot(in0), b1 = __ballot(in1);
            if ((int)lane == 2 * g) my = b0;
            if ((int)lane == 2 * g + 1) my = b1;
            cv += __popcll(b0) + __popcll(b1);
        }
        if (lane < 2 * U) m[(size_t)c * 2 * U + lane] = my;
        if (lane == 0) ci[c] = cv;
    }
}
// (d) read-only ceiling: float4 loads, trivial reduction
__global__ __launch_bounds__(256) void v_read4(const float4* __restrict__ xy4, uint32_t n4, float* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    float acc = 0.f;
    for (uint32_t i = tid; i + 3 * nt < n4; i += 4 * nt) {
        const float4 a = xy4[i], b = xy4[i + nt], c = xy4[i + 2 * nt], d = xy4[i + 3 * nt];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// (e) float2 x 16 per lane (more bytes in flight), f64
__global__ __launch_bounds__(256) void v_f2x16_f64(const float2* __restrict__ xy, uint32_t n, double xc, double yc, double r2, unsigned long long* m, uint32_t* ci) {
    const uint32_t lane = threadIdx.x & 63u, wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nch = n / 1024;
    for (uint32_t c = wid; c < nch; c += nw) {
        float2 p[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) p[t] = xy[c * 1024 + t * 64 + lane];
        unsigned long long my = 0; uint32_t cv = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const double dx = (double)p[t].x - xc, dy = (double)p[t].y - yc;
            const bool in = (__float_as_uint(p[t].x) != 0xFFC0DEADu) && (dx * dx + dy * dy < r2);
            const unsigned long long b = __ballot(in);
            if ((int)lane == t) my = b;
            cv += __popcll(b);
        }
        if (lane < 16) m[(size_t)c * 16 + lane] = my;
        if (lane == 0) ci[c] = cv;
    }
}

template <class F> double timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 40u * 1024 * 1024;  // points
    float2* xy; unsigned long long* m; uint32_t* ci; float* out;
    CK(hipMalloc(&xy, (size_t)n * 8)); CK(hipMalloc(&m, (size_t)n / 8 + 64)); CK(hipMalloc(&ci, (size_t)n / 64 + 64)); CK(hipMalloc(&out, 64));
    std::vector<float2> h(n);
    for (uint32_t i = 0; i < n; ++i) { h[i].x = (float)((i * 2654435761u) % 2000000) * 1e-3f; h[i].y = (float)((i * 40503u) % 400000) * 1e-3f; }
    CK(hipMemcpy(xy, h.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    const double GB = (double)n * 8 / 1e9;
    for (int grid : {1024, 2048, 4096}) {
        double t;
        t = timeit([&] { hipLaunchKernelGGL(v_f2x8_f64, dim3(grid), dim3(256), 0, 0, xy, n, 500.0, 100.0, 6400.0, m, ci); }, 20);
        printf("grid %4d  f2x8  f64        : %7.1f us  %6.1f GB/s\n", grid, t * 1e3, GB / (t * 1e-3));
        t = timeit([&] { hipLaunchKernelGGL(v_f2x8_f32, dim3(grid), dim3(256), 0, 0, xy, n, 500.0f, 100.0f, 6400.0f, m, ci); }, 20);
        printf("grid %4d  f2x8  f32        : %7.1f us  %6.1f GB/s\n", grid, t * 1e3, GB / (t * 1e-3));
        t = timeit([&] { hipLaunchKernelGGL(v_f2x16_f64, dim3(grid), dim3(256), 0, 0, xy, n, 500.0, 100.0, 6400.0, m, ci); }, 20);
        printf("grid %4d  f2x16 f64        : %7.1f us  %6.1f GB/s\n", grid, t * 1e3, GB / (t * 1e-3));
        t = timeit([&] { hipLaunchKernelGGL(v_f4_f64<2>, dim3(grid), dim3(256), 0, 0, (const float4*)xy, n, 500.0, 100.0, 6400.0, m, ci); }, 20);
        printf("grid %4d  f4x2  f64 nointl : %7.1f us  %6.1f GB/s\n", grid, t * 1e3, GB / (t * 1e-3));
        t = timeit([&] { hipLaunchKernelGGL(v_f4_f64<4>, dim3(grid), dim3(256), 0, 0, (const float4*)xy, n, 500.0, 100.0, 6400.0, m, ci); }, 20);
        printf("grid %4d  f4x4  f64 nointl : %7.1f us  %6.1f GB/s\n", grid, t * 1e3, GB / (t * 1e-3));
        t = timeit([&] { hipLaunchKernelGGL(v_f4_f64<8>, dim3(grid), dim3(256), 0, 0, (const float4*)xy, n, 500.0, 100.0, 6400.0, m, ci); }, 20);
        printf("grid %4d  f4x8  f64 nointl : %7.1f us  %6.1f GB/s\n", grid, t * 1e3, GB / (t * 1e-3));
        t = timeit([&] { hipLaunchKernelGGL(v_read4, dim3(grid), dim3(256), 0, 0, (const float4*)xy, n / 2, out); }, 20);
        printf("grid %4d  read-only float4 : %7.1f us  %6.1f GB/s\n", grid, t * 1e3, GB / (t * 1e-3));
    }
    return 0;
}
