// queue_probe: does "a fifth busy compute queue slows every kernel of every queue" (MEASUREMENTS R5 item 8) exist in a BARE process --
// no torch, no library of ours, nothing but HIP streams?  
//
// K streams, each a chain of dependent launches of one latency-bound kernel (one workgroup chases pointers through a 1 MB table:
// ~`hops` dependent L2 hits, nothing else -- what a step's launches look like).  Per stream count K we report the kernel's OWN duration
// (its first thread stamps the 100 MHz counter at start and end: no tracer, no events), the time per launch of a chain (wall), and the
// hardware queue every stream landed on as far as the kernel can see it (HW_ID's queue / pipe / me fields).
//   usage: queue_probe [max_streams = 8] [launches = 400] [hops = 64] [idle = 0|1]
//   idle = 1: streams beyond the first four are created and used ONCE (one launch), then sit idle while the first four run the chains
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/queue_probe tools/queue_probe.hip ; run with GPU_MAX_HW_QUEUES unset / 4 / 8 / 16.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter() * 0ull + wall_clock64(); }

__global__ __launch_bounds__(256) void k_chase(const unsigned *__restrict__ tab, unsigned mask, int hops, unsigned start, unsigned long long *stamps,
                                                unsigned *hw, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned p = (start + threadIdx.x * 977u) & mask;
    for (int i = 0; i < hops; ++i) p = tab[p];
    if (p == 0xFFFFFFFFu) *sink = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[0] = t0;
        stamps[1] = wall_clock64();
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));  // gfx9: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [15:13] se, [23:20] vmid, [26:24] queue, [31:30] me
        *hw = id;
    }
}

int main(int argc, char **argv) {
    const int max_streams = argc > 1 ? atoi(argv[1]) : 8;
    const int launches = argc > 2 ? atoi(argv[2]) : 400;
    const int hops = argc > 3 ? atoi(argv[3]) : 64;
    const int idle_mode = argc > 4 ? atoi(argv[4]) : 0;
    const unsigned N = 1u << 18;  // 1 MB table
    std::vector<unsigned> h(N);
    unsigned x = 12345u;
    for (unsigned i = 0; i < N; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = (x >> 8) & (N - 1);
    }
    unsigned *tab, *sink;
    hipMalloc(&tab, N * sizeof(unsigned));
    hipMalloc(&sink, 64);
    hipMemcpy(tab, h.data(), N * sizeof(unsigned), hipMemcpyHostToDevice);
    const char *env = getenv("GPU_MAX_HW_QUEUES");
    printf("queue_probe: GPU_MAX_HW_QUEUES=%s, %d launches per chain, %d hops, idle mode %d\n", env ? env : "(unset)", launches, hops, idle_mode);
    std::vector<hipStream_t> st(max_streams);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long *stamps;
    unsigned *hw;
    hipHostMalloc(&stamps, sizeof(unsigned long long) * 2 * max_streams * launches);
    hipHostMalloc(&hw, sizeof(unsigned) * max_streams * launches);
    for (int K = 1; K <= max_streams; ++K) {
        const int busy = idle_mode ? std::min(K, 4) : K;
        if (idle_mode && K > 4)  // the extra streams have been used once: their hardware queue exists, and stays idle
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(256), 0, st[K - 1], tab, N - 1, 1, 0u, stamps, hw, sink);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < launches; ++i)
            for (int s = 0; s < busy; ++s)
                hipLaunchKernelGGL(k_chase, dim3(1), dim3(256), 0, st[s], tab, N - 1, hops, (unsigned)(i * 131 + s * 7), stamps + 2 * (s * launches + i),
                                   hw + s * launches + i, sink);
        hipDeviceSynchronize();
        const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("K = %d streams (%d busy): %.2f us per launch of a chain (wall);", K, busy, wall_us / launches);
        for (int s = 0; s < busy; ++s) {
            std::vector<double> d;
            for (int i = launches / 4; i < launches; ++i) d.push_back((double)(stamps[2 * (s * launches + i) + 1] - stamps[2 * (s * launches + i)]) * 0.01);
            std::sort(d.begin(), d.end());
            const unsigned id = hw[s * launches + launches - 1];
            // the stream's own rate on the device clock: first start to last end of ITS chain
            const double span = (double)(stamps[2 * (s * launches + launches - 1) + 1] - stamps[2 * (s * launches)]) * 0.01 / launches;
            printf("  s%d: %.1f us/launch, kernel %.1f us median / %.1f p90, me %u pipe %u queue %u;", s, span, d[d.size() / 2], d[d.size() * 9 / 10], id >> 30,
                   (id >> 6) & 3u, (id >> 24) & 7u);
        }
        printf("\n");
    }
    // ---- what an event costs a chain of dependent launches (round 6): between two launches of ONE stream, (a) nothing, (b) an event record,
    // (c) a wait for an event another stream recorded long ago (complete when hipStreamWaitEvent is called: the runtime drops it), (d) both,
    // (e) a LIVE join: the other stream records behind a kernel that ends at about the same time; with 4 x longer kernels: (f) nothing,
    // (g) a wait that is enqueued before its event is complete but satisfied when the queue gets to it, (h) the same + a record ----
    {
        hipEvent_t ev[2], evl;
        hipEventCreateWithFlags(&ev[0], hipEventDisableTiming | hipEventDisableSystemFence);
        hipEventCreateWithFlags(&ev[1], hipEventDisableTiming | hipEventDisableSystemFence);
        hipEventCreateWithFlags(&evl, hipEventDisableTiming | hipEventDisableSystemFence);
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(256), 0, st[1], tab, N - 1, 1, 0u, stamps, hw, sink);
        hipEventRecord(ev[1], st[1]);
        hipDeviceSynchronize();
        const char *names[8] = {"nothing", "event record", "wait on a signalled event", "record + signalled wait", "live join with a second stream",
                                "(long kernels) nothing", "(long kernels) wait enqueued early, satisfied when reached", "(long kernels) the same + a record"};
        for (int mode = 0; mode < 8; ++mode) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < launches; ++i) {
                if (mode >= 6) {  // the other stream: a short kernel and its event, long before this stream gets to the wait
                    hipLaunchKernelGGL(k_chase, dim3(1), dim3(256), 0, st[1], tab, N - 1, 4, (unsigned)(i * 7), stamps + 2 * (launches + i), hw + launches + i, sink);
                    hipEventRecord(evl, st[1]);
                }
                hipLaunchKernelGGL(k_chase, dim3(1), dim3(256), 0, st[0], tab, N - 1, mode >= 5 ? 4 * hops : hops, (unsigned)(i * 131), stamps + 2 * i, hw + i, sink);
                if (mode >= 6) hipStreamWaitEvent(st[0], evl, 0);
                if (mode == 7) hipEventRecord(ev[0], st[0]);
                if (mode == 1 || mode == 3) hipEventRecord(ev[0], st[0]);
                if (mode == 2 || mode == 3) hipStreamWaitEvent(st[0], ev[1], 0);
                if (mode == 4) {
                    hipLaunchKernelGGL(k_chase, dim3(1), dim3(256), 0, st[1], tab, N - 1, hops, (unsigned)(i * 7), stamps + 2 * (launches + i), hw + launches + i, sink);
                    hipEventRecord(evl, st[1]);
                    hipStreamWaitEvent(st[0], evl, 0);
                }
            }
            hipDeviceSynchronize();
            const double span = (double)(stamps[2 * (launches - 1) + 1] - stamps[0]) * 0.01 / launches;
            printf("between two launches of a chain: %-62s %.2f us per launch (device clock), %.2f (wall)\n", names[mode], span,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches);
        }
    }
    return 0;
}
