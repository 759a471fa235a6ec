// What does a latency-bound kernel on one stream pay for what ANOTHER stream does meanwhile?  ("why can two
// independent sequences not share 256 CUs".)  Victims: (V1) one wavefront chasing dependent loads through a table, timed per hop by
// the constant-rate counter inside the kernel; (V2) a chain of small dependent launches (each reads what the previous one wrote), timed
// by events.  Aggressors on a second stream, in flight for the whole measurement:
//   none | back-to-back EMPTY launches (1 workgroup) | empty launches of 256 workgroups | ONE long ALU-only kernel on every CU |
//   one long ALU-only kernel on 8 workgroups | back-to-back launches that dirty 256 KB each | a long streaming read (HBM traffic) |
//   empty launches on two streams.
// If empty launches hurt as much as anything, the contended resource is the kernel BOUNDARY (the acquire / release of every dispatch:
// L2 write-back + invalidate on all XCDs), not bandwidth, CUs or queues.
//   hipcc --offload-arch=gfx950 -O3 -o tools/interfere_probe tools/interfere_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty() {}
// every stream starts behind a gate the host opens once EVERYTHING is enqueued (the aggressor batch would otherwise be half done before
// the victim is in its queue); gives up after 50 ms so that a full queue can never hang the box
__global__ void k_gate(volatile int *flag) {
    const unsigned long long t0 = wall_clock64();
    while (!*flag && wall_clock64() - t0 < 5000000ull) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_spin(unsigned long long ticks, unsigned long long *sink) {  // ALU only, until `ticks` of the 100 MHz counter have passed
    const unsigned long long t0 = wall_clock64();
    float x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        for (int i = 0; i < 256; ++i) x = x * 1.0001f + 0.5f;
    }
    if (x == 12345.f) *sink = 1;
}
__global__ void k_dirty(uint32_t *buf, uint32_t n, uint32_t v) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = v + i;
}
__global__ void k_stream(const uint4 *src, size_t n, int reps, uint32_t *sink) {
    uint32_t acc = 0;
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const uint4 v = src[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    if (acc == 0x12345678u) *sink = acc;
}
// V1: lane 0 .. 63 each chase their own cycle; out[0] = ticks of the whole chase (wall clock, 10 ns), out[1] = hops
__global__ void k_chase(const uint32_t *tab, uint32_t start_stride, int hops, unsigned long long *out) {
    uint32_t p = threadIdx.x * start_stride;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < hops; ++i) p = tab[p];
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = (unsigned long long)hops;
    }
    if (p == 0xFFFFFFFFu) out[2] = p;
}
// V2: one link of a dependent chain: dst[i] = src[perm-ish index] + 1 (reads what the previous launch wrote)
__global__ void k_link(const uint32_t *src, uint32_t *dst, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[(i * 2654435761u) % n] + 1u;
}

struct Ctx {
    hipStream_t sv, sa, sa2;
    uint32_t *tab_small, *tab_big, *link_a, *link_b, *dirty;
    uint4 *big;
    size_t big_n;
    unsigned long long *out, *sink;
    int *gate;  // pinned host memory
};

static void make_cycle(std::vector<uint32_t> &t, uint32_t n, uint32_t stride_words) {  // one random cycle over n slots spaced stride_words apart
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::mt19937 rng(12345);
    std::shuffle(order.begin() + 1, order.end(), rng);
    t.assign((size_t)n * stride_words, 0u);
    for (uint32_t i = 0; i < n; ++i) t[(size_t)order[i] * stride_words] = order[(i + 1) % n] * stride_words;
}

enum Agg { A_NONE, A_EMPTY1, A_EMPTY256, A_SPIN_ALL, A_SPIN8, A_DIRTY, A_STREAM, A_EMPTY_2S, A_N };
static const char *agg_name[] = {"none", "empty launches (1 wg)", "empty launches (256 wg)", "one ALU kernel, 256 wg", "one ALU kernel, 8 wg",
                                 "launches dirtying 256 KB", "streaming read (HBM)", "empty launches, 2 streams"};

static void start_aggressor(Ctx &c, Agg a, double ms) {
    const unsigned long long ticks = (unsigned long long)(ms * 1e5);  // 100 MHz
    const int nlaunch = std::min(1500, (int)(ms * 1000.0 / 2.5));     // ~2.5 us per trivial launch; bounded: the queue must not fill behind the gate
    if (a != A_NONE) hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, c.sa, (volatile int *)c.gate);
    if (a == A_EMPTY_2S) hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, c.sa2, (volatile int *)c.gate);
    switch (a) {
    case A_NONE: break;
    case A_EMPTY1: for (int i = 0; i < nlaunch; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, c.sa); break;
    case A_EMPTY256: for (int i = 0; i < nlaunch; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, c.sa); break;
    case A_SPIN_ALL: hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, c.sa, ticks, c.sink); break;
    case A_SPIN8: hipLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, c.sa, ticks, c.sink); break;
    case A_DIRTY: for (int i = 0; i < nlaunch; ++i) hipLaunchKernelGGL(k_dirty, dim3(64), dim3(256), 0, c.sa, c.dirty, 65536u, (uint32_t)i); break;
    case A_STREAM: hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, c.sa, (const uint4 *)c.big, c.big_n, (int)(ms / 0.25) + 1, (uint32_t *)c.sink); break;
    case A_EMPTY_2S:
        for (int i = 0; i < nlaunch; ++i) {
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, c.sa);
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, c.sa2);
        }
        break;
    default: break;
    }
}

int main() {
    Ctx c;
    CK(hipStreamCreateWithFlags(&c.sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c.sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c.sa2, hipStreamNonBlocking));
    // chase tables: 64-byte stride so that every hop is a new cache line.  small: 16 k lines = 1 MB (fits one XCD's L2); big: 1 M lines = 64 MB (Infinity Cache)
    std::vector<uint32_t> h;
    const uint32_t n_small = 16384, n_big = 1u << 20;
    make_cycle(h, n_small, 16);
    CK(hipMalloc(&c.tab_small, h.size() * 4));
    CK(hipMemcpy(c.tab_small, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    make_cycle(h, n_big, 16);
    CK(hipMalloc(&c.tab_big, h.size() * 4));
    CK(hipMemcpy(c.tab_big, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const uint32_t link_n = 262144;  // 1 MB per link array
    CK(hipMalloc(&c.link_a, link_n * 4));
    CK(hipMalloc(&c.link_b, link_n * 4));
    CK(hipMemset(c.link_a, 0, link_n * 4));
    CK(hipMalloc(&c.dirty, 65536 * 4));
    c.big_n = (size_t)(1u << 30) / 16;
    CK(hipMalloc(&c.big, c.big_n * 16));
    CK(hipMemset(c.big, 1, c.big_n * 16));
    CK(hipMalloc(&c.out, 64));
    CK(hipMalloc(&c.sink, 64));
    CK(hipHostMalloc((void **)&c.gate, 64, hipHostMallocMapped));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    const int hops = 2000, links = 40;
    printf("%-28s | chase 1 MB table (ns/hop) | chase 64 MB table (ns/hop) | chain of %d dependent launches (us/launch)\n", "other stream", links);
    for (int a = 0; a < A_N; ++a) {
        double res[3] = {0, 0, 0};
        for (int v = 0; v < 3; ++v) {
            double best = 1e30, sum = 0;
            const int reps = 3;
            for (int r = 0; r < reps; ++r) {
                CK(hipDeviceSynchronize());
                *(volatile int *)c.gate = 0;
                start_aggressor(c, (Agg)a, 4.0);
                hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, c.sv, (volatile int *)c.gate);
                double val = 0;
                if (v < 2) {
                    // warm the table into the cache level it would live in (one untimed pass), then the timed chase
                    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, c.sv, (const uint32_t *)(v == 0 ? c.tab_small : c.tab_big), 16u * 97u, hops, c.out);
                    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, c.sv, (const uint32_t *)(v == 0 ? c.tab_small : c.tab_big), 16u * 97u, hops, c.out);
                    *(volatile int *)c.gate = 1;
                    CK(hipStreamSynchronize(c.sv));
                    unsigned long long o[2];
                    CK(hipMemcpy(o, c.out, 16, hipMemcpyDeviceToHost));
                    val = (double)o[0] * 10.0 / (double)o[1];
                } else {
                    CK(hipEventRecord(e0, c.sv));
                    for (int l = 0; l < links; ++l)
                        hipLaunchKernelGGL(k_link, dim3(link_n / 256), dim3(256), 0, c.sv, (const uint32_t *)((l & 1) ? c.link_b : c.link_a),
                                           (l & 1) ? c.link_a : c.link_b, link_n);
                    CK(hipEventRecord(e1, c.sv));
                    *(volatile int *)c.gate = 1;
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    val = ms * 1000.0 / links;
                }
                best = std::min(best, val);
                sum += val;
                CK(hipDeviceSynchronize());
            }
            res[v] = sum / reps;
            (void)best;
        }
        printf("%-28s | %25.1f | %26.1f | %10.2f\n", agg_name[a], res[0], res[1], res[2]);
    }
    return 0;
}
