// cycles of the wavefront-level sort primitives on LDS-resident data (one wavefront, nothing else on the CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../erasor_amd/csrc/exact_sort.hip.h"

__global__ __launch_bounds__(1024) void k_probe(const uint32_t *keys, uint32_t n, int op, int nwaves_active, unsigned long long *out) {
    __shared__ uint32_t sK[16][2048 / 8], sV[16][2048 / 8], sL[16][2048 / 8], sR[16][2048 / 8];  // per-wave regions when nwaves_active > 1 (n <= 256)
    __shared__ uint32_t bK[2048], bV[2048], bL[2048], bR[2048];
    __shared__ uint32_t sH[2048 / 32 + 2];
    __shared__ uint32_t wsc[16][128];
    __shared__ uint32_t nfb;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t *K = nwaves_active > 1 ? sK[wave] : bK, *V = nwaves_active > 1 ? sV[wave] : bV;
    uint32_t *L = nwaves_active > 1 ? sL[wave] : bL, *R = nwaves_active > 1 ? sR[wave] : bR;
    if ((int)wave < nwaves_active)
        for (uint32_t i = lane; i < n; i += 64) { K[i] = keys[i] + wave; V[i] = i; }
    for (uint32_t i = threadIdx.x; i < 2048 / 32 + 2; i += blockDim.x) sH[i] = 0;
    __syncthreads();
    if ((int)wave >= nwaves_active) return;
    const unsigned long long t0 = clock64();
    uint32_t cut = 0;
    if (op == 0) cut = esort::wave_partition(K, V, L, R, 0u, n);
    else if (op == 1) esort::wave_small_subtree(K, V, sH, 0u, n, 20, &nfb, wsc[wave]);
    else if (op == 2) { if (lane == 0) esort::move_median_to_first(K, V, 0u, n); esort::wave_sync(); }
    const unsigned long long t1 = clock64();
    if (lane == 0 && wave == 0) { out[0] = t1 - t0; out[1] = cut; }
}

int main() {
    uint32_t *dk; unsigned long long *dout, h[2];
    std::vector<uint32_t> k(2048);
    srand(1);
    for (auto &x : k) x = rand() % 800;
    hipMalloc(&dk, 2048 * 4); hipMalloc(&dout, 16);
    hipMemcpy(dk, k.data(), 2048 * 4, hipMemcpyHostToDevice);
    const char *names[3] = {"wave_partition", "wave_small_subtree", "move_median_to_first"};
    for (int op = 0; op < 3; ++op)
        for (int nw : {1, 16})
            for (uint32_t n : {24u, 48u, 64u, 128u, 256u, 512u, 1024u, 2048u}) {
                if (op == 1 && n > 64) continue;
                if (nw > 1 && n > 256) continue;
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(k_probe, dim3(1), dim3(1024), 0, 0, dk, n, op, nw, dout);
                    hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
                }
                printf("%-22s n=%4u waves=%2d : %6llu cycles (cut %llu)\n", names[op], n, nw, h[0], h[1]);
            }
    return 0;
}
