// GPU micro-benchmark: one workgroup sorts n keys exactly like std::sort -- block_esort (one wavefront per segment) against
// block_esort_sync (level-synchronous) -- same input, outputs compared, shader cycles per call.  Build: see tools/gpu_r03b.sh
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../erasor_amd/csrc/exact_sort.hip.h"

static constexpr uint32_t LMAX = 4096;

__global__ __launch_bounds__(1024) void k_old(const uint32_t *keys, uint32_t n, uint32_t *outK, uint32_t *outV, unsigned long long *cyc, uint32_t *ctr) {
    __shared__ uint32_t pool[4 * LMAX];
    __shared__ uint32_t sH[LMAX / 32 + 2];
    __shared__ esort::Seg qa[LMAX / 16 + 2], qb[LMAX / 16 + 2];
    __shared__ uint32_t qcnt[2];
    uint32_t *sK = pool, *sV = pool + LMAX, *sL = pool + 2 * LMAX, *sR = pool + 3 * LMAX;
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    for (uint32_t i = tid; i < n; i += bs) {
        sK[i] = keys[i];
        sV[i] = i;
    }
    __syncthreads();
    const unsigned long long t0 = clock64();
    esort::block_esort(sK, sV, sL, sR, sH, sL, sR, 0u, n, 2 * esort::lg2_floor(n), qa, qb, qcnt, (uint32_t)(LMAX / 16 + 2), ctr, ctr + 1);
    const unsigned long long t1 = clock64();
    for (uint32_t i = tid; i < n; i += bs) {
        outK[i] = sL[i];
        outV[i] = sR[i];
    }
    if (tid == 0) cyc[0] = t1 - t0;
}

template <int EM>
__global__ __launch_bounds__(1024) void k_new(const uint32_t *keys, uint32_t n, uint32_t *outK, uint32_t *outV, unsigned long long *cyc, uint32_t *ctr) {
    __shared__ uint2 sKV[LMAX];
    __shared__ uint2 sLL[LMAX], sRR[LMAX];
    __shared__ uint32_t sPS[LMAX], sCut[LMAX], sTab[68];
    __shared__ unsigned long long ts[20];
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    uint32_t k[EM], v[EM];
#pragma unroll
    for (int e = 0; e < EM; ++e) {
        const uint32_t i = (uint32_t)e * bs + tid;
        k[e] = i < n ? keys[i] : 0u;
        v[e] = i;
    }
    if (tid < 20) ts[tid] = 0;
    __syncthreads();
    const unsigned long long t0 = clock64();
    esort::block_esort_sync<EM>(k, v, n, sKV, sLL, sRR, sPS, sCut, sTab, (uint32_t *)sLL, (uint32_t *)sRR, ctr, ts);
    const unsigned long long t1 = clock64();
    for (uint32_t i = tid; i < n; i += bs) {
        outK[i] = ((uint32_t *)sLL)[i];
        outV[i] = ((uint32_t *)sRR)[i];
    }
    if (tid == 0) {
        cyc[0] = t1 - t0;
        for (int i = 0; i < 17; ++i) cyc[1 + i] = ts[i];
    }
}

int main() {
    std::mt19937 rng(7);
    uint32_t *dk, *oK, *oV, *oK2, *oV2, *ctr;
    unsigned long long *cyc;
    hipMalloc(&dk, LMAX * 4);
    hipMalloc(&oK, LMAX * 4);
    hipMalloc(&oV, LMAX * 4);
    hipMalloc(&oK2, LMAX * 4);
    hipMalloc(&oV2, LMAX * 4);
    hipMalloc(&ctr, 64);
    hipMalloc(&cyc, 64 * 8);
    hipMemset(ctr, 0, 64);
    int bad = 0;
    for (uint32_t n : {100u, 300u, 600u, 1000u, 1024u, 1500u, 2048u, 4096u}) {
        for (int kind = 0; kind < 2; ++kind) {  // 0: z-like float keys with a few ties, 1: voxel-index-like keys (many ties)
            std::vector<uint32_t> keys(n);
            for (auto &x : keys) {
                if (kind == 0) {
                    const float z = -1.7f + 0.0005f * (float)(rng() % 6000);
                    uint32_t b;
                    memcpy(&b, &z, 4);
                    x = esort::float_key(b);
                } else
                    x = rng() % (n / 4 + 2);
            }
            std::vector<std::pair<uint32_t, uint32_t>> ref(n);
            for (uint32_t i = 0; i < n; ++i) ref[i] = {keys[i], i};
            std::sort(ref.begin(), ref.end(), [](auto &a, auto &b) { return a.first < b.first; });
            hipMemcpy(dk, keys.data(), n * 4, hipMemcpyHostToDevice);
            unsigned long long c_old = ~0ull, c_new = ~0ull, lv[18] = {0};
            for (int rep = 0; rep < 5; ++rep) {
                unsigned long long c[32];
                hipLaunchKernelGGL(k_old, dim3(1), dim3(1024), 0, 0, dk, n, oK, oV, cyc, ctr);
                hipMemcpy(c, cyc, 8, hipMemcpyDeviceToHost);
                c_old = std::min(c_old, c[0]);
                if (n <= 1024) hipLaunchKernelGGL(k_new<1>, dim3(1), dim3(1024), 0, 0, dk, n, oK2, oV2, cyc, ctr);
                else if (n <= 2048) hipLaunchKernelGGL(k_new<2>, dim3(1), dim3(1024), 0, 0, dk, n, oK2, oV2, cyc, ctr);
                else hipLaunchKernelGGL(k_new<4>, dim3(1), dim3(1024), 0, 0, dk, n, oK2, oV2, cyc, ctr);
                hipMemcpy(c, cyc, 18 * 8, hipMemcpyDeviceToHost);
                if (c[0] < c_new) {
                    c_new = c[0];
                    memcpy(lv, c, sizeof(lv));
                }
            }
            std::vector<uint32_t> a(n), b(n), a2(n), b2(n);
            hipMemcpy(a.data(), oK, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), oV, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(a2.data(), oK2, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b2.data(), oV2, n * 4, hipMemcpyDeviceToHost);
            bool ok_old = true, ok_new = true;
            for (uint32_t i = 0; i < n; ++i) {
                ok_old = ok_old && a[i] == ref[i].first && b[i] == ref[i].second;
                ok_new = ok_new && a2[i] == ref[i].first && b2[i] == ref[i].second;
            }
            int levels = 0;
            for (int i = 0; i < 14; ++i) levels += lv[2 + i] != 0;
            printf("n=%4u kind=%d: block_esort %7llu cycles (%s), block_esort_sync %7llu cycles (%s), %d levels, first levels:", n, kind, c_old,
                   ok_old ? "== std::sort" : "MISMATCH", c_new, ok_new ? "== std::sort" : "MISMATCH", levels);
            for (int i = 0; i < 4 && lv[3 + i]; ++i) printf(" %llu", lv[3 + i] - lv[2 + i]);
            printf("; leaf ranking %llu\n", lv[17] - lv[16]);
            bad += !ok_old || !ok_new;
        }
    }
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
