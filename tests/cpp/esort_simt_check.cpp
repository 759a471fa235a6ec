// Runs the DEVICE sort code (erasor_amd/csrc/exact_sort.hip.h: wave_partition, wave_small_subtree, block_partition,
// block_esort) unmodified on the CPU -- one thread per lane, tests/cpp/simt_emu -- and pins its output to the real
// libstdc++ std::sort: the permutation of equal keys included.  Built and run by tests/test_exact_sort.py.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../erasor_amd/csrc/exact_sort.hip.h"

struct KV {
    uint32_t k, v;
    bool operator<(const KV &o) const { return k < o.k; }
};

static int g_fail = 0;

// one workgroup of `nthreads` sorts K/V[0..n) like k_rgpf2 / k_binvox2 / k_esort_final do (K2/V2 alias posL/posR)
static void check(const char *what, std::vector<uint32_t> keys, unsigned nthreads, uint32_t qcap_override = 0) {
    const uint32_t n = (uint32_t)keys.size();
    std::vector<KV> ref(n);
    for (uint32_t i = 0; i < n; ++i) ref[i] = {keys[i], i};
    std::sort(ref.begin(), ref.end());
    std::vector<uint32_t> K(keys), V(n), posL(n + 64), posR(n + 64), head(n / 32 + 4, 0);
    for (uint32_t i = 0; i < n; ++i) V[i] = i;
    const uint32_t qcap = qcap_override ? qcap_override : n / 16 + 2;
    std::vector<esort::Seg> qa(qcap + 1), qb(qcap + 1);
    uint32_t qcnt[2] = {0, 0}, n_fallback = 0, overflow = 0;
    simt::run_block(nthreads, [&] {
        esort::block_esort(K.data(), V.data(), posL.data(), posR.data(), head.data(), posL.data(), posR.data(), 0u, n,
                           2 * esort::lg2_floor(n ? n : 1), qa.data(), qb.data(), qcnt, qcap, &n_fallback, &overflow);
    });
    bool ok = overflow == 0;
    for (uint32_t i = 0; ok && i < n; ++i) ok = posL[i] == ref[i].k && posR[i] == ref[i].v;
    printf("%-44s n=%5u threads=%4u heapsorts=%u  %s\n", what, n, nthreads, n_fallback, ok ? "ok" : "MISMATCH");
    if (!ok) ++g_fail;
}

// the level-synchronous variant (block_esort_sync): keys enter in registers, position i = e * blockDim.x + tid
static void check_sync(const char *what, std::vector<uint32_t> keys, unsigned nthreads) {
    const uint32_t n = (uint32_t)keys.size();
    std::vector<KV> ref(n);
    for (uint32_t i = 0; i < n; ++i) ref[i] = {keys[i], i};
    std::sort(ref.begin(), ref.end());
    std::vector<uint2> kv(n + 64);
    std::vector<uint2> LL(n + 64, uint2{0xDEADBEEFu, 0xDEADBEEFu}), RR(n + 64, uint2{0xDEADBEEFu, 0xDEADBEEFu});
    std::vector<uint32_t> PS(n + 64), Cut(n + 64), Tab(68, 0xA5A5A5A5u);
    uint32_t n_fallback = 0;
    simt::run_block(nthreads, [&] {
        uint32_t k[4], v[4];
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = (uint32_t)e * blockDim.x + threadIdx.x;
            k[e] = i < n ? keys[i] : 0u;
            v[e] = i;
        }
        esort::block_esort_sync<4>(k, v, n, kv.data(), LL.data(), RR.data(), PS.data(), Cut.data(), Tab.data(), (uint32_t *)LL.data(), (uint32_t *)RR.data(), &n_fallback);
    });
    bool ok = true;
    for (uint32_t i = 0; ok && i < n; ++i) ok = ((uint32_t *)LL.data())[i] == ref[i].k && ((uint32_t *)RR.data())[i] == ref[i].v;
    printf("sync: %-38s n=%5u threads=%4u heapsorts=%u  %s\n", what, n, nthreads, n_fallback, ok ? "ok" : "MISMATCH");
    if (!ok) ++g_fail;
}

int main(int argc, char **argv) {
    const int quick = argc > 1 ? atoi(argv[1]) : 0;
    std::mt19937 rng(20210310);
    auto uni = [&](uint32_t n, uint32_t range) {
        std::vector<uint32_t> k(n);
        for (auto &x : k) x = rng() % range;
        return k;
    };
    // sizes around the thresholds: 16 (leaf), 64 (register subtree), 2048 (whole-workgroup partitions), with few / many ties
    const uint32_t sizes[] = {1, 16, 17, 40, 64, 65, 130, 600, 1000, 2048, 2049, 3000, 4096};
    for (uint32_t n : sizes) {
        if (quick && n > 1100) continue;
        check("uniform keys, few ties", uni(n, 1u << 30), 1024);
        check("uniform keys, many ties", uni(n, n / 4 + 2), 1024);
        check_sync("uniform keys, few ties", uni(n, 1u << 30), 1024);
        check_sync("uniform keys, many ties", uni(n, n / 4 + 2), 1024);
        check_sync("uniform keys, very many ties", uni(n, 7), 1024);
        if (n <= 1024) check_sync("uniform keys, many ties (4 wavefronts)", uni(n, n / 4 + 2), 256);
    }
    check("all keys equal", std::vector<uint32_t>(700, 7u), 1024);
    check_sync("all keys equal", std::vector<uint32_t>(700, 7u), 1024);
    check_sync("all keys equal", std::vector<uint32_t>(3000, 7u), 1024);
    {
        std::vector<uint32_t> k(900);
        for (uint32_t i = 0; i < k.size(); ++i) k[i] = i;
        check("ascending", k, 1024);
        check_sync("ascending", k, 1024);
        std::reverse(k.begin(), k.end());
        check("descending", k, 1024);
        check_sync("descending", k, 1024);
    }
    for (uint32_t n : {1000u, 4096u}) {  // median-of-three killer (Musser): drives introsort into its heapsort fallback
        if (quick && n > 1100) continue;
        std::vector<uint32_t> k(n, 0u);
        const uint32_t h = n / 2;
        for (uint32_t i = 1; i <= h; ++i) {
            if (i & 1u) {
                k[i - 1] = i;
                k[i] = h + i;
            }
            k[h + i - 1] = 2 * i;
        }
        const int before = g_fail;
        check("median-of-3 adversary (heapsort fallback)", k, 1024);
        check_sync("median-of-3 adversary (heapsort fallback)", k, 1024);
        (void)before;
    }
    check("a workgroup of four wavefronts", uni(1500, 300), 256);
    check("z-like keys (float bits), ties", [&] {
        std::vector<uint32_t> k(1200);
        for (auto &x : k) {
            const float z = -1.5f + 0.01f * (float)(rng() % 300);
            uint32_t b;
            memcpy(&b, &z, 4);
            x = esort::float_key(b);
        }
        return k;
    }(), 1024);
    for (int rep = 0; rep < (quick ? 4 : 24); ++rep) {  // random sizes, tie densities
        const uint32_t n = 1 + rng() % 4096, range = 1 + rng() % (rep & 1 ? 64 : 100000);
        check_sync("random size / tie density", uni(n, range), 1024);
    }
    printf("%s\n", g_fail ? "FAILED" : "ALL OK");
    return g_fail ? 1 : 0;
}
