// Pins erasor_amd/csrc/exact_sort_core.h (the parallel formulation the HIP kernels execute)
// against the real libstdc++ std::sort on the host.  Built and run by tests/test_exact_sort.py.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../erasor_amd/csrc/exact_sort_core.h"

struct KV {
    uint32_t k, v;
    bool operator<(const KV &o) const { return k < o.k; }
};

struct Seg {
    uint32_t first, last;
    int depth;
};

// level-synchronous emulation: every partition computed from stop lists (as the wavefronts do)
static void emu_sort(std::vector<uint32_t> &K, std::vector<uint32_t> &V, int *n_heap) {
    const uint32_t n = (uint32_t)K.size();
    if (n == 0) return;
    std::vector<uint8_t> head(n + 1, 0);
    head[0] = 1;
    std::vector<Seg> q, nq;
    if (n > esort::kThreshold) q.push_back({0, n, 2 * esort::lg2_floor(n)});
    std::vector<uint32_t> posL(n), posR(n);
    while (!q.empty()) {
        nq.clear();
        for (const Seg &s : q) {
            if (s.depth == 0) {
                esort::heapsort_exact(K.data(), V.data(), s.first, s.last);
                if (n_heap) ++*n_heap;
                continue;
            }
            esort::move_median_to_first(K.data(), V.data(), s.first, s.last);
            const uint32_t p = K[s.first], lo = s.first + 1, hi = s.last;
            uint32_t nL = 0, nR = 0;
            for (uint32_t i = lo; i < hi; ++i) {
                if (!(K[i] < p)) posL[lo + nL++] = i;
                if (!(p < K[i])) posR[lo + nR++] = i;
            }
            uint32_t m = 0;
            const uint32_t lim = nL < nR ? nL : nR;
            while (m < lim && posL[lo + m] < posR[lo + nR - 1 - m]) ++m;
            const uint32_t cut = esort::cut_from_lists(posL.data() + lo, posR.data() + lo, nL, nR, m);
            for (uint32_t k = 0; k < m; ++k) esort::swap_kv(K.data(), V.data(), posL[lo + k], posR[lo + nR - 1 - k]);
            head[cut] = 1;
            if (cut - s.first > esort::kThreshold) nq.push_back({s.first, cut, s.depth - 1});
            if (s.last - cut > esort::kThreshold) nq.push_back({cut, s.last, s.depth - 1});
        }
        q.swap(nq);
    }
    // leaves: stable sort inside each
    std::vector<uint32_t> K2(n), V2(n);
    uint32_t a = 0;
    while (a < n) {
        uint32_t b = a + 1;
        while (b < n && !head[b]) ++b;
        for (uint32_t i = a; i < b; ++i) {
            const uint32_t r = esort::leaf_rank(K.data(), a, b, i);
            K2[a + r] = K[i];
            V2[a + r] = V[i];
        }
        a = b;
    }
    K.swap(K2);
    V.swap(V2);
}

static int check(const std::vector<uint32_t> &keys, const char *what, int *n_heap) {
    const size_t n = keys.size();
    std::vector<KV> ref(n);
    for (size_t i = 0; i < n; ++i) ref[i] = {keys[i], (uint32_t)i};
    std::sort(ref.begin(), ref.end());
    std::vector<uint32_t> K = keys, V(n);
    for (size_t i = 0; i < n; ++i) V[i] = (uint32_t)i;
    emu_sort(K, V, n_heap);
    for (size_t i = 0; i < n; ++i)
        if (K[i] != ref[i].k || V[i] != ref[i].v) {
            fprintf(stderr, "MISMATCH %s n=%zu at %zu: got (%u,%u) want (%u,%u)\n", what, n, i, K[i], V[i], ref[i].k, ref[i].v);
            return 1;
        }
    return 0;
}

// median-of-3 killer (Musser) to force the heapsort fallback
static std::vector<uint32_t> killer(uint32_t n) {
    std::vector<uint32_t> a(n);
    const uint32_t k = n / 2;
    for (uint32_t i = 1; i <= k; ++i) {
        if (i & 1) {
            a[i - 1] = i;
            a[i] = k + i;
        }
        a[k + i - 1] = 2 * i;
    }
    return a;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(12345);
    int bad = 0, n_heap = 0;
    const uint32_t sizes[] = {0, 1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 64, 100, 257, 1000, 4097, 20000};
    for (uint32_t n : sizes) {
        for (uint32_t range : {1u, 2u, 3u, 7u, 50u, 1000u, 0xFFFFFFFFu}) {
            for (int rep = 0; rep < 4; ++rep) {
                std::vector<uint32_t> k(n);
                for (auto &x : k) x = range == 0xFFFFFFFFu ? rng() : rng() % range;
                bad += check(k, "random", &n_heap);
            }
        }
        std::vector<uint32_t> s(n);
        for (uint32_t i = 0; i < n; ++i) s[i] = i;
        bad += check(s, "sorted", &n_heap);
        for (uint32_t i = 0; i < n; ++i) s[i] = n - i;
        bad += check(s, "reverse", &n_heap);
        for (uint32_t i = 0; i < n; ++i) s[i] = i < n / 2 ? i : n - i;
        bad += check(s, "organ", &n_heap);
        for (uint32_t i = 0; i < n; ++i) s[i] = i / 5;
        bad += check(s, "runs", &n_heap);
        for (uint32_t i = 0; i < n; ++i) s[i] = (i * 7919u) % 13u;
        bad += check(s, "mod13", &n_heap);
    }
    for (int r = 0; r < rounds; ++r) {
        const uint32_t n = 17 + rng() % 3000;
        const uint32_t range = 1 + rng() % (1 + (rng() % 2 ? 10 : 5000));
        std::vector<uint32_t> k(n);
        for (auto &x : k) x = rng() % range;
        bad += check(k, "fuzz", &n_heap);
    }
    for (uint32_t n : {64u, 1000u, 4096u, 30000u}) bad += check(killer(n), "killer", &n_heap);
    printf("esort_check: %s (heapsort fallbacks exercised: %d)\n", bad ? "FAIL" : "OK", n_heap);
    return bad ? 1 : 0;
}
