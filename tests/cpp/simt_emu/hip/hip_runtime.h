// A host stand-in for <hip/hip_runtime.h>, just large enough to run erasor_amd/csrc/exact_sort.hip.h on the CPU:
// one OS thread per lane, 64 lanes per wavefront, the cross-lane intrinsics as wavefront barriers over a shared slot
// array.  TEST INFRASTRUCTURE (tests/cpp/esort_simt_check.cpp): the device code is compiled unmodified.
//
// Stricter than the hardware in one respect, on purpose: lanes do not run in lockstep here, so a wavefront that relies on
// another lane's LDS write without the wave_sync() (__threadfence_block) the code is supposed to issue reads stale data.
#ifndef ERASOR_SIMT_EMU_HIP_RUNTIME_H
#define ERASOR_SIMT_EMU_HIP_RUNTIME_H
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace simt {
struct Wave {
    pthread_barrier_t bar;
    uint64_t slot[2][64];
};
struct Dim {
    unsigned x = 0, y = 0, z = 0;
};
struct Ctx {
    Wave *wave = nullptr;
    pthread_barrier_t *block = nullptr;
    unsigned gen = 0;  // collectives issued by this lane (parity selects the slot array)
};
inline thread_local Ctx ctx;
inline void wave_barrier() { pthread_barrier_wait(&ctx.wave->bar); }
// every lane publishes a value, then reads whatever it needs of the others' (two slot arrays: a lane can be at most one
// collective ahead of the slowest one)
template <class F>
inline auto exchange(uint64_t mine, unsigned lane, F &&read) {
    uint64_t *s = ctx.wave->slot[ctx.gen++ & 1u];
    __atomic_store_n(&s[lane], mine, __ATOMIC_RELEASE);
    wave_barrier();
    return read((const uint64_t *)s);
}
}  // namespace simt

inline thread_local simt::Dim threadIdx;
inline simt::Dim blockDim, blockIdx, gridDim;

inline unsigned simt_lane() { return threadIdx.x & 63u; }

inline void __syncthreads() {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    pthread_barrier_wait(simt::ctx.block);
}
// intra-wavefront visibility of earlier writes: on the device the lanes run in lockstep and the fence orders memory; here
// the lanes have to meet as well
inline void __threadfence_block() {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    simt::wave_barrier();
}
inline void __threadfence() { __threadfence_block(); }

inline uint64_t __ballot(int pred) {
    return simt::exchange(pred ? 1u : 0u, simt_lane(), [](const uint64_t *s) {
        uint64_t m = 0;
        for (int i = 0; i < 64; ++i) m |= (uint64_t)(__atomic_load_n(&s[i], __ATOMIC_ACQUIRE) & 1u) << i;
        return m;
    });
}
template <class T>
inline T simt_shfl_from(T v, int src) {
    static_assert(sizeof(T) <= 8, "64-bit payloads at most");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const unsigned lane = simt_lane();
    const int from = (src < 0 || src > 63) ? (int)lane : src;
    const uint64_t r = simt::exchange(bits, lane, [from](const uint64_t *s) { return __atomic_load_n(&s[from], __ATOMIC_ACQUIRE); });
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <class T>
inline T __shfl(T v, int src, int = 64) { return simt_shfl_from(v, src & 63); }
template <class T>
inline T __shfl_up(T v, unsigned d, int = 64) { return simt_shfl_from(v, (int)simt_lane() - (int)d); }
template <class T>
inline T __shfl_down(T v, unsigned d, int = 64) { return simt_shfl_from(v, (int)simt_lane() + (int)d); }
template <class T>
inline T __shfl_xor(T v, int m, int = 64) { return simt_shfl_from(v, (int)(simt_lane() ^ (unsigned)m)); }
inline uint32_t __builtin_amdgcn_readlane(uint32_t v, uint32_t lane) { return simt_shfl_from(v, (int)(lane & 63u)); }
inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t v) { return simt_shfl_from(v, 0); }

inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline unsigned long long clock64() { return 0; }
inline unsigned long long wall_clock64() { return 0; }

inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t atomicAnd(uint32_t *p, uint32_t v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
using std::max;
using std::min;

namespace simt {
// run `body` as ONE workgroup of nthreads (a multiple of 64) threads
inline void run_block(unsigned nthreads, const std::function<void()> &body) {
    const unsigned nw = nthreads / 64;
    std::vector<Wave> waves(nw);
    for (auto &w : waves) pthread_barrier_init(&w.bar, nullptr, 64);
    pthread_barrier_t block;
    pthread_barrier_init(&block, nullptr, nthreads);
    blockDim.x = nthreads;
    blockDim.y = blockDim.z = 1;
    gridDim.x = gridDim.y = gridDim.z = 1;
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            threadIdx.x = t;
            ctx.wave = &waves[t / 64];
            ctx.block = &block;
            ctx.gen = 0;
            body();
        });
    for (auto &t : th) t.join();
    for (auto &w : waves) pthread_barrier_destroy(&w.bar);
    pthread_barrier_destroy(&block);
}
}  // namespace simt
#endif
