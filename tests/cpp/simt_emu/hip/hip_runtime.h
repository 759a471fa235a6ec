// A host stand-in for <hip/hip_runtime.h>, just large enough to run erasor_amd/csrc/exact_sort.hip.h on the CPU:
// one OS thread per lane, 64 lanes per wavefront, the cross-lane intrinsics as wavefront barriers over a shared slot
// array.  TEST INFRASTRUCTURE (tests/cpp/esort_simt_check.cpp): the device code is compiled unmodified.
//
// Stricter than the hardware in one respect, on purpose: lanes do not run in lockstep here, so a wavefront that relies on
// another lane's LDS write without the wave_sync() (__threadfence_block) the code is supposed to issue reads stale data.
#ifndef ERASOR_SIMT_EMU_HIP_RUNTIME_H
#define ERASOR_SIMT_EMU_HIP_RUNTIME_H
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace simt {
// a barrier that threads can leave for good (a lane that returns from the kernel no longer takes part, like a finished
// wavefront on the device)
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    unsigned need = 0, arrived = 0, gen = 0;
    void wait() {
        std::unique_lock<std::mutex> l(m);
        if (++arrived >= need) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else {
            const unsigned g = gen;
            cv.wait(l, [&] { return gen != g; });
        }
    }
    void leave() {
        std::unique_lock<std::mutex> l(m);
        --need;
        if (need > 0 && arrived >= need) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        }
    }
};
// Cross-lane exchanges (ballot, shuffles, readlane) do NOT use a barrier: the lanes of a wavefront may have diverged into
// groups that exchange among themselves (eight lanes per voxel in the label searches).  Every lane logs what it publishes,
// tagged (epoch, number of the exchange since the last barrier); a reader waits for the tag it expects from its source lane.
// Lanes that run the same code have the same count, so the tags match inside a converged group; the counts are reset at
// every barrier (where a wavefront is converged again).  A lane that has returned from the kernel reads as zero.
static constexpr unsigned LOGN = 4096;
struct Entry {
    std::atomic<uint64_t> tag{0};
    uint64_t val = 0;
};
struct Wave {
    Barrier bar;
    Entry log[64][LOGN];
    std::atomic<int> gone[64];
};
struct Dim {
    unsigned x = 0, y = 0, z = 0;
};
struct Ctx {
    Wave *wave = nullptr;
    Barrier *block = nullptr;
    uint32_t epoch = 1, seq = 0;
};
inline thread_local Ctx ctx;
inline void wave_barrier() {
    ctx.wave->bar.wait();
    ++ctx.epoch;
    ctx.seq = 0;
}
inline void publish(unsigned lane, uint64_t v, uint64_t tag) {
    Entry &e = ctx.wave->log[lane][ctx.seq % LOGN];
    e.val = v;
    e.tag.store(tag, std::memory_order_release);
}
inline uint64_t fetch(unsigned src, uint64_t tag) {
    Entry &e = ctx.wave->log[src][ctx.seq % LOGN];
    for (unsigned spins = 0;; ++spins) {
        const uint64_t t = e.tag.load(std::memory_order_acquire);
        if (t == tag) return e.val;
        if (t > tag) {
            fprintf(stderr, "simt_emu: lane %u ran %u exchanges ahead of a reader (log too short)\n", src, LOGN);
            abort();
        }
        if (ctx.wave->gone[src].load(std::memory_order_acquire)) return 0;
        if (spins > 64) sched_yield();
    }
}
template <class F>
inline auto exchange(uint64_t mine, unsigned lane, F &&read) {
    const uint64_t tag = ((uint64_t)ctx.epoch << 32) | (ctx.seq + 1u);
    publish(lane, mine, tag);
    auto r = read(tag);
    ++ctx.seq;
    return r;
}
}  // namespace simt

inline thread_local simt::Dim threadIdx;
inline simt::Dim blockDim, blockIdx, gridDim;

inline unsigned simt_lane() { return threadIdx.x & 63u; }

inline void __syncthreads() {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    simt::ctx.block->wait();
    ++simt::ctx.epoch;
    simt::ctx.seq = 0;
}
// intra-wavefront visibility of earlier writes: on the device the lanes run in lockstep and the fence orders memory; here
// the lanes have to meet as well
inline void __threadfence_block() {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    simt::wave_barrier();
}
inline void __threadfence() { __threadfence_block(); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }  // (one lane, towards the host: no meeting)

inline uint64_t __ballot(int pred) {
    return simt::exchange(pred ? 1u : 0u, simt_lane(), [](uint64_t tag) {
        uint64_t m = 0;
        for (unsigned i = 0; i < 64; ++i) m |= (simt::fetch(i, tag) & 1u) << i;
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
    const uint64_t r = simt::exchange(bits, lane, [from](uint64_t tag) { return simt::fetch((unsigned)from, tag); });
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
inline uint32_t atomicCAS(uint32_t *p, uint32_t expected, uint32_t desired) {
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expected;
}
inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t atomicMin(uint32_t *p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) {
    unsigned long long o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
    unsigned long long o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
using std::max;
using std::min;
using std::isfinite;

// ---- vector types and bit casts ----
struct float2 {
    float x, y;
};
struct alignas(16) float4 {
    float x, y, z, w;
};
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline uint32_t __float_as_uint(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
inline int __float_as_int(float f) {
    int u;
    memcpy(&u, &f, 4);
    return u;
}
inline float __uint_as_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline float __int_as_float(int u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

namespace simt {
// run `body` as ONE workgroup of nthreads (a multiple of 64) threads
inline void run_block(unsigned nthreads, const std::function<void()> &body) {
    const unsigned nw = nthreads / 64;
    std::unique_ptr<Wave[]> waves(new Wave[nw]);
    for (unsigned w = 0; w < nw; ++w) {
        waves[w].bar.need = 64;
        for (auto &g : waves[w].gone) g.store(0);
    }
    Barrier block;
    block.need = nthreads;
    blockDim.x = nthreads;
    blockDim.y = blockDim.z = 1;
    if (gridDim.x == 0) gridDim.x = 1;
    gridDim.y = gridDim.z = 1;
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            threadIdx.x = t;
            ctx.wave = &waves[t / 64];
            ctx.block = &block;
            ctx.epoch = 1;
            ctx.seq = 0;
            body();
            // the lane has returned from the kernel: it neither arrives at later barriers nor answers later exchanges
            waves[t / 64].gone[t & 63u].store(1, std::memory_order_release);
            waves[t / 64].bar.leave();
            block.leave();
        });
    for (auto &t : th) t.join();
}
// a grid of workgroups, one after the other (static __shared__ storage is one workgroup's LDS)
inline void run_grid(unsigned nblocks, unsigned nthreads, const std::function<void()> &body) {
    for (unsigned b = 0; b < nblocks; ++b) {
        blockIdx.x = b;
        blockIdx.y = blockIdx.z = 0;
        gridDim.x = nblocks;
        run_block(nthreads, body);
    }
    gridDim.x = 1;
    blockIdx.x = 0;
}
}  // namespace simt
#endif
