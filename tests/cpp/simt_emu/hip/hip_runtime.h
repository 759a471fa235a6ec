// A host stand-in for <hip/hip_runtime.h>, large enough to compile erasor_amd/csrc/{exact_sort.hip.h, kernels.hip.h,
// erasor_hip.hip} UNMODIFIED and run them on the CPU.  TEST INFRASTRUCTURE ONLY (tests/cpp/*_simt_check.cpp,
// tests/test_full_step_on_cpu.py): nothing here is on the product path, and it is ~10^5 times slower than the device.
//
// Execution model
//   * a kernel launch runs its grid synchronously, workgroup after workgroup (static __shared__ storage = the LDS of the one
//     workgroup that is running);
//   * a workgroup = one OS thread per WAVEFRONT; the 64 lanes of a wavefront are fibers (ucontext) that the wavefront's
//     thread runs round-robin.  A lane runs until it has to wait: for another lane's value in a cross-lane exchange
//     (__ballot, __shfl*, readlane), at the intra-wavefront fence, at __syncthreads -- then the next lane runs;
//   * exchanges are tagged (workgroup epoch, number of the exchange since the last __syncthreads): a reader waits for the
//     tag it expects from its source lane, so lane groups that have diverged (the eight lanes of a voxel in the label
//     searches) can exchange among themselves; a lane that has returned from the kernel reads as zero, like an inactive lane;
//   * the intra-wavefront fence (esort::wave_sync = __threadfence_block) is where lanes MEET: on the device the lanes run in
//     lockstep and the fence orders memory; here a lane may be far ahead.  A fence may sit in lane-divergent code
//     (`if (valid) { read; wave_sync(); update; }`), so it opens once every lane of the wavefront has arrived, is waiting at
//     the workgroup barrier or has returned.  This is stricter than the hardware on purpose: code that relies on another
//     lane's write without the fence it is supposed to issue reads stale data here;
//   * the host API is synchronous (streams and events are no-ops): the order of execution is the order of the host's calls,
//     one valid serialisation of the program -- a wait on an event always follows its record in host order.
#ifndef ERASOR_SIMT_EMU_HIP_RUNTIME_H
#define ERASOR_SIMT_EMU_HIP_RUNTIME_H
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define ERASOR_NO_HIPGRAPH 1  // (the stand-in has no graph API: the product code launches kernel by kernel)
#define ERASOR_NO_WORKER_THREAD 1  // (a launch runs synchronously here, on the caller's thread: no second launching thread)
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace simt {
struct Dim {
    unsigned x = 0, y = 0, z = 0;
};
}  // namespace simt
inline thread_local simt::Dim threadIdx;  // (of the lane the wavefront's thread is running right now)
inline simt::Dim blockDim, blockIdx, gridDim;

namespace simt {
static constexpr unsigned LOGN = 1024;        // exchanges a lane may be ahead of its slowest reader
static constexpr size_t STACK = 256 * 1024;   // per lane
enum State : uint8_t { RUN, W_EXCH, W_FENCE, W_BLOCK, DONE };
struct Entry {
    uint64_t tag = 0, val = 0;
};
struct Lane {
    ucontext_t uc;
    char *stack = nullptr;
    State st = DONE;
    unsigned tid = 0;
    uint32_t epoch = 0, seq = 0;
    // what it waits for
    unsigned w_src = 0;
    uint64_t w_tag = 0;
    unsigned w_gen = 0;
};
struct Block;
struct Wave {
    Lane lane[64];
    std::unique_ptr<Entry[]> log{new Entry[64 * LOGN]};
    unsigned present = 0, n_done = 0, fence_arrived = 0, fence_gen = 0, cur = 0;
    unsigned at_block = 0, at_block_gen = 0;  // lanes of this wavefront waiting for workgroup-barrier generation at_block_gen to pass
    ucontext_t sched;
    Block *blk = nullptr;
    Entry &entry(unsigned l, uint32_t seq) { return log[(size_t)l * LOGN + (seq % LOGN)]; }
    inline unsigned waiting_at_block() const;
    bool fence_open() const { return fence_arrived > 0 && fence_arrived + waiting_at_block() + n_done >= present; }
    void open_fence() {
        fence_arrived = 0;
        ++fence_gen;
    }
};
struct Block {
    std::mutex m;  // arrivals / departures of lanes of different wavefront threads
    unsigned need = 0, arrived = 0;
    std::atomic<unsigned> gen{0};  // what the wavefront threads wait on
    const std::function<void()> *body = nullptr;
    uint32_t serial = 0;
    void open() {
        arrived = 0;
        gen.fetch_add(1, std::memory_order_acq_rel);
        gen.notify_all();
    }
    void arrive() {  // a lane has reached the workgroup barrier: the last one opens it
        std::lock_guard<std::mutex> l(m);
        if (++arrived >= need) open();
    }
    void leave() {  // a lane has returned from the kernel for good
        std::lock_guard<std::mutex> l(m);
        --need;
        if (need > 0 && arrived >= need) open();
    }
};
// (once the barrier has opened its waiters run on -- they may still come to a fence -- even if they have not been resumed yet)
inline unsigned Wave::waiting_at_block() const { return blk->gen.load(std::memory_order_acquire) == at_block_gen ? at_block : 0u; }
inline thread_local Wave *tw = nullptr;  // the wavefront this OS thread runs
inline Lane &me() { return tw->lane[tw->cur]; }
inline void yield_to_scheduler() { swapcontext(&me().uc, &tw->sched); }

inline void lane_entry() {
    (*tw->blk->body)();
    Lane &l = me();
    l.st = DONE;
    ++tw->n_done;
    if (tw->fence_open()) tw->open_fence();
    tw->blk->leave();
    yield_to_scheduler();  // never resumed
}
inline void run_wave(Wave &w, Block &b, unsigned first_tid, unsigned present) {
    tw = &w;
    w.blk = &b;
    w.present = present;
    w.n_done = w.fence_arrived = w.at_block = 0;
    w.at_block_gen = b.gen.load(std::memory_order_acquire) - 1u;
    for (unsigned l = 0; l < 64; ++l) {
        Lane &L = w.lane[l];
        L.tid = first_tid + l;
        if (l >= present) {
            L.st = DONE;
            continue;
        }
        if (!L.stack) L.stack = (char *)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        getcontext(&L.uc);
        L.uc.uc_stack.ss_sp = L.stack;
        L.uc.uc_stack.ss_size = STACK;
        L.uc.uc_link = nullptr;
        makecontext(&L.uc, (void (*)())lane_entry, 0);
        L.st = RUN;
        L.epoch = (b.serial << 12) + 1u;
        L.seq = 0;
    }
    unsigned idle_rounds = 0;
    while (w.n_done < present) {
        bool progress = false;
        const unsigned bgen = b.gen.load(std::memory_order_acquire);
        for (unsigned l = 0; l < present; ++l) {
            Lane &L = w.lane[l];
            bool ready = false;
            switch (L.st) {
                case RUN: ready = true; break;
                case W_EXCH: ready = w.entry(L.w_src, L.seq).tag == L.w_tag || w.lane[L.w_src].st == DONE; break;
                case W_FENCE: ready = w.fence_gen != L.w_gen; break;
                case W_BLOCK: ready = bgen != L.w_gen; break;
                case DONE: break;
            }
            if (!ready) continue;
            progress = true;
            w.cur = l;
            threadIdx.x = L.tid;
            L.st = RUN;
            swapcontext(&w.sched, &L.uc);
        }
        if (progress) {
            idle_rounds = 0;
            continue;
        }
        // every live lane waits: for the workgroup barrier (other wavefronts have to arrive) -- or for something that will
        // never come (an exchange between lanes that have diverged for good)
        bool any_block = false;
        for (unsigned l = 0; l < present; ++l) any_block = any_block || w.lane[l].st == W_BLOCK;
        if (any_block) {
            // (sleep in slices: a workgroup barrier that never opens -- lanes of some wavefront stuck elsewhere -- is reported
            // instead of hanging the test)
            const auto t0 = std::chrono::steady_clock::now();
            while (b.gen.load(std::memory_order_acquire) == bgen) {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                    fprintf(stderr, "simt_emu: workgroup %u: the barrier does not open; wavefront of lane %u: fence_arrived %u at_block %u done %u of %u\n",
                            blockIdx.x, w.lane[0].tid, w.fence_arrived, w.at_block, w.n_done, present);
                    for (unsigned l = 0; l < present; ++l)
                        if (w.lane[l].st != W_BLOCK) fprintf(stderr, "   lane %u state %d (waits for lane %u)\n", l, (int)w.lane[l].st, w.lane[l].w_src);
                    abort();
                }
            }
            continue;
        }
        if (++idle_rounds > 1000000) {
            fprintf(stderr, "simt_emu: wavefront of workgroup %u is stuck (an exchange or fence between lanes that never meet)\n", blockIdx.x);
            for (unsigned l = 0; l < present; ++l)
                fprintf(stderr, "   lane %u state %d src %u\n", l, (int)w.lane[l].st, w.lane[l].w_src);
            abort();
        }
    }
}

// ---- what the intrinsics call ----
template <class F>
inline auto exchange(uint64_t mine, F &&read) {
    Lane &L = me();
    const uint64_t tag = ((uint64_t)L.epoch << 32) | (L.seq + 1u);
    Entry &e = tw->entry(tw->cur, L.seq);
    e.val = mine;
    e.tag = tag;
    auto r = read(tag);
    ++me().seq;
    return r;
}
inline uint64_t fetch(unsigned src, uint64_t tag) {
    for (;;) {
        Lane &L = me();
        Entry &e = tw->entry(src, L.seq);
        if (e.tag == tag) return e.val;
        if (e.tag > tag && (e.tag >> 32) == (tag >> 32)) {
            fprintf(stderr, "simt_emu: lane %u ran %u exchanges ahead of a reader (log too short)\n", src, LOGN);
            abort();
        }
        if (tw->lane[src].st == DONE) return 0;
        L.st = W_EXCH;
        L.w_src = src;
        L.w_tag = tag;
        yield_to_scheduler();
    }
}
inline void fence() {
    Wave &w = *tw;
    Lane &L = me();
    ++w.fence_arrived;
    if (w.fence_open()) {
        w.open_fence();
        return;
    }
    L.st = W_FENCE;
    L.w_gen = w.fence_gen;
    yield_to_scheduler();
}
inline void block_barrier() {
    Wave &w = *tw;
    Lane &L = me();
    L.w_gen = w.blk->gen.load(std::memory_order_acquire);
    L.st = W_BLOCK;
    if (w.at_block_gen != L.w_gen) {
        w.at_block_gen = L.w_gen;
        w.at_block = 0;
    }
    ++w.at_block;
    if (w.fence_open()) w.open_fence();
    w.blk->arrive();
    yield_to_scheduler();
    Lane &L2 = me();
    ++L2.epoch;  // (the exchange counters restart where a workgroup is converged for certain)
    L2.seq = 0;
}

// ---- a pool of wavefront threads, kept across workgroups ----
struct Pool {
    static constexpr unsigned MAXW = 16;
    struct Worker {
        std::atomic<unsigned> go{0};
        std::thread th;
    };
    Worker workers[MAXW];
    unsigned nworkers = 0;
    std::unique_ptr<Wave[]> waves{new Wave[MAXW]};
    Block block;
    unsigned nthreads = 0, job = 0, nwaves = 0;
    std::atomic<unsigned> done{0}, done_job{0};
    static constexpr unsigned QUIT = 0xFFFFFFFFu;
    void worker(unsigned wv) {
        unsigned seen = 0;
        for (;;) {
            workers[wv].go.wait(seen, std::memory_order_acquire);
            const unsigned j = workers[wv].go.load(std::memory_order_acquire);
            if (j == QUIT) return;
            if (j == seen) continue;
            seen = j;
            run_wave(waves[wv], block, wv * 64u, std::min(64u, nthreads - wv * 64u));
            if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == nwaves) {
                done_job.store(j, std::memory_order_release);
                done_job.notify_one();
            }
        }
    }
    void run(unsigned n, const std::function<void()> &fn) {
        if (n == 0 || n > MAXW * 64u) {
            fprintf(stderr, "simt_emu: workgroup of %u threads\n", n);
            abort();
        }
        nwaves = (n + 63) / 64;
        while (nworkers < nwaves) {
            const unsigned w = nworkers++;
            workers[w].th = std::thread([this, w] { worker(w); });
        }
        block.need = n;
        block.arrived = 0;
        block.body = &fn;
        ++block.serial;
        blockDim.x = n;
        blockDim.y = blockDim.z = 1;
        nthreads = n;
        done.store(0, std::memory_order_relaxed);
        ++job;
        if (job == QUIT) job = 1;
        for (unsigned w = 0; w < nwaves; ++w) {
            workers[w].go.store(job, std::memory_order_release);
            workers[w].go.notify_one();
        }
        unsigned dj = done_job.load(std::memory_order_acquire);
        while (dj != job) {
            done_job.wait(dj, std::memory_order_acquire);
            dj = done_job.load(std::memory_order_acquire);
        }
    }
    ~Pool() {
        for (unsigned w = 0; w < nworkers; ++w) {
            workers[w].go.store(QUIT, std::memory_order_release);
            workers[w].go.notify_one();
        }
        for (unsigned w = 0; w < nworkers; ++w) workers[w].th.join();
    }
};
inline Pool &pool() {
    static Pool p;
    return p;
}
// run `body` as ONE workgroup of nthreads threads
inline void run_block(unsigned nthreads, const std::function<void()> &body, unsigned grid_y = 1) {
    if (gridDim.x == 0) gridDim.x = 1;
    gridDim.y = grid_y;
    gridDim.z = 1;
    pool().run(nthreads, body);
}
// a grid of workgroups, one after the other (static __shared__ storage is one workgroup's LDS)
inline void run_grid(unsigned nblocks, unsigned nthreads, const std::function<void()> &body, unsigned nblocks_y = 1) {
    for (unsigned by = 0; by < nblocks_y; ++by)
        for (unsigned b = 0; b < nblocks; ++b) {
            blockIdx.x = b;
            blockIdx.y = by;
            blockIdx.z = 0;
            gridDim.x = nblocks;
            run_block(nthreads, body, nblocks_y);
        }
    gridDim.x = 1;
    gridDim.y = 1;
    blockIdx.x = 0;
    blockIdx.y = 0;
}
}  // namespace simt

// ---- device intrinsics ----
inline unsigned simt_lane() { return threadIdx.x & 63u; }
inline void __syncthreads() {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    simt::block_barrier();
}
inline void __threadfence_block() {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    simt::fence();
}
inline void __threadfence() { __threadfence_block(); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }  // (one lane, towards the host: no meeting)

inline uint64_t __ballot(int pred) {
    return simt::exchange(pred ? 1u : 0u, [](uint64_t tag) {
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
    const int from = (src < 0 || src > 63) ? (int)simt_lane() : src;
    const uint64_t r = simt::exchange(bits, [from](uint64_t tag) { return simt::fetch((unsigned)from, tag); });
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
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_waitcnt(int) {}

// DPP controls used by esort::wave_incl_scan: row_shr:n (0x110 + n), row_bcast:15 (0x142), row_bcast:31 (0x143); a lane without a
// source (or switched off by row_mask / bank_mask) keeps `old`
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool) {
    const unsigned lane = simt_lane(), row = lane >> 4, bank = (lane >> 2) & 3u;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11F) {
        const unsigned sh = (unsigned)ctrl - 0x110u;
        if ((lane & 15u) >= sh) from = (int)(lane - sh);
    } else if (ctrl == 0x142) {
        if (row >= 1) from = (int)(row * 16u - 1u);
    } else if (ctrl == 0x143) {
        if (lane >= 32) from = 31;
    } else {
        fprintf(stderr, "simt_emu: DPP control 0x%x not modelled\n", ctrl);
        abort();
    }
    const int got = simt_shfl_from(src, from < 0 ? (int)lane : from);  // (every lane takes part in the exchange)
    const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> bank) & 1);
    return (from >= 0 && enabled) ? got : old;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline unsigned long long clock64() { return 0; }
inline unsigned long long wall_clock64() {  // 100 MHz like the device's counter
    return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}

inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t atomicAnd(uint32_t *p, uint32_t v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t atomicCAS(uint32_t *p, uint32_t expected, uint32_t desired) {
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expected;
}
template <class T>
inline T simt_atomic_minmax(T *p, T v, bool want_max) {
    T o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while ((want_max ? o < v : o > v) && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
inline uint32_t atomicMax(uint32_t *p, uint32_t v) { return simt_atomic_minmax(p, v, true); }
inline uint32_t atomicMin(uint32_t *p, uint32_t v) { return simt_atomic_minmax(p, v, false); }
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { return simt_atomic_minmax(p, v, true); }
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) { return simt_atomic_minmax(p, v, false); }
using std::isfinite;
using std::max;
using std::min;

// ---- vector types and bit casts ----
struct float2 {
    float x, y;
};
struct alignas(16) float4 {
    float x, y, z, w;
};
struct alignas(8) uint2 {
    uint32_t x, y;
};
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct alignas(16) uint4 {
    uint32_t x, y, z, w;
};
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
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

// ---- the host API, synchronous ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipEventDisableSystemFence = 0x20000000 };
struct simt_stream_t {
    int id;
};
struct simt_event_t {
    int id;
};
typedef simt_stream_t *hipStream_t;
typedef simt_event_t *hipEvent_t;
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline const char *hipGetErrorString(hipError_t) { return "simt_emu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) {
    *least = 0;
    *greatest = -1;
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) {  // filled with a pattern: nothing may rely on fresh memory being zero
    const size_t bytes = ((n ? n : 1) + 255) & ~(size_t)255;
    *p = aligned_alloc(256, bytes);
    if (!*p) return 2;
    memset(*p, 0xA5, bytes);
    return hipSuccess;
}
inline hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) {
    *p = calloc(1, n ? n : 1);
    return *p ? hipSuccess : 2;
}
inline hipError_t hipHostFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipMemcpy(void *d, const void *s_, size_t n, hipMemcpyKind) {
    if (n) memmove(d, s_, n);
    return hipSuccess;
}
// (a __device__ variable is a plain global here)
#define HIP_SYMBOL(x) x
template <class T>
inline hipError_t hipMemcpyToSymbol(T &sym, const void *src, size_t n) {
    memcpy((void *)&sym, src, n);
    return hipSuccess;
}
template <class T>
inline hipError_t hipMemcpyFromSymbol(void *dst, const T &sym, size_t n) {
    memcpy(dst, (const void *)&sym, n);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) {
    if (n) memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s_, size_t n, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(d, s_, n, k); }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s_, int, size_t n, hipStream_t = nullptr) {  // (one address space here)
    memcpy(d, s_, n);
    return hipSuccess;
}
inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) {
    *can = 1;
    return hipSuccess;
}
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s_, unsigned, int) {
    *s_ = new simt_stream_t{0};
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s_, unsigned) {
    *s_ = new simt_stream_t{0};
    return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t s_) {
    delete s_;
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new simt_event_t{0};
    return hipSuccess;
}
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
    *ms = 0.001f;
    return hipSuccess;
}
inline bool simt_trace_on() {
    static const bool on = getenv("SIMT_EMU_TRACE") != nullptr;  // one line per launch, with its duration
    return on;
}
#define SIMT_LAUNCH(kern, grid, block, ...)                                                                                    \
    do {                                                                                                                       \
        const auto t0_ = std::chrono::steady_clock::now();                                                                     \
        simt::run_grid(dim3(grid).x, dim3(block).x, [&] { kern(__VA_ARGS__); }, dim3(grid).y);                                               \
        if (simt_trace_on())                                                                                                   \
            fprintf(stderr, "[simt_emu] %s <<<%u, %u>>> %.0f ms\n", #kern, dim3(grid).x, dim3(block).x,                        \
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count());               \
    } while (0)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) SIMT_LAUNCH(kern, grid, block, __VA_ARGS__)
#define hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, ev0, ev1, flags, ...) SIMT_LAUNCH(kern, grid, block, __VA_ARGS__)
#endif
