// exact_sort.hip.h — wavefront / workgroup drivers of exact_sort_core.h (gfx950, wave64).
//
//   block_esort()      : one workgroup sorts K[0..n) / V[0..n) (LDS or global pointers) exactly as
//                        std::sort would: level-synchronous segment queue, ONE WAVEFRONT PER SEGMENT
//                        (ballot-ranked stop lists), leaf finalisation by stable ranking.
//   block_partition()  : one workgroup performs a single partition of a large global-memory segment
//                        (used for the top levels of the query-scan voxel sort).
#ifndef ERASOR_EXACT_SORT_HIP_H
#define ERASOR_EXACT_SORT_HIP_H

#include <hip/hip_runtime.h>

#include "exact_sort_core.h"

namespace esort {

struct Seg {
    uint32_t first, last;
    int32_t depth;
};

__device__ __forceinline__ uint64_t lanemask_lt() {
    const uint32_t lane = threadIdx.x & 63u;
    return lane == 0 ? 0ull : (~0ull >> (64u - lane));
}

// inclusive prefix sum over the 64 lanes on DPP (row_shr 1 / 2 / 4 / 8 inside rows of 16, then row_bcast:15 / :31 across rows):
// six VALU operations, no LDS crossbar (a __shfl_up ladder is six dependent ds_bpermute round trips)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    uint32_t vv = x;
    vv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vv, 0x111, 0xf, 0xf, false);
    vv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vv, 0x112, 0xf, 0xf, false);
    vv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vv, 0x114, 0xf, 0xf, false);
    vv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vv, 0x118, 0xf, 0xf, false);
    vv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vv, 0x142, 0xa, 0xf, false);
    vv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vv, 0x143, 0xc, 0xf, false);
    return vv;
}

// intra-wave visibility of LDS/global writes made by other lanes of the same wave
__device__ __forceinline__ void wave_sync() { __threadfence_block(); }
// The same for LDS ONLY.  A wavefront's LDS operations are performed in program order, so a lane sees what another lane of its
// wavefront wrote to LDS by an earlier instruction without waiting for anything: all that is needed is that the COMPILER keeps the
// order.  wave_sync() is a workgroup-scope fence: it also waits for every global load and store the wavefront has in flight
// (s_waitcnt vmcnt(0)) -- behind a scattered store that is a memory round trip per call (round 4: eight of them per tile in
// k_mb_scatter_w).  (The tests' CPU stand-in, where the lanes of a wavefront are threads of their own, keeps the meeting point.)
__device__ __forceinline__ void wave_sync_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#else
    wave_sync();
#endif
}

// One wavefront partitions [first,last) (size > kThreshold).  posL/posR: scratch, same index space
// as K (entries [first+1,last) are used).  Returns the cut (wave-uniform).
//
// A partition is a chain of dependent memory round trips (~130 cycles each in LDS), not arithmetic: a short segment
// used to take ~16 of them (the sequential median move alone five).  This version needs five or six:
//   1. the old head, the three median candidates and the first 256 keys are fetched TOGETHER; the median swap is
//      applied to the registers (the key that moves to the median's place is patched in) while lane 0 stores it;
//   2. stop lists: four strips of loads in flight at a time;
//   3. m (a monotone predicate over the paired stop lists) by 64-ary search: two rounds for any m <= 4096;
//   4. the cut's two table entries ride along with the first swap indices; swaps run two strips at a time.
// The swaps performed and the cut are those of the sequential algorithm (exact_sort_core.h).
template <class KP, class VP, class PP>
__device__ __forceinline__ uint32_t wave_partition(KP K, VP V, PP posL, PP posR, uint32_t first, uint32_t last,
                                                   unsigned long long *ts = nullptr) {
#define WP_STAMP(i) do { if (ts && (threadIdx.x & 63u) == 0) ts[i] = clock64(); } while (0)
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lt = lanemask_lt();
    const uint32_t lo = first + 1, hi = last;
    WP_STAMP(0);
    // ---- median of (first+1, mid, last-1) to first: std::__move_median_to_first ----
    const uint32_t a = first + 1, b = first + (last - first) / 2, c = last - 1;
    const uint32_t cand = K[lane == 0 ? first : (lane == 1 ? a : (lane == 2 ? b : c))];
    uint32_t kq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = lo + (uint32_t)r * 64u + lane;
        kq[r] = i < hi ? (uint32_t)K[i] : 0u;
    }
    const uint32_t kf = __builtin_amdgcn_readlane(cand, 0), ka = __builtin_amdgcn_readlane(cand, 1), kb = __builtin_amdgcn_readlane(cand, 2),
                   kc = __builtin_amdgcn_readlane(cand, 3);
    uint32_t mpos, p;
    if (ka < kb) {
        if (kb < kc) { mpos = b; p = kb; }
        else if (ka < kc) { mpos = c; p = kc; }
        else { mpos = a; p = ka; }
    } else if (ka < kc) { mpos = a; p = ka; }
    else if (kb < kc) { mpos = c; p = kc; }
    else { mpos = b; p = kb; }
    if (lane == 0) {  // iter_swap(first, median), unconditional like the library's
        const uint32_t v0 = V[first], v1 = V[mpos];
        K[first] = p;
        K[mpos] = kf;
        V[first] = v1;
        V[mpos] = v0;
    }
    // ---- stop lists: L = positions (ascending) with !(k < p), R = positions with !(p < k) ----
    WP_STAMP(1);
    uint32_t nL = 0, nR = 0;
    for (uint32_t base = lo; base < hi; base += 256) {
        if (base != lo) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t i = base + (uint32_t)r * 64u + lane;
                kq[r] = i < hi ? (uint32_t)K[i] : 0u;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t i = base + (uint32_t)r * 64u + lane;
            const bool valid = i < hi;
            const uint32_t k = i == mpos ? kf : kq[r];  // (the median's place holds the old head now, whatever the load saw)
            const bool isL = valid && !(k < p);
            const bool isR = valid && !(p < k);
            const uint64_t mL = __ballot(isL), mR = __ballot(isR);
            if (isL) posL[lo + nL + __popcll(mL & lt)] = i;
            if (isR) posR[lo + nR + __popcll(mR & lt)] = i;
            nL += __popcll(mL);
            nR += __popcll(mR);
        }
    }
    wave_sync();
    WP_STAMP(2);
    // ---- m = #{k < min(nL, nR) : L[k] < R[k]}, R[k] = posR[lo + nR - 1 - k]; the predicate is monotone ----
    const uint32_t lim = nL < nR ? nL : nR;
    uint32_t mlo = 0, cnt = lim;  // invariant: the predicate holds below mlo and m <= mlo + cnt
    while (cnt > 64) {
        const uint32_t step = (cnt + 63u) / 64u;
        const uint32_t k = mlo + lane * step;
        const bool ok = (k < mlo + cnt) && ((uint32_t)posL[lo + k] < (uint32_t)posR[lo + nR - 1 - k]);
        const uint32_t t = (uint32_t)__popcll(__ballot(ok));
        if (t == 0) {
            cnt = 0;
            break;
        }
        const uint32_t nlo = mlo + (t - 1u) * step + 1u;
        uint32_t nhi = mlo + t * step;
        if (t == 64u || nhi > mlo + cnt) nhi = mlo + cnt;
        cnt = nhi - nlo;
        mlo = nlo;
    }
    uint32_t m;
    {
        const uint32_t k = mlo + lane;
        const bool ok = (lane < cnt) && ((uint32_t)posL[lo + k] < (uint32_t)posR[lo + nR - 1 - k]);
        m = mlo + (uint32_t)__popcll(__ballot(ok));
    }
    WP_STAMP(3);
    // ---- cut = min(L[m] if it exists, R[m-1] if m > 0); swaps (L[k], R[k]) for k < m (pairwise disjoint positions) ----
    const uint32_t cl = posL[lo + (m < nL ? m : 0u)], cr = posR[lo + (m > 0 ? nR - m : 0u)];  // (both lists are non-empty)
    for (uint32_t k0 = 0; k0 < m; k0 += 128) {
        const uint32_t k1 = k0 + lane, k2 = k0 + 64u + lane;
        const bool v1 = k1 < m, v2 = k2 < m;
        uint32_t i1 = 0, j1 = 0, i2 = 0, j2 = 0;
        if (v1) {
            i1 = posL[lo + k1];
            j1 = posR[lo + nR - 1 - k1];
        }
        if (v2) {
            i2 = posL[lo + k2];
            j2 = posR[lo + nR - 1 - k2];
        }
        uint32_t a1 = 0, b1 = 0, c1 = 0, d1 = 0, a2 = 0, b2 = 0, c2 = 0, d2 = 0;
        if (v1) {
            a1 = K[i1];
            b1 = K[j1];
            c1 = V[i1];
            d1 = V[j1];
        }
        if (v2) {
            a2 = K[i2];
            b2 = K[j2];
            c2 = V[i2];
            d2 = V[j2];
        }
        if (v1) {
            K[i1] = b1;
            K[j1] = a1;
            V[i1] = d1;
            V[j1] = c1;
        }
        if (v2) {
            K[i2] = b2;
            K[j2] = a2;
            V[i2] = d2;
            V[j2] = c2;
        }
    }
    uint32_t cut = m < nL ? cl : 0xFFFFFFFFu;
    if (m > 0 && cr < cut) cut = cr;
    wave_sync();
    WP_STAMP(4);
    if (ts && lane == 0) ts[5] = m;
#undef WP_STAMP
    return cut;
}

// The whole introsort subtree of a segment of <= 64 keys, one key per lane, in registers: every partition is two
// ballots, a scalar loop over the stop masks (pairs L[k] <-> R[k]) and ONE permute.  Leaves (<= 16 keys) are marked in
// head[] for the caller's stable leaf ranking.  Exactly the same swaps as wave_partition / the sequential algorithm.
template <class KP, class VP, class HP>
__device__ __forceinline__ void wave_small_subtree(KP K, VP V, HP head, uint32_t first, uint32_t last, int32_t depth, uint32_t *n_fallback,
                                                   uint32_t *wsc /* 128 uint32 of LDS owned by this wavefront */) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lt = lanemask_lt();
    const uint64_t gt = ~(lt | (1ull << lane));  // lanes above this one
    const uint32_t n = last - first;             // 17..64
    uint32_t k = lane < n ? (uint32_t)K[first + lane] : 0xFFFFFFFFu;
    uint32_t v = lane < n ? (uint32_t)V[first + lane] : 0u;
    uint64_t heads = 0;  // bit i: a leaf starts at lane i (i > 0)
    uint32_t *wsL = wsc, *wsR = wsc + 64;
    // pending sub-ranges (lane indices), at most 64/17 = 3 alive at once
    uint32_t sf[4], se[4];
    int32_t sd[4];
    int sp = 0;
    sf[0] = 0; se[0] = n; sd[0] = depth; sp = 1;
    while (sp > 0) {
        --sp;
        uint32_t f = sf[0], e = se[0];
        int32_t d = sd[0];
#pragma unroll
        for (int t = 1; t < 4; ++t)
            if (sp == t) { f = sf[t]; e = se[t]; d = sd[t]; }
        if (d == 0) {
            // depth budget exhausted: exact heapsort of this sub-range through memory (rare)
            if (lane < n) { K[first + lane] = k; V[first + lane] = v; }
            wave_sync();
            if (lane == 0) {
                heapsort_exact(K, V, first + f, first + e);
                atomicAdd(n_fallback, 1u);
            }
            wave_sync();
            if (lane < n) { k = K[first + lane]; v = V[first + lane]; }
            for (uint32_t i = f + 1; i < e; ++i) heads |= 1ull << i;  // fully sorted: every key its own leaf
            continue;
        }
        // __move_median_to_first(f, f+1, mid, e-1)
        const uint32_t a = f + 1, b = f + (e - f) / 2, c = e - 1;
        const uint32_t ka = __builtin_amdgcn_readlane(k, a), kb = __builtin_amdgcn_readlane(k, b), kc = __builtin_amdgcn_readlane(k, c);
        uint32_t mp;
        if (ka < kb) mp = (kb < kc) ? b : ((ka < kc) ? c : a);
        else mp = (ka < kc) ? a : ((kb < kc) ? c : b);
        const uint32_t kf = __builtin_amdgcn_readlane(k, f), vf = __builtin_amdgcn_readlane(v, f);
        const uint32_t p = __builtin_amdgcn_readlane(k, mp), vm = __builtin_amdgcn_readlane(v, mp);
        if (lane == f) { k = p; v = vm; }
        if (lane == mp) { k = kf; v = vf; }
        const bool inr = lane > f && lane < e;
        const bool isL = inr && !(k < p), isR = inr && !(p < k);
        const uint64_t ml = __ballot(isL), mr = __ballot(isR);
        // rank tables: wsL[k] = k-th left stop (ascending), wsR[k] = k-th right stop counted from the top
        const uint32_t rkL = (uint32_t)__popcll(ml & lt), rkR = (uint32_t)__popcll(mr & gt);
        if (isL) wsL[rkL] = lane;
        if (isR) wsR[rkR] = lane;
        wave_sync();
        const uint32_t nL = (uint32_t)__popcll(ml), nR = (uint32_t)__popcll(mr), lim = nL < nR ? nL : nR;
        const uint32_t lk = lane < nL ? wsL[lane] : 0xFFFFFFFFu, rk = lane < nR ? wsR[lane] : 0u;
        const uint32_t m = (uint32_t)__popcll(__ballot(lane < lim && lk < rk));  // monotone predicate: count == first false
        uint32_t cut = 0xFFFFFFFFu;
        if (m < nL) cut = __builtin_amdgcn_readlane(lk, m);
        if (m > 0) {
            const uint32_t r = __builtin_amdgcn_readlane(rk, m - 1);
            if (r < cut) cut = r;
        }
        uint32_t src = lane;
        if (isL && rkL < m) src = wsR[rkL];
        if (isR && rkR < m) src = wsL[rkR];
        wave_sync();  // the tables are rewritten by the next partition
        k = __shfl(k, (int)src, 64);
        v = __shfl(v, (int)src, 64);
        heads |= 1ull << cut;
        // children (order irrelevant): [f,cut) and [cut,e)
        if (cut - f > (uint32_t)kThreshold) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (sp == t) { sf[t] = f; se[t] = cut; sd[t] = d - 1; }
            ++sp;
        }
        if (e - cut > (uint32_t)kThreshold) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (sp == t) { sf[t] = cut; se[t] = e; sd[t] = d - 1; }
            ++sp;
        }
    }
    if (lane < n) {
        K[first + lane] = k;
        V[first + lane] = v;
        if (lane > 0 && ((heads >> lane) & 1ull)) atomicOr(&head[(first + lane) >> 5], 1u << ((first + lane) & 31u));
    }
    wave_sync();
}

template <int EPT, class KP, class VP, class PP>
__device__ __forceinline__ uint32_t block_partition(KP K, VP V, PP posL, PP posR, uint32_t first, uint32_t last, uint32_t *sm,
                                                    bool median_done = false);

static constexpr uint32_t kBlockMin = 2048;  // inside block_esort, segments longer than this are partitioned by the whole workgroup

// Whole-subtree sort of K[base..base+n) by one workgroup.  All pointers index the same space
// (element e lives at K[e]); the segment handled is [seg_first, seg_last) with introsort depth
// budget `depth`.  qa/qb: LDS queues of capacity qcap; qcnt: 2 LDS counters; head: byte flags for
// [seg_first, seg_last] (indexable with the same element indices).  On return K2/V2[seg range] hold
// the final order.  K2/V2 may alias posL/posR (they are dead by then) but not K/V.
template <class KP, class VP, class PP, class HP, class K2P, class V2P>
__device__ __forceinline__ void block_esort(KP K, VP V, PP posL, PP posR, HP head, K2P K2, V2P V2, uint32_t seg_first, uint32_t seg_last,
                            int32_t depth, Seg *qa, Seg *qb, uint32_t *qcnt, uint32_t qcap, uint32_t *n_fallback,
                            uint32_t *overflow_flag, unsigned long long *tstamp = nullptr) {
#define ESORT_STAMP(i) do { if (tstamp && threadIdx.x == 0) tstamp[i] = clock64(); } while (0)
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    const uint32_t wave = tid >> 6, nwaves = bs >> 6, lane = tid & 63u;
    const uint32_t n = seg_last - seg_first;
    // head flags: one bit per element index (word e>>5, bit e&31); a set bit starts a leaf.  Words are shared with the
    // neighbouring segments at the boundaries, so everything goes through atomics on whole words.
    for (uint32_t w = (seg_first >> 5) + tid; w <= (seg_last >> 5); w += bs) {
        // clear only the bits strictly inside (seg_first, seg_last): boundary bits are shared with the neighbours
        const uint32_t lo_e = max(seg_first + 1u, w << 5), hi_e = min(seg_last, (w << 5) + 32u);
        if (lo_e < hi_e) {
            const uint32_t hb = hi_e - (w << 5), lb = lo_e - (w << 5);
            const uint32_t m = (hb == 32u ? 0xFFFFFFFFu : ((1u << hb) - 1u)) & ~((1u << lb) - 1u);
            atomicAnd(&head[w], ~m);
        }
    }
    __syncthreads();
    if (tid == 0) {
        atomicOr(&head[seg_first >> 5], 1u << (seg_first & 31u));
        atomicOr(&head[seg_last >> 5], 1u << (seg_last & 31u));
    }
    if (tid == 0) {
        qcnt[0] = 0;
        qcnt[1] = 0;
        if (n > (uint32_t)kThreshold) {
            qa[0].first = seg_first;
            qa[0].last = seg_last;
            qa[0].depth = depth;
            qcnt[0] = 1;
        }
    }
    __shared__ uint32_t s_wsc[16][128];  // per-wavefront rank tables of wave_small_subtree
    __syncthreads();
    ESORT_STAMP(0);
    // phase 1: long segments are partitioned by the WHOLE workgroup, one after the other (a short stack); their
    // children go back on the stack or, once short enough, to the wavefront-per-segment queue of phase 2.
    // The queue holds one level of pairwise disjoint segments longer than kThreshold, i.e. at most n / 17 + 1 entries:
    // while n <= 16 * (qcap - 2) every short piece joins ONE queue run at the end (most parallel); a longer segment
    // (only reachable on the global-memory paths) runs the queue once per <= kBlockMin piece instead, which keeps every
    // run within kBlockMin / 17 + 1 <= qcap entries -- no input can overflow it.
    __shared__ Seg s_small[4096 / 16 + 2];  // segments of <= 64 keys, finished by wave_small_subtree after the level loop
    __shared__ uint32_t s_nsmall, s_small_next;
    const uint32_t small_cap = qcap < (uint32_t)(4096 / 16 + 2) ? qcap : (uint32_t)(4096 / 16 + 2);
    __shared__ Seg stk[72];  // DFS: one sibling per level, depth <= 2 lg n <= 64
    __shared__ uint32_t sm_bp[40];
    __shared__ int sp, s_mode;
    __shared__ Seg cur_sg;
    const bool seq_pieces = n > 16u * (qcap - 2u) || qcap < kBlockMin / 16u + 2u;
    enum { M_EXIT = 0, M_PART = 1, M_QUEUE = 2 };
    if (tid == 0) {
        sp = 0;
        if ((n > kBlockMin && depth > 0) || (seq_pieces && n > (uint32_t)kThreshold)) {
            stk[0].first = seg_first;
            stk[0].last = seg_last;
            stk[0].depth = depth;
            sp = 1;
            qcnt[0] = 0;
        }
    }
    __syncthreads();
    bool drained = false;  // block-uniform
    int lvl_ = 0;
    for (;;) {
        if (tid == 0) {
            if (sp > 0) {
                cur_sg = stk[--sp];
                const uint32_t len = cur_sg.last - cur_sg.first;
                if (len > kBlockMin && cur_sg.depth > 0) s_mode = M_PART;
                else {  // a short (or depth-exhausted) piece on its own: only in seq_pieces mode
                    qa[0] = cur_sg;
                    qcnt[0] = 1;
                    qcnt[1] = 0;
                    s_mode = M_QUEUE;
                }
            } else
                s_mode = drained ? M_EXIT : M_QUEUE;
        }
        __syncthreads();
        const int mode = s_mode;
        const Seg sg0 = cur_sg;
        const bool stack_empty = sp == 0;
        __syncthreads();
        if (mode == M_EXIT) break;
        if (mode == M_PART) {
            const uint32_t cut = block_partition<4>(K, V, posL, posR, sg0.first, sg0.last, sm_bp);
            if (tid == 0) {
                atomicOr(&head[cut >> 5], 1u << (cut & 31u));
                const Seg ch[2] = {{sg0.first, cut, sg0.depth - 1}, {cut, sg0.last, sg0.depth - 1}};
                for (int t = 0; t < 2; ++t) {
                    const uint32_t len = ch[t].last - ch[t].first;
                    if (len <= (uint32_t)kThreshold) continue;
                    if ((len > kBlockMin && ch[t].depth > 0) || seq_pieces) {
                        if (sp < 72) stk[sp++] = ch[t];
                        else *overflow_flag = 4;
                    } else {
                        const uint32_t at = qcnt[0];
                        if (at < qcap) {
                            qa[at] = ch[t];
                            qcnt[0] = at + 1;
                        } else
                            *overflow_flag = 3;
                    }
                }
            }
            __syncthreads();
            continue;
        }
        if (stack_empty && mode == M_QUEUE) drained = true;  // this run also takes whatever phase 1 queued: nothing is left after it
        if (lvl_ == 0) ESORT_STAMP(1);
        // phase 2: level-synchronous queue, one wavefront per segment.  Segments of <= 64 keys do not take part in the
        // levels (a register-resident subtree costs several partitions' worth of time and would stall every level it
        // appears in): they are set aside and handed out to the wavefronts, dynamically, once the levels are done.
        if (tid == 0) {
            s_nsmall = 0;
            s_small_next = 0;
        }
        __syncthreads();
        int cur = 0;
    for (;;) {
        const uint32_t nseg = __builtin_amdgcn_readfirstlane(qcnt[cur]);
        if (nseg == 0) break;
        if (lvl_ < 12) ESORT_STAMP(4 + lvl_);
        ++lvl_;
        Seg *q = cur ? qb : qa;
        Seg *qn = cur ? qa : qb;
        for (uint32_t s = wave; s < nseg; s += nwaves) {
            const Seg sg = q[s];
            if (sg.depth == 0) {
                if (lane == 0) {
                    heapsort_exact(K, V, sg.first, sg.last);
                    // a heapsorted segment is fully sorted: make every element its own leaf
                    for (uint32_t i = sg.first; i < sg.last; ++i) atomicOr(&head[i >> 5], 1u << (i & 31u));
                    atomicAdd(n_fallback, 1u);
                }
                continue;
            }
            if (sg.last - sg.first <= 64u) {  // (a short initial segment) whole subtree in registers, later
                if (lane == 0) {
                    const uint32_t at = atomicAdd(&s_nsmall, 1u);
                    if (at < small_cap) s_small[at] = sg; else *overflow_flag = 3;
                }
                continue;
            }
            const uint32_t cut = wave_partition(K, V, posL, posR, sg.first, sg.last, (tstamp && lvl_ == 1) ? tstamp + 16 : nullptr);
            if (tstamp && lvl_ == 1 && lane == 0) tstamp[22] = clock64();
            if (lane == 0) atomicOr(&head[cut >> 5], 1u << (cut & 31u));
            if (lane < 2) {  // the two children are queued side by side (their order in a queue is irrelevant)
                Seg ch;
                ch.first = lane == 0 ? sg.first : cut;
                ch.last = lane == 0 ? cut : sg.last;
                ch.depth = sg.depth - 1;
                const uint32_t len = ch.last - ch.first;
                if (len > (uint32_t)kThreshold) {
                    if (len <= 64u && ch.depth > 0) {
                        const uint32_t at = atomicAdd(&s_nsmall, 1u);
                        if (at < small_cap) s_small[at] = ch; else *overflow_flag = 3;
                    } else {
                        const uint32_t at = atomicAdd(&qcnt[cur ^ 1], 1u);
                        if (at < qcap) qn[at] = ch; else *overflow_flag = 3;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (qcnt[cur ^ 1] > qcap) qcnt[cur ^ 1] = qcap;
            qcnt[cur] = 0;
        }
        __syncthreads();
        cur ^= 1;
    }
        {   // the set-aside short segments, one wavefront each, handed out dynamically (their cost varies 4x with the length)
            const uint32_t ns = s_nsmall < small_cap ? s_nsmall : small_cap;
            for (;;) {
                uint32_t i = 0;
                if (lane == 0) i = atomicAdd(&s_small_next, 1u);
                i = __shfl(i, 0, 64);
                if (i >= ns) break;
                const Seg sg = s_small[i];
                wave_small_subtree(K, V, head, sg.first, sg.last, sg.depth, n_fallback, &s_wsc[wave][0]);
            }
            __syncthreads();
        }
    }
    __syncthreads();
    ESORT_STAMP(2);
    // leaves -> stable ranks.  Results go to registers first because K2/V2 may alias posL/posR only,
    // never K/V, so a direct write is safe.
    for (uint32_t i = seg_first + tid; i < seg_last; i += bs) {
        // leaf [a,b) of element i from the head bitmask: a = highest set bit <= i, b = lowest set bit > i (leaves are
        // <= 16 long, so both lie within 16 positions: at most two 32-bit words each)
        const uint32_t wi = i >> 5, bi = i & 31u;
        const uint32_t w0 = head[wi];
        const uint32_t lowm = w0 & ((2u << bi) - 1u);  // bits <= bi
        uint32_t a, b;
        if (lowm) a = (wi << 5) + (31u - (uint32_t)__builtin_clz(lowm));
        else {
            const uint32_t wp = wi > (seg_first >> 5) ? head[wi - 1] : 0u;
            a = wp ? ((wi - 1) << 5) + (31u - (uint32_t)__builtin_clz(wp)) : seg_first;
        }
        const uint32_t highm = bi == 31u ? 0u : (w0 & ~((2u << bi) - 1u));  // bits > bi
        if (highm) b = (wi << 5) + (uint32_t)__builtin_ctz(highm);
        else {
            const uint32_t wn = wi < (seg_last >> 5) ? head[wi + 1] : 0u;
            b = wn ? ((wi + 1) << 5) + (uint32_t)__builtin_ctz(wn) : seg_last;
        }
        // only reachable after a segment-queue overflow (already flagged as an error): stay inside the segment
        if (a < seg_first) a = seg_first;
        if (b > seg_last) b = seg_last;
        if (i >= a + (uint32_t)kThreshold) a = i - (uint32_t)kThreshold + 1u;
        // stable rank inside the leaf: 16 independent loads, predicated
        const uint32_t ki = K[i];
        uint32_t r = 0;
#pragma unroll
        for (int t = 0; t < kThreshold; ++t) {
            const uint32_t j = a + t;
            const bool valid = j < b;
            const uint32_t kj = valid ? (uint32_t)K[j] : 0u;
            r += (valid && ((kj < ki) || (kj == ki && j < i))) ? 1u : 0u;
        }
        K2[a + r] = ki;
        V2[a + r] = V[i];
    }
    __syncthreads();
    ESORT_STAMP(3);
#undef ESORT_STAMP
}

// One workgroup performs ONE partition of the global-memory segment [first,last).  sm: >= 2*nwaves+4 uint32.
// Every thread owns EPT consecutive positions of a tile, so a tile is bs*EPT elements and the stop lists are
// written in ascending position order with one block scan per tile.
template <int EPT, class KP, class VP, class PP>
__device__ __forceinline__ uint32_t block_partition(KP K, VP V, PP posL, PP posR, uint32_t first, uint32_t last, uint32_t *sm,
                                                    bool median_done) {
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    const uint32_t wave = tid >> 6, nwaves = bs >> 6, lane = tid & 63u;
    if (tid == 0 && !median_done) move_median_to_first(K, V, first, last);
    __threadfence_block();
    __syncthreads();
    const uint32_t p = K[first];
    const uint32_t lo = first + 1, hi = last;
    uint32_t carryL = 0, carryR = 0;  // uniform across the block (recomputed identically by all threads)
    for (uint32_t base = lo; base < hi; base += bs * EPT) {
        const uint32_t i0 = base + tid * EPT;
        uint32_t k[EPT];
        uint32_t fl = 0, fr = 0;  // bit j: element i0+j is a left / right stop
#pragma unroll
        for (int j = 0; j < EPT; ++j) k[j] = (i0 + j < hi) ? (uint32_t)K[i0 + j] : 0u;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const bool valid = i0 + j < hi;
            fl |= (valid && !(k[j] < p)) ? (1u << j) : 0u;
            fr |= (valid && !(p < k[j])) ? (1u << j) : 0u;
        }
        const uint32_t cl = __popc(fl), cr = __popc(fr);
        // wave-inclusive scans of (cl, cr) packed in one 32-bit word (each < 2^16 per wave: 64*EPT)
        const uint32_t inc = wave_incl_scan(cl | (cr << 16));  // (DPP row shifts; round 2: a six-step __shfl_up ladder through the LDS crossbar)
        if (lane == 63) {
            sm[wave] = inc & 0xFFFFu;
            sm[nwaves + wave] = inc >> 16;
        }
        __syncthreads();
        // second level on DPP as well (a tile holds at most bs * EPT <= 8192 stops of either kind: the packed halves cannot carry)
        const uint32_t wt = lane < nwaves ? (sm[lane] | (sm[nwaves + lane] << 16)) : 0u;
        const uint32_t wi = wave_incl_scan(wt);
        const uint32_t tot2 = __builtin_amdgcn_readlane(wi, 63), pre2 = __builtin_amdgcn_readlane(wi - wt, __builtin_amdgcn_readfirstlane(wave));
        const uint32_t preL = pre2 & 0xFFFFu, preR = pre2 >> 16, totL = tot2 & 0xFFFFu, totR = tot2 >> 16;
        uint32_t oL = lo + carryL + preL + (inc & 0xFFFFu) - cl;
        uint32_t oR = lo + carryR + preR + (inc >> 16) - cr;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            if (fl & (1u << j)) posL[oL++] = i0 + j;
            if (fr & (1u << j)) posR[oR++] = i0 + j;
        }
        carryL += totL;
        carryR += totR;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t nL = carryL, nR = carryR;
    const uint32_t lim = nL < nR ? nL : nR;
    // m = number of k < lim with posL[k] < R[k]; monotone -> count in parallel
    uint32_t cnt = 0;
    for (uint32_t k0 = tid; k0 < lim; k0 += 4 * bs) {
        uint32_t l4[4], r4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t k = k0 + u * bs;
            l4[u] = k < lim ? (uint32_t)posL[lo + k] : 1u;
            r4[u] = k < lim ? (uint32_t)posR[lo + nR - 1 - k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt += (l4[u] < r4[u]) ? 1u : 0u;
    }
    cnt = __builtin_amdgcn_readlane(wave_incl_scan(cnt), 63);
    if (lane == 0) sm[wave] = cnt;
    __syncthreads();
    const uint32_t m = __builtin_amdgcn_readlane(wave_incl_scan(lane < nwaves ? sm[lane] : 0u), 63);
    __syncthreads();
    uint32_t cut = 0xFFFFFFFFu;
    if (m < nL) cut = posL[lo + m];
    if (m > 0) {
        const uint32_t r = posR[lo + nR - m];
        if (r < cut) cut = r;
    }
    __syncthreads();
    // swaps touch pairwise-distinct positions: batch 4 per thread so that 4 x (2 position + 4 key/value) loads are in flight
    for (uint32_t k0 = tid; k0 < m; k0 += 4 * bs) {
        uint32_t a[4], b[4], ka[4], kb[4], va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t k = k0 + u * bs;
            const bool ok = k < m;
            a[u] = ok ? (uint32_t)posL[lo + k] : first;
            b[u] = ok ? (uint32_t)posR[lo + nR - 1 - k] : first;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ka[u] = K[a[u]];
            kb[u] = K[b[u]];
            va[u] = V[a[u]];
            vb[u] = V[b[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u * bs < m) {
                K[a[u]] = kb[u];
                K[b[u]] = ka[u];
                V[a[u]] = vb[u];
                V[b[u]] = va[u];
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    return cut;
}

// ------------------------------------------------------------------------------------------------
// block_esort_sync: the same introsort, LEVEL-SYNCHRONOUS over a whole workgroup, for n <= EMAX * blockDim.x keys
// that live in registers (position i = e * blockDim.x + tid holds k[e] / v[e]) and in an LDS mirror (sKV).
//
// Every position carries its segment [f, l) and depth budget d in registers; one loop iteration performs ONE partition of
// EVERY segment longer than kThreshold at once.  What makes a level cheap is that nothing has to be searched:
//   PL(i)  = # left stops  (!(key < pivot)) at positions <  i   (whole array; ballot + one 64-entry table scan)
//   SG(i)  = # right stops (!(pivot < key)) at positions >  i
//   A = PL(i) - PL(f+1) = rank of a left stop in its segment's list L;  G = SG(i) - SG(l-1) = rank of a right stop in R;
//   a left stop is swapped  <=>  L[A] < R[A]  <=>  G > A;    a right stop is swapped  <=>  L[G] < R[G]  <=>  A > G
//   (exact_sort_core.h: swaps = pairs (L[k], R[k]), k < m, the predicate is monotone); the stop lists carry the stops'
//   (key, value) in global rank order, so the partner's pair is ONE read: R[A] resp. L[G];
//   cut = min(L[m], R[m-1]) = the lowest position that is an UNswapped left stop or a SWAPPED right stop, found with one
//   LDS atomicMin per (wavefront, segment).
// A level is a chain of five dependent LDS round trips and four workgroup barriers, whatever the number of segments (the
// table scan runs on DPP row shifts, not on ds_bpermute); the one-wavefront-per-segment formulation (block_esort) pays
// >= 2.2 k cycles per partition and 10 k for the top level of a 1000-key bin.  A segment whose depth budget is exhausted
// is heapsorted by one lane (exact, rare), like in block_esort.
// Results: K2[0..n) / V2[0..n) (may alias sLL / sRR, not sKV).  sTab: 68 words.  Requires blockDim.x <= 1024, EMAX <= 4.
// ------------------------------------------------------------------------------------------------
struct KVKeyRef {
    uint2 *p;
    __device__ __forceinline__ uint32_t &operator[](size_t i) const { return p[i].x; }
};
struct KVValRef {
    uint2 *p;
    __device__ __forceinline__ uint32_t &operator[](size_t i) const { return p[i].y; }
};

template <int EMAX, class K2P, class V2P>
__device__ __forceinline__ void block_esort_sync(uint32_t (&k)[EMAX], uint32_t (&v)[EMAX], uint32_t n, uint2 *sKV, uint2 *sLL, uint2 *sRR,
                                                 uint32_t *sPS, uint32_t *sCut, uint32_t *sTab, K2P K2, V2P V2, uint32_t *n_fallback,
                                                 unsigned long long *tstamp = nullptr, int32_t depth_budget = -1) {
    const uint32_t tid = threadIdx.x, bs = blockDim.x, lane = tid & 63u, wave = tid >> 6, nw = bs >> 6;
    const uint64_t lt = lanemask_lt();
    const uint64_t gt = ~(lt | (1ull << lane));
    const uint32_t E = (n + bs - 1) / bs;  // rows of positions in use (<= EMAX)
    uint32_t f[EMAX], l[EMAX];
    int32_t d[EMAX];
    // (a segment of a larger sort arrives with what is left of ITS introsort depth budget)
    const int32_t depth0 = depth_budget >= 0 ? depth_budget : 2 * lg2_floor(n ? n : 1u);
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const uint32_t i = (uint32_t)e * bs + tid;
        f[e] = 0;
        l[e] = i < n ? n : 0u;  // (positions beyond n: an empty segment, never active)
        d[e] = depth0;
        if (i < n) sKV[i] = make_uint2(k[e], v[e]);
    }
    // [0, 64): (row, wavefront) counts; [64]: a segment is left; [65]: a heapsort is due (a segment that arrives with no budget left)
    if (tid < 68) sTab[tid] = ((tid == 64 || (tid == 65 && depth0 == 0)) && n > (uint32_t)kThreshold) ? 1u : 0u;
    __syncthreads();
    if (tstamp && tid == 0) tstamp[0] = clock64();
    int lvl = 0;
    for (;;) {
        // ---- phase 0: pivot of my segment (std::__move_median_to_first), my key after the median move, stop flags ----
        // (the loop's flags are fetched together with the median candidates: one round trip)
        const uint32_t fl_any = sTab[64], fl_heap = sTab[65];
        uint2 qf[EMAX], qa[EMAX], qb[EMAX], qc[EMAX];
        bool act[EMAX];
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            act[e] = (uint32_t)e < E && l[e] - f[e] > (uint32_t)kThreshold;
            qf[e] = qa[e] = qb[e] = qc[e] = make_uint2(0u, 0u);
            if (act[e]) {
                qf[e] = sKV[f[e]];
                qa[e] = sKV[f[e] + 1];
                qb[e] = sKV[f[e] + (l[e] - f[e]) / 2];
                qc[e] = sKV[l[e] - 1];
            }
        }
        if (!fl_any) break;
        if (tstamp && tid == 0 && lvl < 14) tstamp[1 + lvl] = clock64();
        ++lvl;
        if (fl_heap) {  // (workgroup-uniform) depth budget exhausted somewhere: exact heapsort of those segments, one lane each
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const uint32_t i = (uint32_t)e * bs + tid;
                if (act[e] && d[e] == 0 && i == f[e]) {
                    heapsort_exact(KVKeyRef{sKV}, KVValRef{sKV}, f[e], l[e]);
                    atomicAdd(n_fallback, 1u);
                }
            }
            __threadfence_block();
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const uint32_t i = (uint32_t)e * bs + tid;
                if (act[e] && d[e] == 0) {  // fully sorted now: every key its own leaf
                    const uint2 kv = sKV[i];
                    k[e] = kv.x;
                    v[e] = kv.y;
                    f[e] = i;
                    l[e] = i + 1;
                    act[e] = false;
                }
            }
        }
        bool isL[EMAX], isR[EMAX], chg[EMAX], any_act = false;
        uint64_t mL[EMAX], mR[EMAX];
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const uint32_t i = (uint32_t)e * bs + tid;
            chg[e] = false;
            isL[e] = isR[e] = false;
            any_act = any_act || act[e];
            if (act[e]) {
                const uint32_t a = f[e] + 1, b = f[e] + (l[e] - f[e]) / 2, c = l[e] - 1;
                // median of (a, b, c) by key, std::__move_median_to_first's decision tree as selects
                const bool ab = qa[e].x < qb[e].x, bc = qb[e].x < qc[e].x, ac = qa[e].x < qc[e].x;
                const bool take_b = ab ? bc : (!ac && !bc), take_c = ab ? (!bc && ac) : (!ac && bc);
                const uint32_t mpos = take_b ? b : (take_c ? c : a);
                const uint32_t pk = take_b ? qb[e].x : (take_c ? qc[e].x : qa[e].x), pv = take_b ? qb[e].y : (take_c ? qc[e].y : qa[e].y);
                if (i == f[e]) { k[e] = pk; v[e] = pv; chg[e] = true; }
                else if (i == mpos) { k[e] = qf[e].x; v[e] = qf[e].y; chg[e] = true; }
                const bool inr = i > f[e];
                isL[e] = inr && !(k[e] < pk);
                isR[e] = inr && !(pk < k[e]);
            }
            mL[e] = __ballot(isL[e]);
            mR[e] = __ballot(isR[e]);
            if (lane == 0 && (uint32_t)e < E) sTab[(uint32_t)e * nw + wave] = (uint32_t)__popcll(mL[e]) | ((uint32_t)__popcll(mR[e]) << 16);
        }
        // a wavefront none of whose positions lies in a segment that is still being partitioned only keeps the barriers company
        const bool wact = __ballot(any_act) != 0ull;
        __syncthreads();  // #1
        if (tid == 0) { sTab[64] = 0; sTab[65] = 0; }  // (everybody has read the flags by now; they are raised again after barrier #3)
        uint32_t PL[EMAX], SG[EMAX];
        if (wact) {
            // ---- phase 1: every wavefront scans the (row, wavefront) table itself: 64 entries, counts packed 16 | 16 ----
            const uint32_t tv = sTab[lane];
            const uint32_t inc = wave_incl_scan(tv);
            const uint32_t totR = __builtin_amdgcn_readlane(inc, 63) >> 16;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const uint32_t t = __builtin_amdgcn_readfirstlane(((uint32_t)e * nw + wave) & 63u);
                const uint32_t it = __builtin_amdgcn_readlane(inc, t), xt = __builtin_amdgcn_readlane(tv, t);
                PL[e] = ((it - xt) & 0xFFFFu) + (uint32_t)__popcll(mL[e] & lt);       // left stops before me
                SG[e] = (totR - (it >> 16)) + (uint32_t)__popcll(mR[e] & gt);          // right stops after me
            }
            // ---- phase 2: stop lists (with their pairs) in global rank order, per-position counts, the keys the median move changed ----
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const uint32_t i = (uint32_t)e * bs + tid;
                if ((uint32_t)e < E && i < n) {
                    sPS[i] = PL[e] | (SG[e] << 16);
                    if (isL[e]) sLL[PL[e]] = make_uint2(k[e], v[e]);
                    if (isR[e]) sRR[SG[e]] = make_uint2(k[e], v[e]);
                    if (chg[e]) sKV[i] = make_uint2(k[e], v[e]);
                    if (act[e] && i == f[e]) sCut[i] = 0xFFFFFFFFu;
                }
            }
        }
        __syncthreads();  // #2
        // ---- phase 3: swapped or not (no search), partner's pair, cut candidates ----
        bool sw[EMAX];
#pragma unroll
        for (int e = 0; e < EMAX; ++e) sw[e] = false;
        if (wact) {
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const uint32_t i = (uint32_t)e * bs + tid;
                bool cand = false;
                if (act[e]) {
                    const uint32_t baseL = sPS[f[e] + 1] & 0xFFFFu, baseR = sPS[l[e] - 1] >> 16;
                    const uint32_t A = PL[e] - baseL, G = SG[e] - baseR;
                    const bool swL = isL[e] && G > A, swR = isR[e] && A > G;
                    sw[e] = swL || swR;
                    cand = (isL[e] && !swL) || swR;
                    if (sw[e]) {
                        const uint2 kv = swL ? sRR[baseR + A] : sLL[baseL + G];
                        k[e] = kv.x;
                        v[e] = kv.y;
                    }
                }
                const uint64_t cm = __ballot(cand);
                if (cand) {  // the lowest candidate of my segment inside this wavefront speaks for it
                    const uint32_t wbase = (uint32_t)e * bs + (wave << 6);
                    const uint32_t s0 = f[e] + 1 > wbase ? f[e] + 1 - wbase : 0u;  // first lane of my segment's range in this wavefront
                    const uint64_t range = lt & ~((s0 >= 64u) ? ~0ull : ((1ull << s0) - 1ull));
                    if ((cm & range) == 0ull) atomicMin(&sCut[f[e]], i);
                }
            }
        }
        __syncthreads();  // #3
        // ---- phase 4: swapped keys land, segments split at the cut ----
        if (wact) {
            bool more = false, heap = false;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const uint32_t i = (uint32_t)e * bs + tid;
                if (sw[e]) sKV[i] = make_uint2(k[e], v[e]);
                if (act[e]) {
                    const uint32_t cut = sCut[f[e]];
                    if (i < cut) l[e] = cut;
                    else f[e] = cut;
                    d[e] -= 1;
                    const bool a2 = l[e] - f[e] > (uint32_t)kThreshold;
                    more = more || a2;
                    heap = heap || (a2 && d[e] == 0);
                }
            }
            if (__ballot(more) != 0ull && lane == 0) sTab[64] = 1;
            if (__ballot(heap) != 0ull && lane == 0) sTab[65] = 1;
        }
        __syncthreads();  // #4
    }
    if (tstamp && tid == 0) tstamp[15] = clock64();
    // ---- leaves (<= 16 keys): __final_insertion_sort == a stable sort of every leaf: rank inside [f, l) ----
    // (K2 / V2 may alias the stop lists: every wavefront left the loop after barrier #4, nobody reads them any more)
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const uint32_t i = (uint32_t)e * bs + tid;
        if ((uint32_t)e < E && i < n) {
            const uint32_t ki = k[e], a = f[e], b = l[e];
            uint32_t r = 0;
#pragma unroll
            for (int t = 0; t < kThreshold; ++t) {
                const uint32_t j = a + (uint32_t)t;
                const bool valid = j < b;
                const uint32_t kj = valid ? sKV[j].x : 0u;
                r += (valid && ((kj < ki) || (kj == ki && j < i))) ? 1u : 0u;
            }
            K2[a + r] = ki;
            V2[a + r] = v[e];
        }
    }
    __syncthreads();
    if (tstamp && tid == 0) tstamp[16] = clock64();
}

}  // namespace esort
#endif
