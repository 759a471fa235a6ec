// revert_bins.hip.h -- the two per-bin stages of a reverted bin (v3, erasor.cpp:510-528): R-GPF (extract_ground, erasor.cpp:233-294;
// estimate_plane_ :183-198; extract_initial_seeds_ :204-231) and voxelize_preserving_labels(curr + ground) (utils.cpp:80-114).
// Included by kernels.hip.h inside namespace ek, behind rgpf_after_sort / binvox_core (the global-memory paths of bins beyond LDS).
//
// Round 4: the stages are OUT-OF-LINE PHASES.  One workgroup of 1024 threads owns a bin, so the kernel is capped at 128 VGPRs; as one
// inlined body (round 3) the rare paths, the two exact sorts, the plane fit and the label search shared ONE register allocation --
// 72 spilled VGPRs, scratch reloads inside the plane-fit and label-search loops.  Now every phase is a function of its own
// (__attribute__((noinline)), no arguments: everything it needs lies in file-scope LDS -- the kernel's pointers in g_ra, the bin in g_rb,
// the parameters in g_dp), so each has its own allocation and what is live across a call is saved once per bin, not reloaded per
// iteration.  The level-synchronous exact sort (esort::block_esort_sync) now takes bins of up to 4096 keys (four per thread): the
// one-wavefront-per-segment sort in LDS, which config 4's dense bins used to take, is gone from these kernels.
//
// LDS: ONE pool of 8 * PB_CAP words (128 KB).
//   sort (n <= 2048):  cap = 2048: pairs [0, 2c) | left stops [2c, 4c) | right stops [4c, 6c) | counts [6c, 7c) | cuts [7c, 8c): the lower
//                      half of the pool; (2048, 4096]: cap = 4096, the whole pool.  Sorted keys / values come back in words
//                      [2 * PB_CAP, 3 * PB_CAP) / [3 * PB_CAP, 4 * PB_CAP) ("sL" / "sR") either way.
//   R-GPF fit:         ground list [0, C) | X [C, 2C) | Y [2C, 3C) | Z [3C, 4C) | covariance products: upper half ("big")
//   voxelisation:      unique keys [0, C) | run begins [C, 2C) | sorted keys [2C, 3C) | sorted indices [3C, 4C) | the cloud: upper half
//                      (staged before the sort for the bounding box and the keys; a sort of more than 2048 keys takes the upper half too,
//                      the cloud is then staged again behind it)
#ifndef ERASOR_REVERT_BINS_HIP_H
#define ERASOR_REVERT_BINS_HIP_H

static constexpr uint32_t RG2_CH = 896;        // list elements per set of covariance product rows (two sets: one is added while the other is formed)
static constexpr uint32_t RG_RS = RG2_CH + 4;  // padded row stride of the covariance product rows (floats)
static constexpr uint32_t PB_CAP = 4096;      // points of a bin / of a bin's cloud that take the LDS-resident path
static constexpr uint32_t BV2_LMAX = PB_CAP;
static_assert(RG_LMAX == PB_CAP && 2 * 9 * RG_RS * sizeof(float) <= PB_CAP * sizeof(float4), "the stages share the pool");

__device__ __forceinline__ float key_to_float(uint32_t k) {  // inverse of esort::float_key (-0 comes back as +0)
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct RevArgs {  // the per-bin launch's pointers (kernel arguments, copied to LDS once per workgroup)
    const uint32_t *moff;
    const float4 *spts;
    const uint32_t *qoff;
    const float4 *sq;
    uint32_t *gsK, *gsV, *gsL, *gsR, *gsH, *gsK2, *gsV2;
    float4 *gsC;
    uint8_t *gflag;
    uint32_t *grank, *glist, *ng_arr;
    float *plane_n;
    double *plane_d;
    float4 *vox_out;
    uint32_t *nvox_out;
    Counters *ctr;
    unsigned long long *dbg;
    // the global-memory paths of the two stages index the same scratch arrays, R-GPF by map offsets, the voxelisation by its own: with
    // both stages in flight in different workgroups (fused launch) the voxelisation works vox_base entries (h_base flag words) further up
    uint32_t vox_base, h_base;
};
struct RevBin {
    uint32_t rk, key, vo;  // entry of the reverted list, bin key, the bin's offset in the voxel scratch
    uint32_t state, m, nc;  // voxelisation: what is left to do (BV_*), cloud size, curr points among them
    // round 6: what every phase used to fetch from global memory at its head (a dependent round trip each, ~0.7 us of one workgroup's chain):
    // the bin's range of the bucketed map / scan, fetched ONCE by rev_set_bin; the ground count and whether R-GPF has left the bin's
    // coordinates (pool words [PB_CAP, 4 * PB_CAP)) and ground list (words [0, ng)) in LDS for the voxelisation behind it
    uint32_t o0, M, qo, cc;
    uint32_t ng, in_lds;
    unsigned long long t_a, t_b;
};
enum : uint32_t { BV_DONE = 0, BV_SORT = 1, BV_RARE = 2 };

__shared__ __attribute__((aligned(16))) uint32_t g_rev_pool[8 * PB_CAP];
__shared__ uint32_t g_sync_stab[68];
__shared__ esort::Seg g_qa[PB_CAP / 16 + 2], g_qb[PB_CAP / 16 + 2];  // (the global-memory paths' segment queues)
__shared__ uint32_t g_qcnt[2];
__shared__ uint32_t g_sm[40];
__shared__ uint32_t g_tab[64];
__shared__ uint32_t g_sbb[6];
__shared__ uint32_t g_carry;
__shared__ float g_n[3];
__shared__ double g_th, g_lpr, g_pd;
__shared__ uint32_t g_chg;
__shared__ unsigned long long g_t[12], g_es[24];
__shared__ RevArgs g_ra;
__shared__ RevBin g_rb;
__shared__ DP g_dp;
__shared__ VoxGrid g_vg;

// ---- the level-synchronous exact sort, out of line (round 3): the caller leaves n (key, value) pairs in the pool's first 2 n words ----
// (two functions: the four-keys-per-thread instantiation needs twice the registers of the common one and would push saves / reloads
// into its level loop if they shared an allocation)
__device__ __attribute__((noinline)) void lds_esort_sync_call2(uint32_t n, uint32_t *n_fallback, unsigned long long *tstamp) {  // n <= 2048
    uint32_t *pool = g_rev_pool;
    const uint2 *sKV = reinterpret_cast<const uint2 *>(pool);
    lds_esort_sync_kv<false>(n, [&](uint32_t i) { return sKV[i].x; }, [&](uint32_t i) { return sKV[i].y; }, pool, g_sync_stab, pool + 2 * PB_CAP,
                             pool + 3 * PB_CAP, n_fallback, tstamp, -1, ESYNC_MAX);
}
__device__ __attribute__((noinline)) void lds_esort_sync_call4(uint32_t n, uint32_t *n_fallback, unsigned long long *tstamp) {  // 2048 < n <= 4096
    uint32_t *pool = g_rev_pool;
    const uint2 *sKV = reinterpret_cast<const uint2 *>(pool);
    uint32_t k[4], v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t i = (uint32_t)e * blockDim.x + threadIdx.x;
        k[e] = i < n ? sKV[i].x : 0u;
        v[e] = i < n ? sKV[i].y : 0u;
    }
    esort::block_esort_sync<4>(k, v, n, reinterpret_cast<uint2 *>(pool), reinterpret_cast<uint2 *>(pool + 2 * PB_CAP), reinterpret_cast<uint2 *>(pool + 4 * PB_CAP),
                               pool + 6 * PB_CAP, pool + 7 * PB_CAP, g_sync_stab, pool + 2 * PB_CAP, pool + 3 * PB_CAP, n_fallback, tstamp, -1);
}
__device__ __forceinline__ void lds_esort_sync_call(uint32_t n, uint32_t *n_fallback, unsigned long long *tstamp) {
    if (n <= ESYNC_MAX) lds_esort_sync_call2(n, n_fallback, tstamp);
    else lds_esort_sync_call4(n, n_fallback, tstamp);
}

// ================================================================================================
// R-GPF
// ================================================================================================
// bins beyond the pool: global scratch, same arithmetic (rare)
__device__ __attribute__((noinline)) void rg_rare_call() {
    const DP P = g_dp;
    const RevArgs a = g_ra;
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    const uint32_t rk = g_rb.rk, key = g_rb.key;
    const uint32_t o0 = g_rb.o0, M = g_rb.M;
    const float4 *pts = a.spts + o0;
    uint32_t *K = a.gsK + o0, *V = a.gsV + o0;
    for (uint32_t i = tid; i < M; i += bs) {
        K[i] = esort::float_key(__float_as_uint(pts[i].z));
        V[i] = i;
    }
    __threadfence_block();
    __syncthreads();
    esort::block_esort(K, V, a.gsL + o0, a.gsR + o0, a.gsH + (o0 >> 5) + 2 * key, a.gsK2 + o0, a.gsV2 + o0, 0u, M, 2 * esort::lg2_floor(M), g_qa, g_qb,
                       g_qcnt, (uint32_t)(PB_CAP / 16 + 2), &a.ctr->n_sort_fallback, &a.ctr->sort_qoverflow);
    __threadfence_block();
    __syncthreads();
    rgpf_after_sort(P, pts, M, o0, rk, a.gsV2 + o0, K, g_sm, reinterpret_cast<float *>(g_rev_pool + 4 * PB_CAP), g_n, &g_th, &g_lpr, &g_carry, a.gflag,
                    a.grank, a.glist, a.ng_arr, a.plane_n, a.plane_d, a.ctr);
}

// (1) std::sort(src_copy, point_cmp), erasor.cpp:239-240: (z key, bin-local index) pairs into the pool; the caller runs the sort
__device__ __attribute__((noinline)) void rg_keys_call() {
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    const uint32_t o0 = g_rb.o0, M = g_rb.M;
    const float4 *pts = g_ra.spts + o0;
    if (g_ra.dbg && tid == 0) {
        g_rb.t_a = wall_clock64();
        g_t[0] = g_rb.t_a;
    }
    if (g_ra.dbg && tid < 24) g_es[tid] = 0;
    uint2 *sKV = reinterpret_cast<uint2 *>(g_rev_pool);
    for (uint32_t i = tid; i < M; i += bs) sKV[i] = make_uint2(esort::float_key(__float_as_uint(pts[i].z)), i);
}

// (2) seeds, gf_iter x (plane fit, classification), ground list / flags / ranks out
__device__ __attribute__((noinline)) void rg_fit_call() {
    const DP P = g_dp;
    const uint32_t tid = threadIdx.x, bs = blockDim.x, lane = tid & 63u, wave = tid >> 6, nw = bs >> 6;
    const uint32_t rk = g_rb.rk;
    const uint32_t o0 = g_rb.o0, M = g_rb.M;
    const float4 *pts = g_ra.spts + o0;
    uint8_t *gflag = g_ra.gflag;
    uint32_t *grank = g_ra.grank, *glist_out = g_ra.glist, *ng_out = g_ra.ng_arr;
    float *plane_n = g_ra.plane_n;
    double *plane_d = g_ra.plane_d;
    Counters *ctr = g_ra.ctr;
    unsigned long long *dbg = g_ra.dbg;
    uint32_t *pool = g_rev_pool;
    uint32_t *sK = pool, *sV = pool + PB_CAP, *sL = pool + 2 * PB_CAP, *sR = pool + 3 * PB_CAP;
    float *sProd = reinterpret_cast<float *>(pool + 4 * PB_CAP);
#define RG_STAMP(i) do { if (dbg && tid == 0) g_t[i] = wall_clock64(); } while (0)
    RG_STAMP(1);
    // (round 6: the first row of the bin's points is on its way while the seeds are counted -- the staging below used to start its round trip
    // to global memory behind them)
    const float4 q_first = tid < M ? pts[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    // sorted keys in sL, sorted bin-local indices in sR
    uint32_t drop = 0, ng = 0;
    {
        // --- drop leading z < min_h (erasor.cpp:242-251); monotone in sorted order ---
        uint32_t cnt = 0;
        for (uint32_t k = tid; k < M; k += bs) cnt += ((double)key_to_float(sL[k]) < P.min_h) ? 1u : 0u;
        uint32_t tot;
        block_excl_scan(cnt, g_sm, tot);
        drop = tot;
        const uint32_t Ms = M - drop;
        // --- extract_initial_seeds_ (erasor.cpp:204-231) ---
        if (wave == 0) {  // (round 6: the lowest heights are FETCHED by 64 lanes at a time -- one LDS round trip instead of one per value of the
            // single lane that used to walk them -- and added in the reference's order, in double, from the lanes' registers)
            uint32_t cl = 0;
            if (P.num_lowest >= 0 && Ms > (uint32_t)P.num_lowest && P.gf_lpr > 0) cl = min((uint32_t)P.gf_lpr, Ms - (uint32_t)P.num_lowest);
            double sum = 0;
            for (uint32_t base = 0; base < cl; base += 64u) {
                const uint32_t cn = min(64u, cl - base);
                const uint32_t vb = lane < cn ? __float_as_uint(key_to_float(sL[drop + (uint32_t)P.num_lowest + base + lane])) : 0u;
                for (uint32_t t = 0; t < cn; ++t) sum += (double)__uint_as_float(__builtin_amdgcn_readlane(vb, t));
            }
            if (lane == 0) g_lpr = cl != 0 ? sum / (int)cl : 0;
        }
        __syncthreads();
        const double seed_thr = g_lpr + P.gf_seeds_h;
        cnt = 0;
        for (uint32_t k = tid; k < Ms; k += bs) cnt += ((double)key_to_float(sL[drop + k]) < seed_thr) ? 1u : 0u;
        block_excl_scan(cnt, g_sm, tot);
        ng = tot;  // seeds = the first ng of the sorted points (the predicate is monotone in z)
    }
    if (dbg && tid == 0) g_rb.t_b = wall_clock64();
    // ---- ground list <- seeds; stage the bin's coordinates in LDS ----
    uint32_t *glist = sK;
    for (uint32_t k = tid; k < ng; k += bs) glist[k] = sR[drop + k];
    __syncthreads();  // sV / sL / sR are dead from here on
    float *X = reinterpret_cast<float *>(sV), *Y = reinterpret_cast<float *>(sL), *Z = reinterpret_cast<float *>(sR);
    if (tid < M) {
        X[tid] = q_first.x;
        Y[tid] = q_first.y;
        Z[tid] = q_first.z;
    }
    for (uint32_t i = tid + bs; i < M; i += bs) {
        const float4 q = pts[i];
        X[i] = q.x;
        Y[i] = q.y;
        Z[i] = q.z;
    }
    __syncthreads();
    RG_STAMP(2);
    const uint32_t E = (M + bs - 1) / bs;  // points per thread in the classification (<= 4)
    // Round 6 (second half): a fit whose ground list is the list of the fit before it IS that fit.  From the second iteration on the list is
    // the previous classification's result in source order; when two successive classifications mark the same points (three bins of four on
    // the synthetic sequences: R-GPF has converged after one iteration), the next iteration's nine sums run over the same values in the same
    // order and its decomposition sees the same matrix -- the plane is copied, the sums (2-12 us) and the one-lane SVD (4-5 us) are not run.
    // (The seeds of iteration 0 come in sorted order, not in source order: iteration 1 always fits.)
    uint64_t pbal[4] = {0ull, 0ull, 0ull, 0ull};
    bool same_list = false;
    for (int it = 0; it < P.gf_iter; ++it) {
      if (it >= 2 && same_list) {
        if (tid == 0) {
            if (ng == 0) atomicAdd(&ctr->n_degenerate, 1u);
            plane_n[((size_t)rk * P.gf_iter + it) * 3 + 0] = g_n[0];
            plane_n[((size_t)rk * P.gf_iter + it) * 3 + 1] = g_n[1];
            plane_n[((size_t)rk * P.gf_iter + it) * 3 + 2] = g_n[2];
            plane_d[(size_t)rk * P.gf_iter + it] = g_pd;
            g_chg = 0u;
        }
        __syncthreads();
      } else {
        // --- estimate_plane_: pcl::computeMeanAndCovarianceMatrix, nine float32 accumulators in list order ---
        // The nine products of every list element are formed by ALL threads (parallel, order-free) into LDS rows padded to RG_RS floats
        // (nine lanes, nine banks, 128-bit reads); then lane a of wave 0 adds row a strictly in list order -- the only part that has to be
        // sequential (float32 addition order is what PCL's result depends on).  x*x etc. are single IEEE multiplies either way.
        // Wave 0 adds chunk c while the other waves form the products of chunk c + 1 in the second set of rows.
        float acc = 0.f;
        auto form = [&](uint32_t cb, float *dst, uint32_t t0, uint32_t step) {
            const uint32_t cn = min(RG2_CH, ng - cb);
            for (uint32_t t = t0; t < cn; t += step) {
                const uint32_t gi = glist[cb + t];
                const float x = X[gi], y = Y[gi], z = Z[gi];
                dst[0 * RG_RS + t] = x * x;
                dst[1 * RG_RS + t] = x * y;
                dst[2 * RG_RS + t] = x * z;
                dst[3 * RG_RS + t] = y * y;
                dst[4 * RG_RS + t] = y * z;
                dst[5 * RG_RS + t] = z * z;
                dst[6 * RG_RS + t] = x;
                dst[7 * RG_RS + t] = y;
                dst[8 * RG_RS + t] = z;
            }
        };
        if (ng) form(0, sProd, tid, bs);
        __syncthreads();
        uint32_t par = 0;
        for (uint32_t cb = 0; cb < ng; cb += RG2_CH, par ^= 1u) {
            const uint32_t cn = min(RG2_CH, ng - cb);
            const float *cur = sProd + par * (9 * RG_RS);
            if (wave != 0 || nw == 1) {
                if (cb + RG2_CH < ng) form(cb + RG2_CH, sProd + (par ^ 1u) * (9 * RG_RS), nw == 1 ? tid : tid - 64u, nw == 1 ? bs : bs - 64u);
            }
            if (wave == 0 && lane < 9) {
                // the loads run one group of four 128-bit reads ahead of the adds (the chain of adds is the only latency left);
                // reads past cn stay inside the pool and are not used
                const float4 *row4 = reinterpret_cast<const float4 *>(cur + lane * RG_RS);
                const uint32_t c4 = cn >> 2;
#define RG_ADD4(v) do { acc += (v).x; acc += (v).y; acc += (v).z; acc += (v).w; } while (0)
#define RG_ADD16(p) do { RG_ADD4(p##0); RG_ADD4(p##1); RG_ADD4(p##2); RG_ADD4(p##3); } while (0)
                uint32_t k = 0;
#if defined(__HIP_DEVICE_COMPILE__)
                // Three sets of four 128-bit reads in rotation, each issued 32 adds (~160 cycles) before its use.  The reads and their
                // waits are written out: left to itself the compiler waits for ALL outstanding reads before the first add of a set
                // (lgkmcnt(0) in the loop whatever the order), i.e. one LDS latency per 16 adds.  A wait names the set it releases as
                // in/out operands, so no use can move above it; LDS reads return in order, two younger sets may stay outstanding.
#define RG_READ4(p, byte_off)                                                                                   \
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%5+16\n\t"                      \
                 "ds_read_b128 %2, %4 offset:%5+32\n\tds_read_b128 %3, %4 offset:%5+48"                         \
                 : "=&v"(p##0), "=&v"(p##1), "=&v"(p##2), "=&v"(p##3) : "v"(addr), "n"(byte_off))
#define RG_WAIT4(p, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(p##0), "+v"(p##1), "+v"(p##2), "+v"(p##3))
#define RG_SB() __builtin_amdgcn_sched_barrier(0)  // (keeps the adds between the reads they are meant to cover)
                if (c4 >= 12) {
                    uint32_t addr = (uint32_t)(uintptr_t)(row4);  // (LDS byte address of the row)
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    f32x4 a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
                    RG_READ4(a, 0);
                    RG_READ4(b, 64);
                    for (; k + 12 <= c4; k += 12, addr += 192) {
                        RG_READ4(c, 128);
                        RG_WAIT4(a, 8);
                        RG_SB();
                        RG_ADD16(a);
                        RG_SB();
                        RG_READ4(a, 192);
                        RG_WAIT4(b, 8);
                        RG_SB();
                        RG_ADD16(b);
                        RG_SB();
                        RG_READ4(b, 256);
                        RG_WAIT4(c, 8);
                        RG_SB();
                        RG_ADD16(c);
                        RG_SB();
                    }
                    RG_WAIT4(a, 0);  // (the reads past the last full group: drained, unused)
                    RG_WAIT4(b, 0);
                }
#undef RG_SB
#undef RG_WAIT4
#undef RG_READ4
#endif
                for (; k < c4; ++k) {
                    const float4 v = row4[k];
                    RG_ADD4(v);
                }
#undef RG_ADD16
#undef RG_ADD4
                const float *row = cur + lane * RG_RS;
                for (uint32_t r = c4 << 2; r < cn; ++r) acc += row[r];
            }
            __syncthreads();
        }
        if (it == 0) RG_STAMP(3);
        if (wave == 0) {
            // (lane k divides its own sum by the count: one division instead of nine in the lane that goes on alone)
            const float accn = ng ? acc / (float)ng : acc;
            float a[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) a[k] = __shfl(accn, k, 64);
            if (lane == 0) {
                float cov[9], mean[3], U[9], sv[3];
                if (ng == 0) {
                    for (int k = 0; k < 9; ++k) cov[k] = 0.f;
                    mean[0] = mean[1] = mean[2] = 0.f;
                    atomicAdd(&ctr->n_degenerate, 1u);
                } else {
                    mean[0] = a[6];
                    mean[1] = a[7];
                    mean[2] = a[8];
                    cov[0] = a[0] - a[6] * a[6];
                    cov[1] = a[1] - a[6] * a[7];
                    cov[2] = a[2] - a[6] * a[8];
                    cov[4] = a[3] - a[7] * a[7];
                    cov[5] = a[4] - a[7] * a[8];
                    cov[8] = a[5] - a[8] * a[8];
                    cov[3] = cov[1];
                    cov[6] = cov[2];
                    cov[7] = cov[5];
                }
                jacobi_svd3(cov, U, sv);
                const float n0 = U[2], n1 = U[5], n2_ = U[8];
                const float dot = (n0 * mean[0] + n1 * mean[1]) + n2_ * mean[2];
                const double d = -dot;
                g_n[0] = n0;
                g_n[1] = n1;
                g_n[2] = n2_;
                g_th = P.gf_dist - d;
                g_pd = d;
                g_chg = 0u;
                plane_n[((size_t)rk * P.gf_iter + it) * 3 + 0] = n0;
                plane_n[((size_t)rk * P.gf_iter + it) * 3 + 1] = n1;
                plane_n[((size_t)rk * P.gf_iter + it) * 3 + 2] = n2_;
                plane_d[(size_t)rk * P.gf_iter + it] = d;
            }
        }
        __syncthreads();
        if (it == 0) RG_STAMP(4);
      }
        // --- points * normal_ < th_dist_d_ in source order (erasor.cpp:265-281) ---
        const float n0 = g_n[0], n1 = g_n[1], n2_ = g_n[2];
        const double th = g_th;
        const bool last = it == P.gf_iter - 1;
        // point i = e * bs + tid; per (e, wave) ground counts -> one 64-entry table scan gives every wave its offset
        uint64_t bal[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            bool g = false;
            const uint32_t i = e * bs + tid;
            if (e < E && i < M) {
                const float res = (X[i] * n0 + Y[i] * n1) + Z[i] * n2_;
                g = (double)res < th;
            }
            bal[e] = __ballot(g);
            if (lane == 0) g_tab[e * 16 + wave] = (e < E && wave < nw) ? (uint32_t)__popcll(bal[e]) : 0u;
        }
        {   // (does this classification mark the points the previous one marked?  g_chg was cleared behind the fit, two barriers ago)
            bool chg = false;
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                chg = chg || bal[e] != pbal[e];
                pbal[e] = bal[e];
            }
            if (chg && lane == 0) g_chg = 1u;
        }
        __syncthreads();
        same_list = it >= 1 && g_chg == 0u;
        if (wave == 0) {  // exclusive scan of the 64 (e-major, wave-minor) counts
            const uint32_t v = g_tab[lane];
            const uint32_t inc = esort::wave_incl_scan(v);  // (DPP row shifts)
            g_tab[lane] = inc - v;
            if (lane == 63) g_carry = inc;
        }
        __syncthreads();
        const uint64_t lt = lanemask_lt();
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t i = e * bs + tid;
            if (e < E && i < M) {
                const bool g = (bal[e] >> lane) & 1ull;
                const uint32_t gr = g_tab[e * 16 + wave] + (uint32_t)__popcll(bal[e] & lt);  // rank among the ground points
                if (g) glist[gr] = i;
                if (last) {
                    gflag[o0 + i] = g ? 1 : 0;
                    grank[o0 + i] = g ? gr : (i - gr);  // rank among ground / among rejected
                }
            }
        }
        ng = g_carry;
        __syncthreads();
        if (it == 0) RG_STAMP(5);
    }
    for (uint32_t k = tid; k < ng; k += bs) glist_out[o0 + k] = glist[k];
    if (tid == 0) {
        ng_out[rk] = ng;
        g_rb.ng = ng;  // (the voxelisation behind this call takes the count, the list and the coordinates from LDS)
        g_rb.in_lds = 1u;
    }
    if (dbg && tid == 0) {  // diagnostics (ERASOR_HIP_SORT_STAMPS): the slowest bin's split between the z-sort and the rest, 10 ns ticks
        const unsigned long long t_c = wall_clock64(), t_a = g_rb.t_a, t_b = g_rb.t_b;
        if (atomicMax(&dbg[16], t_c - t_a) < t_c - t_a) {
            dbg[17] = t_b - t_a;
            dbg[18] = t_c - t_b;
            dbg[19] = M;
            dbg[32] = 0;                // key load
            dbg[33] = g_t[1] - g_t[0];  // key load + exact sort
            dbg[34] = t_b - g_t[1];     // seeds
            dbg[35] = g_t[2] - t_b;     // staging
            dbg[36] = g_t[3] - g_t[2];  // covariance sums (iteration 0)
            dbg[37] = g_t[4] - g_t[3];  // SVD
            dbg[38] = g_t[5] - g_t[4];  // classification
            dbg[39] = ng;
            for (int i = 0; i < 24; ++i) dbg[40 + i] = g_es[i];
        }
    }
#undef RG_STAMP
}

// R-GPF of the bin in g_rb (workgroup-uniform control flow; ends with the bin's ground list / count in global memory)
__device__ __forceinline__ void rgpf_stage() {
    const uint32_t M = g_rb.M;
    if (g_ra.dbg && threadIdx.x == 0) {  // diagnostics: reverted bins, those beyond the pool, points
        atomicAdd(&g_ra.dbg[23], 1ull + (M > PB_CAP ? (1ull << 32) : 0ull));
        atomicAdd(&g_ra.dbg[63], (unsigned long long)M);
    }
    if (M > PB_CAP) {
        rg_rare_call();
    } else {
        rg_keys_call();
        lds_esort_sync_call(M, &g_ra.ctr->n_sort_fallback, g_ra.dbg ? g_es : nullptr);
        rg_fit_call();
    }
}

// ================================================================================================
// per-bin voxelize_preserving_labels(curr points + reverted ground, /erasor/map_voxel_size) -- erasor.cpp:523-528
// ================================================================================================
__device__ __forceinline__ float4 bv_input(uint32_t j, uint32_t nc, const float4 *__restrict__ sqb, const float4 *__restrict__ sptb,
                                           const uint32_t *__restrict__ glb) {
    // input cloud: curr bin points (scan order) then the reverted ground (source order)
    return j < nc ? sqb[j] : sptb[glb[j - nc]];
}

__device__ __attribute__((noinline)) void bv_rare_call() {  // clouds beyond the pool: global scratch, brute-force search (binvox_core)
    const DP P = g_dp;
    const RevArgs a = g_ra;
    const uint32_t rk = g_rb.rk, vo = g_rb.vo, vb = a.vox_base;
    const uint32_t mo = g_rb.o0, qo = g_rb.qo;
    const uint32_t nc = g_rb.cc, m = nc + a.ng_arr[rk];
    binvox_core(P, m, nc, a.sq + qo, a.spts + mo, a.glist + mo, a.gsK + vb + vo, a.gsV + vb + vo, a.gsK2 + vb + vo, a.gsV2 + vb + vo, a.gsC + vo,
                a.gsL + vb + vo, a.gsR + vb + vo, a.gsH + a.h_base + (vo >> 5) + 2 * rk, g_qa, g_qb, g_qcnt, g_sm, g_sbb, &g_carry, a.vox_out + vo,
                a.nvox_out + rk, a.ctr);
}

// (1) the cloud, its bounding box, VoxelGrid geometry, keys into the pool -- or one of the short ends (empty bin, index overflow)
__device__ __attribute__((noinline)) void bv_keys_call() {
    const DP P = g_dp;
    const uint32_t tid = threadIdx.x, bs = blockDim.x, lane = tid & 63u;
    const uint32_t rk = g_rb.rk, vo = g_rb.vo;
    const uint32_t mo = g_rb.o0, qo = g_rb.qo;
    const uint32_t in_lds = g_rb.in_lds;
    const uint32_t nc = g_rb.cc, ngr = in_lds ? g_rb.ng : g_ra.ng_arr[rk];
    const uint32_t m = nc + ngr;
    uint32_t *nvox_out = g_ra.nvox_out;
    if (tid == 0) {
        g_rb.m = m;
        g_rb.nc = nc;
        g_rb.state = BV_DONE;
        if (g_ra.dbg) g_rb.t_a = wall_clock64();
    }
    if (nc == 0) {  // selected = bin_curr with is_occupied == false: r_pod2pc skips the bin
        if (tid == 0) nvox_out[rk] = 0;
        return;
    }
    if (m > PB_CAP) {
        if (tid == 0) g_rb.state = BV_RARE;
        return;
    }
    const float4 *sqb = g_ra.sq + qo, *sptb = g_ra.spts + mo;
    const uint32_t *glb = g_ra.glist + mo;
    float4 *vout = g_ra.vox_out + vo;
    float4 *sC = reinterpret_cast<float4 *>(g_rev_pool + 4 * PB_CAP);
    if (in_lds) {
        // round 6: the reverted ground comes from where R-GPF left it -- list and coordinates in LDS, only the label from global memory (one
        // round trip, not the list's and then the points')
        const uint32_t *gl = g_rev_pool;
        const float *X = reinterpret_cast<const float *>(g_rev_pool + PB_CAP), *Y = reinterpret_cast<const float *>(g_rev_pool + 2 * PB_CAP),
                    *Z = reinterpret_cast<const float *>(g_rev_pool + 3 * PB_CAP);
        for (uint32_t j = tid; j < m; j += bs) {
            if (j < nc) {
                sC[j] = sqb[j];
            } else {
                const uint32_t gi = gl[j - nc];
                sC[j] = make_float4(X[gi], Y[gi], Z[gi], sptb[gi].w);
            }
        }
    } else {
        for (uint32_t j = tid; j < m; j += bs) sC[j] = bv_input(j, nc, sqb, sptb, glb);
    }
    if (tid < 3) g_sbb[tid] = 0xFFFFFFFFu;
    if (tid >= 3 && tid < 6) g_sbb[tid] = 0u;
    __syncthreads();
    {
        uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
        for (uint32_t j = tid; j < m; j += bs) {
            const float4 p = sC[j];
            const uint32_t k3[3] = {fkey_ord(p.x), fkey_ord(p.y), fkey_ord(p.z)};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                mn[a] = k3[a] < mn[a] ? k3[a] : mn[a];
                mx[a] = k3[a] > mx[a] ? k3[a] : mx[a];
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = wave_minmax_u<false>(mn[a]);
            mx[a] = wave_minmax_u<true>(mx[a]);
            if (lane == 0) {
                atomicMin(&g_sbb[a], mn[a]);
                atomicMax(&g_sbb[3 + a], mx[a]);
            }
        }
    }
    __syncthreads();
    const float mnf[3] = {fkey_inv(g_sbb[0]), fkey_inv(g_sbb[1]), fkey_inv(g_sbb[2])};
    const float mxf[3] = {fkey_inv(g_sbb[3]), fkey_inv(g_sbb[4]), fkey_inv(g_sbb[5])};
    const VoxGrid g = vox_grid_from_bbox(mnf, mxf, P.leaf_map);
    if (g.overflow) {  // VoxelGrid returns the input unchanged (utils.cpp:88-91); the label search then finds the point itself or
        // its first exact duplicate (distance 0, lowest index)
        for (uint32_t j = tid; j < m; j += bs) {
            const float4 p = sC[j];
            float label = p.w;
            for (uint32_t i = 0; i < j; ++i) {
                const float4 q = sC[i];
                if (q.x == p.x && q.y == p.y && q.z == p.z) {
                    label = q.w;
                    break;
                }
            }
            vout[j] = make_float4(p.x, p.y, p.z, label);
        }
        if (tid == 0) {
            atomicAdd(&g_ra.ctr->n_voxel_overflow, 1u);
            nvox_out[rk] = m;
        }
        return;
    }
    if (tid == 0) {
        g_vg = g;
        g_rb.state = BV_SORT;
    }
    uint2 *sKV = reinterpret_cast<uint2 *>(g_rev_pool);
    for (uint32_t j = tid; j < m; j += bs) {
        const float4 p = sC[j];
        sKV[j] = make_uint2(vox_index(g, p.x, p.y, p.z), j);
    }
}

// (2) behind the exact sort: run heads, then per voxel the CentroidPoint float sums in sorted order and the exact 1-NN label
// (utils.cpp:94-112) by walking the voxel grid: own cell, then shells of neighbour cells found by binary search in the sorted unique
// keys, pruned by conservative cell bounds -- the same minimum over (distance, index) pairs as every centroid x every point
__device__ __attribute__((noinline)) void bv_label_call() {
    const uint32_t tid = threadIdx.x, bs = blockDim.x, lane = tid & 63u, wave = tid >> 6, nw = bs >> 6;
    const uint32_t rk = g_rb.rk, m = g_rb.m;
    const VoxGrid g = g_vg;
    uint32_t *pool = g_rev_pool;
    uint32_t *sK = pool, *sV = pool + PB_CAP, *sL = pool + 2 * PB_CAP, *sR = pool + 3 * PB_CAP;
    float4 *sC = reinterpret_cast<float4 *>(pool + 4 * PB_CAP);
    float4 *vout = g_ra.vox_out + g_rb.vo;
    if (m > ESYNC_MAX) {  // the sort of more than 2048 keys took the upper half of the pool: the cloud again
        const uint32_t nc = g_rb.nc;
        const float4 *sqb = g_ra.sq + g_rb.qo, *sptb = g_ra.spts + g_rb.o0;
        const uint32_t *glb = g_ra.glist + g_rb.o0;
        for (uint32_t j = tid; j < m; j += bs) sC[j] = bv_input(j, nc, sqb, sptb, glb);
    }
    // ---- run heads: unique keys (ascending) -> sK, run begins -> sV ----
    const uint32_t E = (m + bs - 1) / bs;
    uint64_t bal[4];
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
        const uint32_t i = e * bs + tid;
        const bool hd = e < E && i < m && (i == 0 || sL[i] != sL[i - 1]);
        bal[e] = __ballot(hd);
        if (lane == 0) g_tab[e * 16 + wave] = (wave < nw) ? (uint32_t)__popcll(bal[e]) : 0u;
    }
    __syncthreads();
    if (wave == 0) {
        const uint32_t v = g_tab[lane];
        const uint32_t inc = esort::wave_incl_scan(v);  // (DPP row shifts)
        g_tab[lane] = inc - v;
        if (lane == 63) g_carry = inc;
    }
    __syncthreads();
    {
        const uint64_t lt = lanemask_lt();
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t i = e * bs + tid;
            if ((bal[e] >> lane) & 1ull) {
                const uint32_t v = g_tab[e * 16 + wave] + (uint32_t)__popcll(bal[e] & lt);
                sK[v] = sL[i];
                sV[v] = i;
            }
        }
    }
    const uint32_t nv = g_carry;
    __syncthreads();
    const int dx = g.div_b[0], dy = g.div_b[1], dz = g.div_b[2];
    const double L = 1.0 / (double)g.inv_leaf;
    const int maxrho = max(dx, max(dy, dz));
    for (uint32_t v0 = 0; v0 < nv; v0 += bs / NN_SUB) {
        const uint32_t v = v0 + tid / NN_SUB, sub = tid & (NN_SUB - 1);
        if (v >= nv) continue;  // (the eight lanes of a voxel take the same branch; no barrier inside this loop)
        const uint32_t rs = sV[v], re = (v + 1 < nv) ? sV[v + 1] : m;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        uint32_t li = rs;
        for (; li + 3 < re; li += 4) {  // four points' two dependent LDS reads in flight at a time; the additions keep their order
            const uint32_t i0 = sR[li], i1 = sR[li + 1], i2 = sR[li + 2], i3 = sR[li + 3];
            const float4 p0 = sC[i0], p1 = sC[i1], p2 = sC[i2], p3 = sC[i3];
            sx += p0.x; sy += p0.y; sz += p0.z;
            sx += p1.x; sy += p1.y; sz += p1.z;
            sx += p2.x; sy += p2.y; sz += p2.z;
            sx += p3.x; sy += p3.y; sz += p3.z;
        }
        for (; li < re; ++li) {  // (the averaged intensity is overwritten by the nearest input point's label)
            const float4 p = sC[sR[li]];
            sx += p.x;
            sy += p.y;
            sz += p.z;
        }
        const float fc = (float)(re - rs);
        const float cx = sx / fc, cy = sy / fc, cz = sz / fc;
        const uint32_t vkey = sK[v];
        const int ci = (int)(vkey % (uint32_t)dx), cj = (int)((vkey / (uint32_t)dx) % (uint32_t)dy), ck = (int)(vkey / ((uint32_t)dx * (uint32_t)dy));
        const double cc[3] = {(double)cx, (double)cy, (double)cz};
        const int cidx[3] = {ci, cj, ck};
        float best = __int_as_float(0x7F800000);
        uint32_t best_i = 0xFFFFFFFFu;
        for (uint32_t li2 = rs + sub; li2 < re; li2 += NN_SUB) {  // stage 0: the voxel's own points
            const uint32_t pi = sR[li2];
            const float4 p = sC[pi];
            nn_take(l2_simple(cx, cy, cz, p.x, p.y, p.z), pi, best, best_i);
        }
        nn_merge(best, best_i);
        bool done = false;
        {
            double gmin = 1e300;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double lo = (double)(g.min_b[a] + cidx[a]) * L;
                const double hi = (double)(g.min_b[a] + cidx[a] + 1) * L;
                const double margin = 1e-3 * L + 1e-6 * fabs(cc[a]);
                gmin = fmin(gmin, fmin(cc[a] - lo, hi - cc[a]) - margin);
            }
            done = (gmin > 0.0 && (double)best <= gmin * gmin);
        }
        // one cell of a shell: pruned by its squared distance d2c from the centroid, found by binary search in the ascending
        // unique keys, its points taken
        auto visit = [&](int ii, int jj, int kk, double d2c, float shell_best) {
            if (d2c > (double)shell_best) return;
            const uint32_t q = (uint32_t)ii + (uint32_t)jj * (uint32_t)dx + (uint32_t)kk * (uint32_t)dx * (uint32_t)dy;
            uint32_t lo = 0, hi = nv;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (sK[mid] < q) lo = mid + 1;
                else hi = mid;
            }
            if (lo >= nv || sK[lo] != q) return;  // empty cell
            const uint32_t ws = sV[lo], we = (lo + 1 < nv) ? sV[lo + 1] : m;
            for (uint32_t lj = ws; lj < we; ++lj) {
                const uint32_t pi = sR[lj];
                const float4 p = sC[pi];
                nn_take(l2_simple(cx, cy, cz, p.x, p.y, p.z), pi, best, best_i);
            }
        };
        auto slab_d2 = [&](int a, int cell_a) -> double {
            const double lo_a = (double)(g.min_b[a] + cell_a) * L, hi_a = (double)(g.min_b[a] + cell_a + 1) * L;
            const double margin = 1e-3 * L + 1e-6 * fabs(cc[a]);
            const double da = fmax(0.0, fmax(lo_a - cc[a], cc[a] - hi_a) - margin);
            return da * da;
        };
        for (int rho = 1; !done; ++rho) {
            const float shell_best = best;
            if (rho == 1) {  // (the first shell by cell number, slab distances from a 3 x 3 table: see k_query_nn)
                double tab[3][3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int o = 0; o < 3; ++o) tab[a][o] = slab_d2(a, cidx[a] + o - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = (int)sub + 8 * r;
                    const int oi = n % 3, oj = (n / 3) % 3, ok = n / 9;
                    const int ii = ci + oi - 1, jj = cj + oj - 1, kk = ck + ok - 1;
                    if (n >= 27 || n == 13 || ii < 0 || ii >= dx || jj < 0 || jj >= dy || kk < 0 || kk >= dz) continue;
                    double d2c = 0.0;
                    d2c += oi == 0 ? tab[0][0] : (oi == 1 ? tab[0][1] : tab[0][2]);
                    d2c += oj == 0 ? tab[1][0] : (oj == 1 ? tab[1][1] : tab[1][2]);
                    d2c += ok == 0 ? tab[2][0] : (ok == 1 ? tab[2][1] : tab[2][2]);
                    visit(ii, jj, kk, d2c, shell_best);
                }
            } else {
                uint32_t turn = 0;
                for (int kk = ck - rho; kk <= ck + rho; ++kk) {
                    if (kk < 0 || kk >= dz) continue;
                    for (int jj = cj - rho; jj <= cj + rho; ++jj) {
                        if (jj < 0 || jj >= dy) continue;
                        const bool shell_jk = (abs(jj - cj) == rho) || (abs(kk - ck) == rho);
                        for (int ii = ci - rho; ii <= ci + rho; ++ii) {
                            if (ii < 0 || ii >= dx) continue;
                            if (!shell_jk && abs(ii - ci) < rho) continue;
                            if ((turn++ & (NN_SUB - 1)) != sub) continue;
                            double d2c = 0.0;
                            d2c += slab_d2(0, ii);
                            d2c += slab_d2(1, jj);
                            d2c += slab_d2(2, kk);
                            visit(ii, jj, kk, d2c, shell_best);
                        }
                    }
                }
            }
            nn_merge(best, best_i);
            if (rho >= maxrho) break;
            double gmin = 1e300;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double lo = (double)(g.min_b[a] + cidx[a] - rho) * L;
                const double hi = (double)(g.min_b[a] + cidx[a] + rho + 1) * L;
                const double margin = 1e-3 * L + 1e-6 * fabs(cc[a]);
                gmin = fmin(gmin, fmin(cc[a] - lo, hi - cc[a]) - margin);
            }
            if (best_i != 0xFFFFFFFFu && gmin > 0.0 && (double)best <= gmin * gmin) break;
        }
        if (sub == 0) vout[v] = make_float4(cx, cy, cz, sC[best_i < m ? best_i : 0u].w);
    }
    if (tid == 0) g_ra.nvox_out[rk] = nv;
    if (g_ra.dbg && tid == 0) {
        const unsigned long long t_c = wall_clock64(), t_a = g_rb.t_a;
        if (atomicMax(&g_ra.dbg[20], t_c - t_a) < t_c - t_a) {
            g_ra.dbg[21] = m;
            g_ra.dbg[22] = nv;
        }
    }
}

// the voxelisation of the bin in g_rb (its ground list and count are in global memory, written by this workgroup or by an earlier launch)
__device__ __forceinline__ void binvox_stage() {
    bv_keys_call();
    __syncthreads();
    const uint32_t state = g_rb.state;
    if (state == BV_SORT) {
        lds_esort_sync_call(g_rb.m, &g_ra.ctr->n_sort_fallback, nullptr);
        __syncthreads();
        bv_label_call();
    } else if (state == BV_RARE) {
        bv_rare_call();
    }
}

// ================================================================================================
// kernels
// ================================================================================================
__device__ __forceinline__ void rev_open(const DP &P, const RevArgs &ra) {  // once per workgroup
    if (threadIdx.x == 0) {
        g_dp = P;
        g_ra = ra;
    }
    __syncthreads();
}
static constexpr uint32_t VO_BY_POSITION = 0xFFFFFFFFu;  // rev_set_bin: the bin's share of the voxel scratch begins at moff[key] + qoff[key]
__device__ __forceinline__ void rev_set_bin(uint32_t rk, uint32_t key, uint32_t vo) {
    __syncthreads();  // the previous bin's LDS (and g_rb) is dead
    if (threadIdx.x == 0) {
        uint32_t o0, M, q0, cc;
        if (vo == VO_BY_POSITION) {  // (the fused launch: rev_select_call has fetched the ranges beside its scan)
            o0 = g_sel[4];
            M = g_sel[5];
            q0 = g_sel[6];
            cc = g_sel[7];
        } else {
            const uint32_t o1 = g_ra.moff[key + 1], q1 = g_ra.qoff[key + 1];  // (four loads in flight together)
            o0 = g_ra.moff[key];
            q0 = g_ra.qoff[key];
            M = o1 - o0;
            cc = q1 - q0;
        }
        g_rb.rk = rk;
        g_rb.key = key;
        g_rb.vo = vo == VO_BY_POSITION ? o0 + q0 : vo;
        g_rb.o0 = o0;
        g_rb.M = M;
        g_rb.qo = q0;
        g_rb.cc = cc;
        g_rb.ng = 0u;
        g_rb.in_lds = 0u;
    }
    __syncthreads();
}

// R-GPF alone over the reverted-bin LIST (v2; ERASOR_HIP_NO_FUSE=1 / --profile-all: v3 in two launches)
__global__ __launch_bounds__(1024) void k_rgpf2(DP P, const uint32_t *__restrict__ rev_list, const DevState *__restrict__ st, RevArgs ra) {
    rev_open(P, ra);
    const uint32_t n_rev = st->n_rev;
    for (uint32_t rk = blockIdx.x; rk < n_rev; rk += gridDim.x) {
        rev_set_bin(rk, rev_list[rk], 0u);
        rgpf_stage();
    }
}
// the voxelisation alone over the list (the ground lists come from k_rgpf2)
__global__ __launch_bounds__(1024) void k_binvox2(DP P, const uint32_t *__restrict__ rev_list, const DevState *__restrict__ st,
                                                  const uint32_t *__restrict__ vox_off, RevArgs ra) {
    rev_open(P, ra);
    const uint32_t n_rev = st->n_rev;
    for (uint32_t rk = blockIdx.x; rk < n_rev; rk += gridDim.x) {
        rev_set_bin(rk, rev_list[rk], vox_off[rk]);
        binvox_stage();
    }
}
// v3's two per-bin stages in ONE launch (erasor.cpp:521-528: extract_ground, then voxelize_preserving_labels of curr + ground): a
// workgroup runs R-GPF on its bin and voxelises it right away -- the ground list it has just written is its own, so no kernel boundary
// is needed in between, and a small bin is through both stages while the largest one is still fitting planes.
__global__ __launch_bounds__(1024) void k_revert_bins(DP P, const uint32_t *__restrict__ rev_list, const DevState *__restrict__ st,
                                                      const uint32_t *__restrict__ vox_off, RevArgs ra) {
    rev_open(P, ra);
    const uint32_t n_rev = st->n_rev;
    for (uint32_t rk = blockIdx.x; rk < n_rev; rk += gridDim.x) {
        rev_set_bin(rk, rev_list[rk], vox_off[rk]);
        rgpf_stage();
        __threadfence_block();
        __syncthreads();  // the bin's ground list and count (global memory, written by this workgroup) are read below
        binvox_stage();
    }
}

// Round 4: the same launch WITHOUT waiting for k_srt4 -- k_srt4 IS the launch's last workgroup.  Workgroups 0 .. gridDim.x - 2 find their
// bins themselves (rev_select_call: the v3 revert decision is local to a bin), the last one computes status / action / reverted list /
// layout prefixes for the write-back beside them.  Same results, same scratch indexing (entry numbers in key order either way); the Scan
// Ratio Test's second pass (14 us on one compute unit + a kernel boundary) is off the dependency chain.
struct SrtArgs {
    const uint32_t *mcnt;
    const float *mmin, *mmax;
    const uint32_t *ccnt;
    const float *cmin, *cmax;
    uint8_t *st1, *status, *action;
    uint32_t *rev_idx, *rev_list, *vox_off;
    DevState *st;
    uint32_t *out_off0, *rev_before, *crej_off;
    const uint8_t *st1_in;
    const uint32_t *moff, *qoff;
};
__shared__ SrtArgs g_srt_args;
// k_srt4's work as an out-of-line call of the fused launch's last workgroup (arguments through LDS, its tables in the per-bin pool, which
// that workgroup does not use): nothing of it shares registers with the per-bin stages
__device__ __attribute__((noinline)) void srt4_call() {
    const SrtArgs a = g_srt_args;
    const DP P = g_dp;
    uint32_t *sm = g_rev_pool;                                        // 96 words (block_excl_scan_n<6>)
    uint8_t *s_st1 = reinterpret_cast<uint8_t *>(g_rev_pool + 128);  // 1024 * SRT_KPT bytes
    srt4_body(P, sm, s_st1, a.mcnt, a.mmin, a.mmax, a.ccnt, a.cmin, a.cmax, a.st1, a.status, a.action, a.rev_idx, a.rev_list, a.vox_off, a.st, a.out_off0,
              a.rev_before, a.crej_off, a.st1_in, a.moff, a.qoff);
}
// (round 5: sa.status == nullptr -- no extra workgroup: k_srt4's work is done on the stream that writes the map back EARLY -- round 6: as the
// first workgroup of k_assemble_early)
__global__ __launch_bounds__(1024) void k_revert_bins_srt(DP P, SrtArgs sa, RevArgs ra) {
    const bool has_srt = sa.status != nullptr;
    const uint32_t stride = gridDim.x - (has_srt ? 1u : 0u);
    if (has_srt && blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x == 0) {
            g_srt_args = sa;
            g_dp = P;
        }
        __syncthreads();
        srt4_call();
        return;
    }
    const unsigned long long w0 = ra.dbg ? wall_clock64() : 0ull;
    CHAIN_STAMP(2);
    rev_open(P, ra);
    for (uint32_t rk = blockIdx.x;; rk += stride) {
        __syncthreads();  // (g_sel of the previous round has been read)
        const unsigned long long w1 = ra.dbg ? wall_clock64() : 0ull;
        rev_select_call(P.B, sa.st1_in, rk, ra.moff, ra.qoff);
        const uint32_t key = g_sel[0], n_rev = g_sel[2];
        if (rk >= n_rev) break;
        rev_set_bin(rk, key, VO_BY_POSITION);
        const unsigned long long w2 = ra.dbg ? wall_clock64() : 0ull;
        rgpf_stage();
        __threadfence_block();
        __syncthreads();  // the bin's ground list and count (global memory, written by this workgroup) are read below
        const unsigned long long w3 = ra.dbg ? wall_clock64() : 0ull;
        binvox_stage();
        if (ra.dbg && threadIdx.x == 0) {  // diagnostics: the workgroup that finishes last, 10 ns ticks since its start
            const unsigned long long w4 = wall_clock64();
            if (atomicMax(&ra.dbg[64], w4 - w0) < w4 - w0) {
                ra.dbg[65] = w1 - w0;
                ra.dbg[66] = w2 - w1;
                ra.dbg[67] = w3 - w2;
                ra.dbg[68] = w4 - w3;
                ra.dbg[69] = g_rb.M;
                ra.dbg[70] = g_rb.m;
            }
        }
    }
}

#endif  // ERASOR_REVERT_BINS_HIP_H
