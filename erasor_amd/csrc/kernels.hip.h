// kernels.hip.h — hand-written gfx950 kernels of the ERASOR hot path.
//
// Reference citations (paths under the reference checkout):
//   erasor.cpp = src/offline_map_updater/src/erasor.cpp, erasor.h = include/erasor/erasor.h,
//   OMU.cpp = src/offline_map_updater/src/OfflineMapUpdater.cpp, utils.cpp = .../erasor_utils.cpp
//
// Arithmetic contract (compiled with -ffp-contract=off, no fast-math): every float/double operation
// below is a separate IEEE operation in the reference's order, so results are bit-identical to the
// reference's x86-64 SSE2 build.  The only libm call whose last bit may differ is atan2 (OCML vs
// glibc); points where that could change a sector index are counted in Counters::n_ambiguous.
#ifndef ERASOR_KERNELS_HIP_H
#define ERASOR_KERNELS_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exact_sort.hip.h"

namespace ek {

static constexpr uint32_t HOLE_BITS = 0xFFC0DEADu;  // x of a tombstoned outskirts slot (a quiet NaN payload)
static constexpr int TILE = 64;
static constexpr int CHUNK_TILES = 16;
#ifndef ERASOR_GATHER_SUB
#define ERASOR_GATHER_SUB 4
#endif
static constexpr uint32_t GATHER_SUB = ERASOR_GATHER_SUB;  // pieces per VoI-resident chunk in k_voi_gather (1, 2, 4, 8 or 16)
static constexpr int CHUNK = TILE * CHUNK_TILES;  // 1024 points handled by one wavefront iteration (16 loads in flight per lane)
static constexpr double INF_H = 10000000000000.0;  // erasor.h:3
static constexpr double PI_REF = 3.1415926535;     // erasor.h:4

// status codes on device
enum : uint8_t { ST_LITTLE = 0, ST_MERGE = 1, ST_MAP = 2, ST_BLOCKED = 3, ST_CURR = 4, ST_NOT_ASSIGNED = 5 };

struct Xf {
    float m[12];
};

// diagnostics (ERASOR_HIP_CHAIN_STAMPS=1): the 100 MHz counter when the first thread of a launch starts, by launch (see erasor_hip.hip:
// print_chain_stamps) -- the untraced timeline of a step's launches over the streams; off: one uniform load per launch
__device__ unsigned long long g_stamps[32];
__device__ int g_stamps_on;
#define CHAIN_STAMP(i)                                                                    \
    do {                                                                                  \
        if (g_stamps_on && blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); \
    } while (0)

struct DP {  // device copy of the parameters
    double max_r, ring_size, sector_size, max_h, min_h, th_bin_max_h, srt_thr, gf_dist, gf_seeds_h, voi_r2;
    int32_t R, S, B, num_lowest, min_pts, gf_iter, gf_lpr, version;
    float leaf_map, leaf_query;
};

struct Counters {
    uint32_t n_neg_sector, n_ambiguous, n_degenerate, n_voxel_overflow, n_sort_fallback, sort_qoverflow, err, pad;
};

// device-resident step state (copied to the host twice per step)
struct DevState {
    // map store
    uint32_t nF, o_begin;
    // VoI split products
    uint32_t voi_total, voiF, valid_total, validF, n_leaving, o_new_begin;
    // query
    uint32_t q_nvox, q_overflow, q_nbinned, n_o_read;  // n_o_read: outskirts chunks the VoI split had to READ this step (the others: skipped by bounding box)
    int32_t q_min_b[3], q_div_b[3];
    // SRT / R-GPF
    uint32_t n_rev, vox_scratch_total;
    // assembly
    uint32_t total_bins, n_static_est, n_ground, n_compl, n_rejected, nF_new, n_curr_rejected;
    uint32_t total_bins0;  // k_srt4: size of the bin part of the output WITHOUT the reverted bins (k_assemble_map<., true> adds them)
    // label counters
    unsigned long long F_static, F_dynamic, O_static, O_dynamic;
    // measurement: the constant-rate counter (100 MHz) when the step's chunk scan began
    unsigned long long t_open;
    // round 5: the VoI-resident region may hold HOLES (x == HOLE_BITS, like a tombstoned outskirts slot): a step whose write-back does
    // not wait for its per-bin launch reserves every reverted bin's place at full size (RESERVED layout, see k_srt4 / k_assemble_late)
    // and what R-GPF / the voxelisation do not fill stays empty.  nF / nF_new are EXTENTS; these count the points.
    uint32_t nF_valid, nF_valid_new;
    uint32_t total_binsR, ground_res;  // reserved layout: extent of the bin part, of the ground_viz part
    uint32_t n_late;                   // entries of the late table the step leaves (LateEnt: two per reverted bin)
    uint32_t n_voi_dead;               // VoI-order slots kept for the previous step's late points that turned out to hold none (overlapped steps)
    uint32_t n_late_leaving;           // ... of them points that left the VoI (counted into n_leaving)
    uint32_t ov_bad;                   // the passes launched ahead of this step found no room (front of the outskirts / VoI-order arrays): they did nothing
};

// A reserved range of the VoI-resident region (round 5).  [start, start + ndata): data slots -- the first `actual` hold points once the
// per-bin launch and k_assemble_late are through, the rest are holes; [start + ndata, start + ntotal): leaving reservations -- never
// points: they give the late points of a bin that may leave the next VoI their places in the outskirts' order (phantoms).
struct LateEnt {
    uint32_t start, ndata, ntotal, actual;
};

__device__ __forceinline__ uint64_t lanemask_lt() { return esort::lanemask_lt(); }

// pcl::transformPointCloud, PCL <= 1.9 scalar formula (OMU.cpp:240,436,447): ((a*x + b*y) + c*z) + d
__device__ __forceinline__ float4 xform(const Xf &T, float4 p) {
    float4 o;
    o.x = ((T.m[0] * p.x + T.m[1] * p.y) + T.m[2] * p.z) + T.m[3];
    o.y = ((T.m[4] * p.x + T.m[5] * p.y) + T.m[6] * p.z) + T.m[7];
    o.z = ((T.m[8] * p.x + T.m[9] * p.y) + T.m[10] * p.z) + T.m[11];
    o.w = p.w;
    return o;
}

// label decode (utils.cpp:64-65): numeric cast, & 0xFFFF, dynamic = 252..259
__device__ __forceinline__ bool is_dynamic_label(float intensity) {
    // static_cast<uint32_t>(float): x86 cvttss2si semantics for in-range values; labels are < 2^24
    const uint32_t u = (uint32_t)intensity;
    const uint32_t sem = u & 0xFFFFu;
    return sem >= 252u && sem <= 259u;
}

// R-POD bin of an egocentric point (erasor.cpp:104-110, 11-21).  Returns theta-major key
// sector*R + ring, or B if the point fails a gate.  The reference's arithmetic, operation by operation, in float64.
// (Round 3 measured a variant that decides ring and sector on float32 estimates and runs this path only near a boundary: bit-exact on
// the whole GPU suite, but k_voi_gather is not bound by these ~250 float64 instructions per point -- 28 us either way -- so it was dropped.)
__device__ __forceinline__ uint32_t bin_key_exact(const DP &P, float x, float y, float z, Counters *ctr) {
    uint32_t key = (uint32_t)P.B;
    const double dx = (double)x, dy = (double)y;
    const double r = sqrt(dx * dx + dy * dy);
    const bool gate = ((double)z < P.max_h) && ((double)z > P.min_h) && (r <= P.max_r);
    if (gate) {
        const double at = atan2(dy, dx);
        const double theta = (dy >= 0) ? at : (2 * PI_REF + at);
        const double q = theta / P.sector_size;
        const bool amb = (q != 0.0) && (fabs(q - rint(q)) < 1e-11);
        int sidx = (int)q;
        if (sidx > P.S - 1) sidx = P.S - 1;
        int ridx = (int)(r / P.ring_size);
        if (ridx > P.R - 1) ridx = P.R - 1;
        const bool neg = sidx < 0;  // y == -0.0f, x < 0: the reference throws (vector::at); defined clamp
        if (neg) sidx = 0;
        key = (uint32_t)(sidx * P.R + ridx);
        if (amb) atomicAdd(&ctr->n_ambiguous, 1u);
        if (neg) atomicAdd(&ctr->n_neg_sector, 1u);
    }
    return key;
}
// Round 4: the same key WITHOUT float64 for every point that is not within a hair of a ring or sector boundary.  The reference's
// arithmetic (float64 sqrt, atan2 and two divisions per point: ~270 instructions, the dominant cost of k_voi_gather) only matters
// where its rounding could move a point across a boundary; a float32 radius (|error| < 3e-5 m up to 100 m) and a float32 angle
// (8-term odd polynomial of min / max, |error| < 1e-6 rad with the quadrant folds) decide every point farther than
// 5e-4 m / 1e-5 rad from the boundaries of the cell they land in, i.e. all but ~4 in 10^4; the rest -- and everything odd:
// y == +-0 (the reference's negative-sector case), zeros, infinities, NaN -- takes bin_key_exact.  A point the fast path accepts
// cannot be `ambiguous` (that needs |q - rint(q)| < 1e-11) nor in a negative sector, so the counters agree too.
__device__ __forceinline__ float fast_rcpf(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(v);
#else
    return 1.0f / v;
#endif
}
__device__ __forceinline__ float fast_sqrtf(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(v);
#else
    return sqrtf(v);
#endif
}
__device__ __forceinline__ uint32_t bin_key(const DP &P, float x, float y, float z, Counters *ctr) {
    const bool zgate = ((double)z < P.max_h) && ((double)z > P.min_h);
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float rf = fast_sqrtf(x * x + y * y);
    const float max_rf = (float)P.max_r, ring_f = (float)P.ring_size, sector_f = (float)P.sector_size;
    constexpr float M_R = 5e-4f, M_A = 1e-5f;  // metres, radians
    if (zgate && rf > max_rf + M_R && rf < 3.0e38f) return (uint32_t)P.B;  // certainly beyond the range
    bool sure = zgate && ay > 0.f && mx < 1.0e18f && rf <= max_rf - M_R;
    const float a = mn * fast_rcpf(mx);
    const float s2 = a * a;
    float pl = -0.004054552409797907f;
    pl = fmaf(pl, s2, 0.021862907335162163f);
    pl = fmaf(pl, s2, -0.055912259966135025f);
    pl = fmaf(pl, s2, 0.09642193466424942f);
    pl = fmaf(pl, s2, -0.1390862762928009f);
    pl = fmaf(pl, s2, 0.19946566224098206f);
    pl = fmaf(pl, s2, -0.33329859375953674f);
    pl = fmaf(pl, s2, 0.9999993443489075f);
    float th = pl * a;                                   // atan(min / max) in [0, pi / 4]
    if (ay > ax) th = 1.57079632679489662f - th;         // first quadrant
    if (x < 0.f) th = 3.14159265358979324f - th;         // upper half plane
    if (y < 0.f) th = 6.28318530717958648f - th;         // [0, 2 pi)
    const int sf = (int)(th * (1.0f / sector_f));
    const float slo = (float)sf * sector_f;
    sure = sure && sf >= 0 && sf < P.S && (th - slo >= M_A) && ((slo + sector_f) - th >= M_A);
    const int rfi = (int)(rf * (1.0f / ring_f));
    const float rlo = (float)rfi * ring_f;
    sure = sure && rfi < P.R && (rf - rlo >= M_R) && ((rlo + ring_f) - rf >= M_R);
    if (sure) return (uint32_t)(sf * P.R + rfi);
    if (!zgate) return (uint32_t)P.B;
    return bin_key_exact(P, x, y, z, ctr);
}
// ---- small block-level helpers -------------------------------------------------------------------
// exclusive scan of one value per thread; sm >= 34 uint32; returns prefix, writes block total.
// Round 3: both levels on DPP row shifts (esort::wave_incl_scan) -- the shuffle ladder this replaces was six dependent
// ds_bpermute round trips plus a 16-step serial loop over the wavefront totals, ~2 us per call in the single-workgroup kernels
// of the dependency chain (k_srt4, k_layout4, the chunk scan), which call it two to five times each.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sm, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const uint32_t inc = esort::wave_incl_scan(v);
    __syncthreads();  // (sm may still be read by the previous call's second level)
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    const uint32_t wt = lane < nw ? sm[lane] : 0u;
    const uint32_t wi = esort::wave_incl_scan(wt);
    total = __builtin_amdgcn_readlane(wi, 63);
    const uint32_t pre = __builtin_amdgcn_readlane(wi - wt, __builtin_amdgcn_readfirstlane(wave));
    return pre + inc - v;
}

// N exclusive block scans behind ONE pair of barriers (round 6: srt4_body ran eight scans one after the other -- sixteen workgroup barriers of
// 1024 threads on the head of the early stream's chain).  sm: 16 * N words.
template <int N>
__device__ __forceinline__ void block_excl_scan_n(const uint32_t (&v)[N], uint32_t *sm, uint32_t (&pre)[N], uint32_t (&total)[N]) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t inc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) inc[j] = esort::wave_incl_scan(v[j]);
    __syncthreads();  // (sm may still be read by an earlier scan's second level)
    if (lane == 63) {
#pragma unroll
        for (int j = 0; j < N; ++j) sm[j * 16 + wave] = inc[j];
    }
    __syncthreads();
    const uint32_t w1 = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const uint32_t wt = lane < nw ? sm[j * 16 + lane] : 0u;
        const uint32_t wi = esort::wave_incl_scan(wt);
        total[j] = __builtin_amdgcn_readlane(wi, 63);
        pre[j] = __builtin_amdgcn_readlane(wi - wt, w1) + inc[j] - v[j];
    }
}

// wave-wide reductions on DPP row shifts (the value every lane gets back is lane 63's inclusive result): a __shfl_down / __shfl_xor
// ladder is six dependent ds_bpermute round trips (~100 cycles each) at the end of a wavefront that lives a few microseconds
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) { return __builtin_amdgcn_readlane(esort::wave_incl_scan(x), 63); }
__device__ __forceinline__ float wave_min_f(float x) {
    const int id = 0x7F800000;  // +inf: what a lane without a source contributes
    float v = x;
#define ERASOR_DPP_MINSTEP(ctrl, rmask) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(id, __float_as_int(v), ctrl, rmask, 0xf, false)))
    ERASOR_DPP_MINSTEP(0x111, 0xf);
    ERASOR_DPP_MINSTEP(0x112, 0xf);
    ERASOR_DPP_MINSTEP(0x114, 0xf);
    ERASOR_DPP_MINSTEP(0x118, 0xf);
    ERASOR_DPP_MINSTEP(0x142, 0xa);
    ERASOR_DPP_MINSTEP(0x143, 0xc);
#undef ERASOR_DPP_MINSTEP
    return __int_as_float((int)__builtin_amdgcn_readlane((uint32_t)__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float x) { return -wave_min_f(-x); }
// the same for unsigned keys (fkey_ord: the bounding boxes of k_bbox and of the per-bin voxelisation); IDENT: what a lane without a source
// contributes (~0u for a minimum, 0u for a maximum)
template <bool MAX>
__device__ __forceinline__ uint32_t wave_minmax_u(uint32_t x) {
    const int id = MAX ? 0 : -1;
    uint32_t v = x;
#define ERASOR_DPP_MMSTEP(ctrl, rmask)                                                                   \
    do {                                                                                                 \
        const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, ctrl, rmask, 0xf, false);  \
        v = MAX ? (o_ > v ? o_ : v) : (o_ < v ? o_ : v);                                                 \
    } while (0)
    ERASOR_DPP_MMSTEP(0x111, 0xf);
    ERASOR_DPP_MMSTEP(0x112, 0xf);
    ERASOR_DPP_MMSTEP(0x114, 0xf);
    ERASOR_DPP_MMSTEP(0x118, 0xf);
    ERASOR_DPP_MMSTEP(0x142, 0xa);
    ERASOR_DPP_MMSTEP(0x143, 0xc);
#undef ERASOR_DPP_MMSTEP
    return __builtin_amdgcn_readlane(v, 63);
}

// ================================================================================================
// (1) voi_split — THE HBM-bound kernel.  fetch_VoI's membership test (OMU.cpp:391-395) over every
// physical entry of the map store.  One wavefront per 1024-entry chunk (CHUNK), 16 loads in flight per lane.
//   F region: dense float4 (the previous step's VoI-resident part of the map, nF entries)
//   O region: outskirts, split SoA-of-pairs layout {x,y} | {z,intensity}; only {x,y} is streamed.
// Outputs: per 64-point tile an in-VoI mask and a valid mask, per chunk (voi | valid<<16).
// ================================================================================================
// result block in pinned host memory: k_step_end stores the step's state and counters there, so the host needs no D2H copies
struct HostOut {
    DevState st;
    Counters ctr;
    unsigned long long t_end;  // measurement: the constant-rate counter when the step's end ran (with st.t_open: the main chain's span)
    unsigned long long seq;    // number of the step these results belong to: written last, the host polls it
};

// end of a step: fold in the query side's counters and voxel count, commit the map sizes, report to the pinned host block
__device__ __forceinline__ void step_end_body(DevState *st, Counters *ctr, HostOut *out, const unsigned long long *lab_slots, const Counters *qctr,
                                              const uint32_t *q_nvox, unsigned long long seq) {
    // round 3: everything is READ first (one memory round trip: the loads are independent and nothing is stored in between), then
    // computed, then written -- interleaved read-modify-writes of st / ctr cost a round trip each (6 us for this one-thread kernel)
    DevState s = *st;
    Counters c = *ctr;
    const Counters q = *qctr;
    const uint32_t nv = *q_nvox;
    unsigned long long ns = 0, nd = 0;
    if (lab_slots) {
        unsigned long long a[16], b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a[i] = lab_slots[i * 8];
            b[i] = lab_slots[i * 8 + 1];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ns += a[i];
            nd += b[i];
        }
        s.F_static = ns;
        s.F_dynamic = nd;
    }
    c.n_neg_sector += q.n_neg_sector;
    c.n_ambiguous += q.n_ambiguous;
    c.n_degenerate += q.n_degenerate;
    c.n_voxel_overflow += q.n_voxel_overflow;
    c.n_sort_fallback += q.n_sort_fallback;
    if (q.sort_qoverflow) c.sort_qoverflow = q.sort_qoverflow;
    if (q.err) c.err = q.err;
    s.q_nvox = nv;
    if (!(c.err || c.sort_qoverflow)) {
        s.nF = s.nF_new;
        s.nF_valid = s.nF_valid_new;
        s.o_begin = s.o_new_begin;
    }
    *st = s;
    *ctr = c;
    if (out) {
        out->st = s;
        out->ctr = c;
        out->t_end = wall_clock64();
        if (g_stamps_on) g_stamps[15] = out->t_end;
        __threadfence_system();
        *(volatile unsigned long long *)&out->seq = seq;
    }
}

__global__ void k_step_end(DevState *st, Counters *ctr, HostOut *out, const unsigned long long *lab_slots, const Counters *qctr,
                           const uint32_t *q_nvox, unsigned long long seq) {
    step_end_body(st, ctr, out, lab_slots, qctr, q_nvox, seq);
}
// Round 4: the step's end as a passenger of the NEXT step's VoI split.  When the next node's pose is known the split is launched right
// behind the step anyway (dev != nullptr below); one extra workgroup of that launch then does k_step_end's work -- everything the step
// wrote is visible at the kernel boundary either way -- and the launch of its own (5 us of dependent loads on one thread + a boundary)
// leaves the main stream's chain.  The other workgroups cannot wait for that commit: they derive the two extents themselves.
struct StepEnd {
    DevState *st;
    Counters *ctr;
    HostOut *out;
    const unsigned long long *lab_slots;
    const Counters *qctr;
    const uint32_t *q_nvox;
    unsigned long long seq;
};

// Round 3: every outskirts chunk carries the bounding box of its valid entries and their number (OMeta, 32 B per 8 KB of {x,y}).  A chunk
// whose box lies outside the VoI circle is NOT READ: its counts come from the record.  The outskirts are written in runs that are close
// in space (set_map keeps the input order -- a voxelised map is sorted by voxel index --, later the points that leave the VoI, step by
// step), and a VoI is a few per cent of a map: most of the pass's bytes go.  The test is conservative -- distance from the centre to the
// box in float64, skipped only beyond r^2 * (1 + 1e-9) -- so no entry that could pass `d^2 < r^2` (OMU.cpp:394) is ever skipped, and the
// masks of the chunks that are read are what they were.  A record is (re)built by the pass itself whenever it reads the chunk
// (known = 1); it is dropped (known = 0) for the chunks a step prepends leaving points to (k_chunk_scan_*), its count is corrected when
// entering points are tombstoned (k_voi_gather), and the host clears all records whenever something else rewrites the store.
struct OMeta {
    float xmin, xmax, ymin, ymax;
    uint32_t valid, known, pad0, pad1;
};
struct OvSplit {  // round 5: the split of an OVERLAPPED step (see k_voi_split); late == nullptr: off
    const LateEnt *late = nullptr;
    const DevState *prev = nullptr;        // the state of the step in flight: extents of the region it writes, entries of its late table
    unsigned long long *lmask = nullptr;   // per tile of the VoI-resident region: the slots that belong to the late table
};
static constexpr uint32_t SPLIT_LATE_LDS = 512;     // late-table entries k_voi_split keeps in LDS (beyond: read where they lie)
static constexpr uint32_t CINFO_READ = 0x80000000u;  // cinfo: voi count | valid count << 16 | "the chunk was read"
static constexpr uint32_t CINFO_HMASK = 0x7FFFu;
__global__ __launch_bounds__(256) void k_voi_split(const float4 *__restrict__ F, uint32_t nF, uint32_t nFchunks,
                                                    const float2 *__restrict__ Oxy, uint32_t o_begin, uint32_t o_chunk0,
                                                    uint32_t nOchunks, double xc, double yc, double r2,
                                                    unsigned long long *__restrict__ vmask,
                                                    unsigned long long *__restrict__ hmask, uint32_t *__restrict__ cinfo,
                                                    const DevState *__restrict__ dev, uint32_t capO_chunks, uint32_t cap_chunks,
                                                    OMeta *__restrict__ ometa, StepEnd se, OvSplit ov = OvSplit()) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t n_late = 0;
    CHAIN_STAMP(5);
    if (ov.late) {
        // round 5, OVERLAPPED steps: the pass of step k + 1 beside step k's per-bin launch.  The region it reads is being written in the
        // reserved layout: its extents are final (k_srt4), the slots of the late table are NOT -- their masks come from the table:
        // a data slot counts as a VoI entry (it keeps a place in VoI order for the point that may come), a leaving reservation as a
        // valid entry outside the VoI (a place in the outskirts' order); lmask marks both for the gather, which leaves them alone
        nF = ov.prev->nF_new;
        nFchunks = (nF + CHUNK - 1) / CHUNK;
        o_begin = ov.prev->o_new_begin;
        o_chunk0 = o_begin / CHUNK;
        nOchunks = capO_chunks - o_chunk0;
        n_late = ov.prev->n_late;
        if (nFchunks + nOchunks > cap_chunks) return;  // (k_chunk_scan_* sees that too: ov_bad)
    } else if (dev && se.st) {
        // the launch also ENDS the previous step (se): its last workgroup is k_step_end; the others take the extents that commit will
        // publish straight from what the step left (the same decision, the same values whichever comes first)
        if (blockIdx.x == gridDim.x - 1) {
            if (threadIdx.x == 0) step_end_body(se.st, se.ctr, se.out, se.lab_slots, se.qctr, se.q_nvox, se.seq);
            return;
        }
        nwaves -= blockDim.x >> 6;
        const uint32_t e0 = se.ctr->err, e1 = se.ctr->sort_qoverflow, e2 = se.qctr->err, e3 = se.qctr->sort_qoverflow;
        const uint32_t nF_old = dev->nF, nF_new = dev->nF_new, ob_old = dev->o_begin, ob_new = dev->o_new_begin;
        const bool ok = !(e0 | e1 | e2 | e3);
        nF = ok ? nF_new : nF_old;
        nFchunks = (nF + CHUNK - 1) / CHUNK;
        o_begin = ok ? ob_new : ob_old;
        o_chunk0 = o_begin / CHUNK;
        nOchunks = capO_chunks - o_chunk0;
        if (nFchunks + nOchunks > cap_chunks) return;
    } else if (dev) {
        // launched AHEAD of its step (right behind the previous step's k_step_end, while the host is still collecting that
        // step's results): the extent of the two regions is what that step has just committed on the device
        nF = dev->nF;
        nFchunks = (nF + CHUNK - 1) / CHUNK;
        o_begin = dev->o_begin;
        o_chunk0 = o_begin / CHUNK;
        nOchunks = capO_chunks - o_chunk0;
        if (nFchunks + nOchunks > cap_chunks) return;  // (the step sees that too and runs the pass itself)
    }
    const uint32_t nchunks = nFchunks + nOchunks;
    // round 6: a workgroup that meets the VoI-resident region takes the late table (two entries per reverted bin of the step in flight) into
    // LDS once -- a wavefront used to walk it in global memory: five dependent round trips of binary search at its head, then an entry
    // per tile
    __shared__ LateEnt s_late[SPLIT_LATE_LDS];
    const bool late_lds = n_late != 0u && n_late <= SPLIT_LATE_LDS && (blockIdx.x * (blockDim.x >> 6) < nFchunks || nwaves < nFchunks);  // (the same in every thread of the workgroup)
    if (late_lds) {
        for (uint32_t i = threadIdx.x; i < n_late; i += blockDim.x) s_late[i] = ov.late[i];
        __syncthreads();
    }
    const auto late_at = [&](uint32_t i) -> LateEnt { return late_lds ? s_late[i] : ov.late[i]; };
    for (uint32_t c = wid; c < nchunks; c += nwaves) {
        unsigned long long myv = 0, myh = 0, myl = 0;
        uint32_t cv = 0, ch = 0, read_flag = 0;
        float bxmin = __int_as_float(0x7F800000), bxmax = __int_as_float(0xFF800000), bymin = bxmin, bymax = bxmax;
        if (c >= nFchunks) {
            const uint32_t a = c - nFchunks + o_chunk0;  // the chunk's number in the store
            if (ometa) {  // can the chunk be skipped?  (the record is the same for every lane: scalar branch)
                const OMeta m = ometa[a];
                if (__builtin_amdgcn_readfirstlane(m.known)) {
                    const double bx = fmax(fmax((double)m.xmin - xc, xc - (double)m.xmax), 0.0);
                    const double by = fmax(fmax((double)m.ymin - yc, yc - (double)m.ymax), 0.0);
                    const bool out = bx * bx + by * by > r2 * (1.0 + 1e-9);  // (an empty chunk has an inverted box: infinitely far)
                    if (__builtin_amdgcn_readfirstlane((uint32_t)out)) {
                        if (lane == 0) cinfo[c] = m.valid << 16;  // (no VoI entry: k_voi_gather never looks at its masks)
                        esort::wave_sync();  // (the wavefront goes on together: the next chunk's ballots want every lane)
                        continue;
                    }
                }
            }
            // ---- outskirts: stream {x,y} only; all 16 loads of the chunk are issued before the first use.  Unconditional
            // loads matter: predicating them (even wave-uniformly) serialises the batch (measured 2.9 vs 5.5 TB/s).
            const uint32_t base = a * CHUNK + lane;
            float2 p[CHUNK_TILES];
#pragma unroll
            for (int t = 0; t < CHUNK_TILES; ++t) p[t] = Oxy[base + t * TILE];  // the buffer covers whole chunks: no bounds test
            const bool first = (c == nFchunks);  // only the chunk that contains o_begin has entries in front of the region
#pragma unroll
            for (int t = 0; t < CHUNK_TILES; ++t) {
                // double dist_square = pow(pt.x - x_criterion, 2) + pow(pt.y - y_criterion, 2)  (OMU.cpp:394)
                const double dx = (double)p[t].x - xc, dy = (double)p[t].y - yc;
                bool valid = __float_as_uint(p[t].x) != HOLE_BITS;
                if (first) valid = valid && (base + t * TILE >= o_begin);
                const bool in = valid && (dx * dx + dy * dy < r2);
                const unsigned long long vm = __ballot(in), hm = __ballot(valid);
                if ((int)lane == t) {
                    myv = vm;
                    myh = hm;
                }
                cv += __popcll(vm);
                ch += __popcll(hm);
                if (valid) {
                    bxmin = fminf(bxmin, p[t].x);
                    bxmax = fmaxf(bxmax, p[t].x);
                    bymin = fminf(bymin, p[t].y);
                    bymax = fmaxf(bymax, p[t].y);
                }
            }
            if (ometa) {  // the chunk's record, from what was just read
                bxmin = wave_min_f(bxmin);
                bxmax = wave_max_f(bxmax);
                bymin = wave_min_f(bymin);
                bymax = wave_max_f(bymax);
                if (lane == 0) {
                    OMeta m;
                    m.xmin = bxmin;
                    m.xmax = bxmax;
                    m.ymin = bymin;
                    m.ymax = bymax;
                    m.valid = ch;
                    m.known = 1u;
                    m.pad0 = m.pad1 = 0u;
                    ometa[a] = m;
                }
            }
            read_flag = CINFO_READ;
        } else {
            // ---- F region: dense float4, two half-chunks to bound the registers ----
            const uint32_t base = c * CHUNK + lane;
            uint32_t e0 = 0;  // first late entry that ends beyond the chunk's start (the table is sorted by position)
            if (n_late) {
                uint32_t lo = 0, hi = n_late;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const LateEnt e = late_at(mid);
                    if (e.start + e.ntotal <= c * CHUNK) lo = mid + 1;
                    else hi = mid;
                }
                e0 = lo;
            }
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                float px[CHUNK_TILES / 2], py[CHUNK_TILES / 2];
#pragma unroll
                for (int t = 0; t < CHUNK_TILES / 2; ++t) {
                    const uint32_t idx = base + (hlf * (CHUNK_TILES / 2) + t) * TILE;
                    const float4 q = idx < nF ? F[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                    px[t] = q.x;
                    py[t] = q.y;
                }
#pragma unroll
                for (int t = 0; t < CHUNK_TILES / 2; ++t) {
                    const int tt = hlf * (CHUNK_TILES / 2) + t;
                    bool valid = base + tt * TILE < nF && __float_as_uint(px[t]) != HOLE_BITS;  // (round 5: a reserved slot nothing filled)
                    const double dx = (double)px[t] - xc, dy = (double)py[t] - yc;
                    bool in = valid && (dx * dx + dy * dy < r2);
                    unsigned long long lm = 0ull;
                    if (n_late) {  // (uniform: the table entries that meet this tile)
                        const uint32_t pos = base + tt * TILE, tile0 = c * CHUNK + tt * TILE;
                        bool ldat = false, lph = false;
                        for (uint32_t e = e0; e < n_late; ++e) {
                            const LateEnt le = late_at(e);
                            if (le.start >= tile0 + TILE) break;
                            if (le.start + le.ntotal <= tile0) continue;
                            ldat = ldat || (pos >= le.start && pos < le.start + le.ndata);
                            lph = lph || (pos >= le.start + le.ndata && pos < le.start + le.ntotal);
                        }
                        if (ldat || lph) {
                            valid = true;
                            in = ldat;
                        }
                        lm = __ballot(ldat || lph);
                    }
                    const unsigned long long vm = __ballot(in), hm = __ballot(valid);
                    if ((int)lane == tt) {
                        myv = vm;
                        myh = hm;
                        myl = lm;
                    }
                    cv += __popcll(vm);
                    ch += __popcll(hm);
                }
            }
        }
        if (lane < CHUNK_TILES) {
            vmask[(size_t)c * CHUNK_TILES + lane] = myv;
            hmask[(size_t)c * CHUNK_TILES + lane] = myh;
            if (ov.lmask && c < nFchunks) ov.lmask[(size_t)c * CHUNK_TILES + lane] = myl;
        }
        if (lane == 0) cinfo[c] = cv | (ch << 16) | read_flag;
    }
}

// ---- chunk prefix sums (two-level) --------------------------------------------------------------
// level 1: 1024 chunks per block; local exclusive prefixes of the voi and valid counts + block totals

// level 2 (single block): exclusive scan of the block totals in place; derive the step's VoI sizes.

// level 1 of the chunk-count scan: 1024 chunks per workgroup, local prefixes + workgroup totals
__global__ __launch_bounds__(256) void k_chunk_scan_local(const uint32_t *__restrict__ cinfo, uint32_t nchunks,
                                                           uint32_t *__restrict__ pvl, uint32_t *__restrict__ phl,
                                                           uint32_t *__restrict__ topv, uint32_t *__restrict__ toph, uint32_t *__restrict__ topr,
                                                           // round 4: != 0: launched AHEAD (see k_chunk_scan_one): the extents come from the
                                                           // committed device state, the grid is an upper bound (workgroups beyond write zeros)
                                                           const DevState *st_dev, uint32_t capO_chunks_dev, uint32_t cap_dev,
                                                           // round 5 (overlapped steps): st_dev is the state of the step IN FLIGHT, the extents are
                                                           // those of the region it is writing
                                                           uint32_t from_new = 0u) {
    __shared__ uint32_t sm[40];
    if (capO_chunks_dev) {
        const uint32_t nF = from_new ? st_dev->nF_new : st_dev->nF, ob = from_new ? st_dev->o_new_begin : st_dev->o_begin;
        nchunks = min((nF + CHUNK - 1) / CHUNK + (capO_chunks_dev - ob / CHUNK), cap_dev);
    }
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t v[4], h[4];
    uint32_t sv = 0, sh = 0, sr = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t ci = (base + j < nchunks) ? cinfo[base + j] : 0u;
        v[j] = ci & 0xFFFFu;
        h[j] = (ci >> 16) & CINFO_HMASK;
        sv += v[j];
        sh += h[j];
        sr += ci >> 31;
    }
    uint32_t tv, th, tr;
    uint32_t pv = block_excl_scan(sv, sm, tv);
    uint32_t ph = block_excl_scan(sh, sm, th);
    (void)block_excl_scan(sr, sm, tr);
    if (threadIdx.x == 0 && topr) topr[blockIdx.x] = tr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < nchunks) {
            pvl[base + j] = pv;
            phl[base + j] = ph;
        }
        pv += v[j];
        ph += h[j];
    }
    if (threadIdx.x == 0) {
        topv[blockIdx.x] = tv;
        toph[blockIdx.x] = th;
    }
}

// level 2 (single workgroup) -- and the opening of the step's state: the host's mirror of the device state replaces it,
// counters and tallies are cleared (this is the first launch of a step that needs them); then the workgroup totals are
// scanned in place and the step's VoI sizes derived.
__global__ __launch_bounds__(1024) void k_chunk_scan_top(uint32_t *__restrict__ topv, uint32_t *__restrict__ toph, uint32_t ntop,
                                                          const uint32_t *__restrict__ pvl, const uint32_t *__restrict__ phl,
                                                          uint32_t nchunks, uint32_t nFchunks, DevState *st, Counters *ctr, DevState init,
                                                          unsigned long long *lab_slots, uint32_t *mb_tot, uint32_t mb_n,
                                                          const uint32_t *__restrict__ topr, OMeta *__restrict__ ometa,
                                                          uint32_t capO_chunks_dev, uint32_t cap_dev /* ahead: see k_chunk_scan_one */,
                                                          const DevState *prev = nullptr, uint32_t cap_voi = 0u /* round 5: see k_chunk_scan_one */) {
    __shared__ uint32_t sm[40];
    __shared__ uint32_t carry[2];
    bool ov_bad = false;
    if (capO_chunks_dev) {
        const DevState *src = prev ? prev : st;
        const uint32_t nF = prev ? src->nF_new : src->nF, ob = prev ? src->o_new_begin : src->o_begin;
        nFchunks = (nF + CHUNK - 1) / CHUNK;
        ov_bad = nFchunks + (capO_chunks_dev - ob / CHUNK) > cap_dev;
        nchunks = min(nFchunks + (capO_chunks_dev - ob / CHUNK), cap_dev);  // (beyond: the step sees that and runs the scan itself)
        ntop = max(1u, (nchunks + 1023u) / 1024u);
        init = *src;
        init.nF = nF;
        init.o_begin = ob;
        __syncthreads();  // (everybody has read the state before thread 0 replaces it)
    }
    uint32_t n_read = 0;
    if (topr)
        for (uint32_t i = threadIdx.x; i < ntop; i += blockDim.x) n_read += topr[i];
    {
        uint32_t tr;
        (void)block_excl_scan(n_read, sm, tr);
        n_read = tr;
    }
    if (lab_slots && threadIdx.x < 128) lab_slots[threadIdx.x] = 0;
    for (uint32_t b = threadIdx.x; b < mb_n; b += blockDim.x) mb_tot[b] = 0;
    if (threadIdx.x == 0) {
        ctr->n_neg_sector = ctr->n_ambiguous = ctr->n_degenerate = ctr->n_voxel_overflow = ctr->n_sort_fallback = 0;
        ctr->sort_qoverflow = ctr->err = 0;
        carry[0] = carry[1] = 0;
    }
    __syncthreads();
    for (uint32_t base = 0; base < ntop; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < ntop ? topv[i] : 0u, h = i < ntop ? toph[i] : 0u;
        uint32_t tv, th;
        const uint32_t pv = block_excl_scan(v, sm, tv);
        const uint32_t ph = block_excl_scan(h, sm, th);
        const uint32_t c0 = carry[0], c1 = carry[1];
        if (i < ntop) {
            topv[i] = c0 + pv;
            toph[i] = c1 + ph;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            carry[0] = c0 + tv;
            carry[1] = c1 + th;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        DevState s = init;
        const uint32_t voi_total = carry[0], valid_total = carry[1];
        uint32_t voiF = voi_total, validF = valid_total;
        if (nFchunks < nchunks) {
            voiF = pvl[nFchunks] + topv[nFchunks >> 10];
            validF = phl[nFchunks] + toph[nFchunks >> 10];
        }
        s.F_static = s.F_dynamic = 0;
        s.n_rev = 0;
        s.voi_total = voi_total;
        s.valid_total = valid_total;
        s.voiF = voiF;
        s.validF = validF;
        s.n_leaving = validF - voiF;
        s.o_new_begin = s.o_begin - (validF - voiF);
        s.n_o_read = n_read;
        s.t_open = wall_clock64();  // (the two-launch scan: stamped here, one launch late)
        s.n_voi_dead = s.n_late_leaving = 0;
        s.ov_bad = (prev && (ov_bad || (uint64_t)(validF - voiF) + CHUNK > s.o_begin || voi_total > cap_voi)) ? 1u : 0u;
        if (s.ov_bad) s.o_new_begin = s.o_begin;
        *st = s;
        // the chunks that receive this step's leaving points (k_voi_gather prepends them to the outskirts): their records are void
        if (ometa && s.o_new_begin < s.o_begin)
            for (uint32_t a = s.o_new_begin / CHUNK; a <= (s.o_begin - 1u) / CHUNK; ++a) ometa[a].known = 0u;
    }
}

// Both levels in ONE launch (maps of up to ~16 M entries: <= 16384 chunks): a single workgroup, 16 consecutive chunk counts
// per thread, writes ABSOLUTE prefixes (the per-group tops are all zero) and opens the step's state exactly like
// k_chunk_scan_top.  One kernel boundary less on the main stream's dependency chain.
// Round 3: the thread's 16 counts come and go as four 128-bit accesses each way (a wavefront then touches 32 cache lines per
// access, all of them fully used within the four), both sums are scanned together on DPP row shifts -- no LDS transposes, two
// workgroup barriers in all (the staged version took 13.9 us for 10.4 k counts).
__global__ __launch_bounds__(1024) void k_chunk_scan_one(const uint32_t *__restrict__ cinfo, uint32_t nchunks, uint32_t *__restrict__ pvl,
                                                          uint32_t *__restrict__ phl, uint32_t *__restrict__ topv, uint32_t *__restrict__ toph,
                                                          uint32_t ntop, uint32_t nFchunks, DevState *st, Counters *ctr, DevState init,
                                                          unsigned long long *lab_slots, uint32_t *mb_tot, uint32_t mb_n, OMeta *__restrict__ ometa,
                                                          // round 4: != 0: launched AHEAD, right behind the next step's VoI split (which was launched
                                                          // ahead too): the extents are what the step in front has committed on the device, and so is
                                                          // the state this step starts from -- the host is still collecting that step's results
                                                          uint32_t capO_chunks_dev, uint32_t cap_dev /* chunks the prefix arrays hold */,
                                                          // round 5, OVERLAPPED steps (prev != nullptr): launched beside the per-bin launch of the
                                                          // step in flight, whose state is *prev -- this step starts from it, with the extents of
                                                          // the region that step is writing (final since its k_srt4) --, into a state of its own
                                                          // (*st); cap_voi: entries the VoI-order arrays hold (beyond, or without room in front of
                                                          // the outskirts: ov_bad, and the passes behind do nothing)
                                                          const DevState *prev = nullptr, uint32_t cap_voi = 0u) {
    __shared__ uint32_t sm[40];
    __shared__ uint32_t s_voiF, s_validF;
    const unsigned long long t_open = wall_clock64();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    CHAIN_STAMP(6);
    bool ov_bad = false;
    // round 6: launched ahead, the number of chunks comes from the device state -- the counts used to be fetched BEHIND it.  They are now
    // fetched beside it, as far as the array goes (cap_dev: what cinfo and the prefix arrays hold); what lies beyond the chunks in use
    // is dropped below.  One dependent round trip less for a launch that is a single workgroup of them.
    const uint32_t lim = capO_chunks_dev ? min(cap_dev, 16384u) : nchunks;
    const uint32_t base = tid * 16;
    uint32_t ci[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t b4 = base + g * 4;
        if (b4 + 3 < lim) {
            const uint4 q = *reinterpret_cast<const uint4 *>(cinfo + b4);
            ci[g * 4 + 0] = q.x;
            ci[g * 4 + 1] = q.y;
            ci[g * 4 + 2] = q.z;
            ci[g * 4 + 3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) ci[g * 4 + j] = b4 + j < lim ? cinfo[b4 + j] : 0u;
        }
    }
    if (capO_chunks_dev) {
        const DevState *src = prev ? prev : st;
        const uint32_t nF = prev ? src->nF_new : src->nF, ob = prev ? src->o_new_begin : src->o_begin;
        nFchunks = (nF + CHUNK - 1) / CHUNK;
        ov_bad = nFchunks + (capO_chunks_dev - ob / CHUNK) > min(cap_dev, 16384u);
        nchunks = min(nFchunks + (capO_chunks_dev - ob / CHUNK), min(cap_dev, 16384u));  // (beyond: the step sees that and runs the scan itself)
        ntop = max(1u, (nchunks + 1023u) / 1024u);
        init = *src;
        init.nF = nF;
        init.o_begin = ob;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (base + j >= nchunks) ci[j] = 0u;  // (counts of chunks this step does not have: whatever an earlier step left there)
    }
    // (the step's housekeeping rides in the shadow of the loads)
    if (lab_slots && tid < 128) lab_slots[tid] = 0;
    for (uint32_t b = tid; b < mb_n; b += blockDim.x) mb_tot[b] = 0;
    for (uint32_t t = tid; t < ntop; t += blockDim.x) {
        topv[t] = 0;
        toph[t] = 0;
    }
    if (tid == 0) {
        ctr->n_neg_sector = ctr->n_ambiguous = ctr->n_degenerate = ctr->n_voxel_overflow = ctr->n_sort_fallback = 0;
        ctr->sort_qoverflow = ctr->err = 0;
    }
    uint32_t sv = 0, sh = 0, sr = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        sr += ci[j] >> 31;  // (the "chunk was read" flag of the VoI split: counted, then dropped)
        ci[j] &= ~CINFO_READ;
        sv += ci[j] & 0xFFFFu;
        sh += ci[j] >> 16;
    }
    sr = wave_sum(sr);
    const uint32_t iv = esort::wave_incl_scan(sv), ih = esort::wave_incl_scan(sh);
    if (lane == 63) {
        sm[wave] = iv;
        sm[16 + wave] = ih;
        sm[32 + (wave & 7u)] = 0u;
    }
    __syncthreads();
    if (lane == 0) atomicAdd(&sm[32], sr);
    const uint32_t nw = blockDim.x >> 6;
    const uint32_t wv = lane < nw ? sm[lane] : 0u, wh = lane < nw ? sm[16 + lane] : 0u;
    const uint32_t wiv = esort::wave_incl_scan(wv), wih = esort::wave_incl_scan(wh);
    const uint32_t cv = __builtin_amdgcn_readlane(wiv, 63), ch = __builtin_amdgcn_readlane(wih, 63);  // totals
    const uint32_t w_ = __builtin_amdgcn_readfirstlane(wave);
    uint32_t pv = __builtin_amdgcn_readlane(wiv - wv, w_) + iv - sv;  // exclusive prefixes of this thread's first chunk
    uint32_t ph = __builtin_amdgcn_readlane(wih - wh, w_) + ih - sh;
    if (nFchunks >= base && nFchunks < base + 16) {  // the F region's share of the totals
        uint32_t a = pv, b2 = ph;
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j)  // (static indices: a run-time index would put ci[] into scratch memory for every thread)
            if (base + j < nFchunks) {
                a += ci[j] & 0xFFFFu;
                b2 += ci[j] >> 16;
            }
        s_voiF = a;
        s_validF = b2;
    }
    uint32_t ov[16], oh[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        ov[j] = pv;
        oh[j] = ph;
        pv += ci[j] & 0xFFFFu;
        ph += ci[j] >> 16;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t b4 = base + g * 4;
        if (b4 + 3 < nchunks) {
            *reinterpret_cast<uint4 *>(pvl + b4) = make_uint4(ov[g * 4], ov[g * 4 + 1], ov[g * 4 + 2], ov[g * 4 + 3]);
            *reinterpret_cast<uint4 *>(phl + b4) = make_uint4(oh[g * 4], oh[g * 4 + 1], oh[g * 4 + 2], oh[g * 4 + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (b4 + j < nchunks) {
                    pvl[b4 + j] = ov[g * 4 + j];
                    phl[b4 + j] = oh[g * 4 + j];
                }
        }
    }
    __syncthreads();
    if (tid == 0) {
        DevState s = init;
        const uint32_t voi_total = cv, valid_total = ch;
        uint32_t voiF = voi_total, validF = valid_total;
        if (nFchunks < nchunks) {
            voiF = s_voiF;
            validF = s_validF;
        }
        s.F_static = s.F_dynamic = 0;
        s.n_rev = 0;
        s.voi_total = voi_total;
        s.valid_total = valid_total;
        s.voiF = voiF;
        s.validF = validF;
        s.n_leaving = validF - voiF;
        s.o_new_begin = s.o_begin - (validF - voiF);
        s.n_o_read = sm[32];
        s.t_open = t_open;
        s.n_voi_dead = s.n_late_leaving = 0;
        s.ov_bad = (prev && (ov_bad || (uint64_t)(validF - voiF) + CHUNK > s.o_begin || voi_total > cap_voi)) ? 1u : 0u;
        if (s.ov_bad) s.o_new_begin = s.o_begin;
        *st = s;
        // the chunks that receive this step's leaving points (k_voi_gather prepends them to the outskirts): their records are void
        if (ometa && s.o_new_begin < s.o_begin)
            for (uint32_t a = s.o_new_begin / CHUNK; a <= (s.o_begin - 1u) / CHUNK; ++a) ometa[a].known = 0u;
    }
}

// ================================================================================================
// (2) voi_gather — for every set VoI bit: fetch the point, egocentric transform (OMU.cpp:435-437),
// R-POD key (erasor.cpp:124-139), write into VoI order; tombstone outskirts sources; move the
// points that left the VoI from the F region to the front of the outskirts region.
// ================================================================================================
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_voi_gather(const float4 *__restrict__ F, uint32_t nF, uint32_t nFchunks,
                                                     float2 *__restrict__ Oxy, float2 *__restrict__ Ozi, uint32_t o_chunk0,
                                                     uint32_t nOchunks, const unsigned long long *__restrict__ vmask,
                                                     const unsigned long long *__restrict__ hmask,
                                                     const uint32_t *__restrict__ cinfo, const uint32_t *__restrict__ pvl,
                                                     const uint32_t *__restrict__ phl, const uint32_t *__restrict__ topv,
                                                     const uint32_t *__restrict__ toph, Xf To2b, DP P, DevState *st,
                                                     Counters *ctr, const Counters *qctr, float4 *__restrict__ voi_ego,
                                                     uint32_t *__restrict__ voi_key, uint32_t *__restrict__ voi_src, OMeta *__restrict__ ometa,
                                                     // round 5, OVERLAPPED steps (lmask != nullptr): launched beside the per-bin launch of the
                                                     // step in flight.  The extents come from *st (opened by the chunk scan ahead); the slots of
                                                     // the late table (lmask) are left alone -- k_late_gather fills their places when the points
                                                     // are there --; entering outskirts entries are NOT tombstoned yet (k_o_commit, once the step
                                                     // is certain: until then a getter may still read the store as the step in flight leaves it)
                                                     const unsigned long long *__restrict__ lmask = nullptr, uint32_t capO_chunks = 0u) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint64_t lt = lanemask_lt();
    const bool ovm = lmask != nullptr;
    CHAIN_STAMP(7);
    // (round 4: the error flags, the step's state and the first item's records are fetched TOGETHER -- behind one another they were three
    // dependent round trips at the head of a wavefront that lives five or six; round 6: the overlapped step's flag and extents with them)
    const uint32_t err_a = ctr->err, err_b = qctr->err;
    const uint32_t validF = st->validF, voiF = st->voiF, o_new_begin = st->o_new_begin;
    if (ovm) {
        const uint32_t f_ov = st->ov_bad, nF_dev = st->nF, ob_dev = st->o_begin;
        if (f_ov) return;
        nF = nF_dev;
        nFchunks = (nF + CHUNK - 1) / CHUNK;
        o_chunk0 = ob_dev / CHUNK;
        nOchunks = capO_chunks - o_chunk0;
    }
    uint32_t dyn_leave = 0, stat_leave = 0, dyn_enter = 0, stat_enter = 0;
    // Work items: a chunk of the VoI-resident region is nearly all VoI (every tile fetches, transforms and runs the R-POD key), and
    // there are only ~800 of them for a 0.8 M-point VoI -- fewer wavefronts than SIMDs, each with 16 dependent rounds.  They are cut
    // into GATHER_SUB pieces of CHUNK_TILES / GATHER_SUB tiles (a piece finds its offsets from the popcounts of the chunk's earlier
    // masks).  Outskirts chunks stay whole: almost all are skipped.
    const uint32_t nitems = nFchunks * GATHER_SUB + nOchunks;
    constexpr int PT = CHUNK_TILES / (int)GATHER_SUB;  // tiles of a piece
    for (uint32_t w = wid; w < nitems; w += nwaves) {
        const bool isF = w < nFchunks * GATHER_SUB;
        const uint32_t c = isF ? w / GATHER_SUB : w - nFchunks * (GATHER_SUB - 1);
        const int t_lo = isF ? (int)(w % GATHER_SUB) * PT : 0;
        const int t_hi = isF ? t_lo + PT : CHUNK_TILES;
        const uint32_t ci = cinfo[c];
        const uint32_t pvl_c = pvl[c], phl_c = phl[c], topv_c = topv[c >> 10], toph_c = toph[c >> 10];
        const unsigned long long mv = lane < CHUNK_TILES ? vmask[(size_t)c * CHUNK_TILES + lane] : 0ull;
        const unsigned long long mh = lane < CHUNK_TILES ? hmask[(size_t)c * CHUNK_TILES + lane] : 0ull;
        const unsigned long long ml = (ovm && isF && lane < CHUNK_TILES) ? lmask[(size_t)c * CHUNK_TILES + lane] : 0ull;
        if (err_a || err_b) return;  // the voxelisation of this step's scan failed: do not touch the map store
        const uint32_t cv = ci & 0xFFFFu, ch = (ci >> 16) & CINFO_HMASK;
        if (cv == 0 && !(isF && ch > cv)) continue;
        uint32_t pv = pvl_c + topv_c;
        uint32_t ph = phl_c + toph_c;
        if (t_lo > 0) {  // entries of the chunk's tiles before this piece
            // (lanes >= CHUNK_TILES hold empty masks; both counts are < 2^16: one packed DPP reduction)
            const uint32_t a2 = wave_sum(((int)lane < t_lo) ? ((uint32_t)__popcll(mv) | ((uint32_t)__popcll(mh) << 16)) : 0u);
            pv += a2 & 0xFFFFu;
            ph += a2 >> 16;
        }
        if (isF) {
            // the piece's points are fetched before the first of them is used: one round trip per piece instead of one per tile
            float4 pp[PT];
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const unsigned long long hm = __shfl(mh, t_lo + j, 64) & ~__shfl(ml, t_lo + j, 64);
                pp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((hm >> lane) & 1ull) pp[j] = F[c * CHUNK + (t_lo + j) * TILE + lane];
            }
#pragma unroll 1
            for (int j = 0; j < PT; ++j) {  // (one copy of the body: the fetched points rotate through pp[0])
                const unsigned long long vm = __shfl(mv, t_lo + j, 64), hm = __shfl(mh, t_lo + j, 64);
                const unsigned long long lm = hm & ~vm;
                const unsigned long long late = __shfl(ml, t_lo + j, 64);  // (slots of the late table: counted, not touched)
                const float4 p = pp[0];
#pragma unroll
                for (int q = 0; q + 1 < PT; ++q) pp[q] = pp[q + 1];
                if (hm != 0ull) {
                    const bool mine = !((late >> lane) & 1ull);
                    const bool in = mine && ((vm >> lane) & 1ull), lv = mine && ((lm >> lane) & 1ull);
                    if (in || lv) {
                        if (in) {
                            const uint32_t rank = pv + __popcll(vm & lt);
                            const float4 e = xform(To2b, p);
                            voi_ego[rank] = e;
                            voi_key[rank] = bin_key(P, e.x, e.y, e.z, ctr);
                            voi_src[rank] = ph + __popcll(hm & lt);  // pre-step LOGICAL map index: the valid entries before it (== idx without holes)
                        } else {
                            // leaving: logical order of the new outskirts = [leaving (F order) | old outskirts]
                            const uint32_t lr = (ph - pv) + __popcll(lm & lt);
                            const uint32_t dst = o_new_begin + lr;
                            Oxy[dst] = make_float2(p.x, p.y);
                            Ozi[dst] = make_float2(p.z, p.w);
                            if (is_dynamic_label(p.w)) ++dyn_leave; else ++stat_leave;
                        }
                    }
                }
                pv += __popcll(vm);
                ph += __popcll(hm);
            }
        } else {
            for (int t = t_lo; t < t_hi; ++t) {
                const unsigned long long vm = __shfl(mv, t, 64), hm = __shfl(mh, t, 64);
                if (vm != 0ull) {
                    const uint32_t idx = (c - nFchunks + o_chunk0) * CHUNK + t * TILE + lane;
                    if ((vm >> lane) & 1ull) {
                        const float2 a = Oxy[idx], b = Ozi[idx];
                        const uint32_t rank = pv + __popcll(vm & lt);
                        const float4 e = xform(To2b, make_float4(a.x, a.y, b.x, b.y));
                        voi_ego[rank] = e;
                        voi_key[rank] = bin_key(P, e.x, e.y, e.z, ctr);
                        voi_src[rank] = ph + __popcll(hm & lt);  // (== points of the VoI-resident region + valid outskirts entries before it)
                        if (!ovm) reinterpret_cast<uint32_t *>(Oxy)[(size_t)idx * 2] = HOLE_BITS;  // tombstone
                        if (is_dynamic_label(b.y)) ++dyn_enter; else ++stat_enter;
                    }
                }
                pv += __popcll(vm);
                ph += __popcll(hm);
            }
        }
        // (the chunk's entering entries are tombstones now: its record counts the valid ones)
        if (!isF && ometa && lane == 0 && !ovm) ometa[c - nFchunks + o_chunk0].valid = ch - cv;
    }
    (void)voiF;
    (void)validF;
    // outskirts label counters (parse_dynamic_obj is maintained incrementally, OMU.cpp:294)
    dyn_leave = wave_sum(dyn_leave);
    stat_leave = wave_sum(stat_leave);
    dyn_enter = wave_sum(dyn_enter);
    stat_enter = wave_sum(stat_enter);
    if (lane == 0) {
        if (dyn_leave | dyn_enter) atomicAdd(&st->O_dynamic, (unsigned long long)dyn_leave - (unsigned long long)dyn_enter);
        if (stat_leave | stat_enter) atomicAdd(&st->O_static, (unsigned long long)stat_leave - (unsigned long long)stat_enter);
    }
}

// ================================================================================================
// Round 5, OVERLAPPED steps: the late half of step k + 1's gather.  Step k's per-bin launch and k_assemble_late have filled the slots of
// the late table (reverted bins' voxels and ground; holes where nothing came); k_voi_split / k_voi_gather of step k + 1, which ran BESIDE
// that, kept a place in VoI order for every data slot and a place in the outskirts' order for every leaving reservation.  Here every
// data slot that holds a point inside the VoI circle goes to its place (egocentric transform, R-POD key: exactly k_voi_gather's
// arithmetic); every other data slot's place gets the key of the dead bucket (B + 1), which the bucketing moves behind everything; a
// point outside the circle goes to the place its reservation holds in front of the outskirts (a bin without reservations cannot have such
// a point: srt4_body's radius test -- err 7 otherwise), unused reservations are tombstoned.  One workgroup per table entry (grid
// stride), one wavefront per 64-slot tile of its range; the launch's LAST workgroup ends step k (see StepEnd): everything that step wrote
// is visible at this kernel's boundary.
// ================================================================================================
__global__ __launch_bounds__(256) void k_late_gather(const float4 *__restrict__ F, const LateEnt *__restrict__ late, const DevState *__restrict__ prev,
                                                      float2 *__restrict__ Oxy, float2 *__restrict__ Ozi,
                                                      const unsigned long long *__restrict__ vmask, const unsigned long long *__restrict__ hmask,
                                                      const uint32_t *__restrict__ pvl, const uint32_t *__restrict__ phl,
                                                      const uint32_t *__restrict__ topv, const uint32_t *__restrict__ toph, double xc, double yc,
                                                      double r2, Xf To2b, DP P, DevState *st, Counters *ctr, const Counters *qctr,
                                                      float4 *__restrict__ voi_ego, uint32_t *__restrict__ voi_key, uint32_t *__restrict__ voi_src,
                                                      StepEnd se) {
    if (blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x == 0 && se.st) step_end_body(se.st, se.ctr, se.out, se.lab_slots, se.qctr, se.q_nvox, se.seq);
        return;
    }
    // (round 6: the three flags and the two extents in one round trip -- as a chain of short-circuit tests they were up to four)
    const uint32_t f_ov = st->ov_bad, f_e = ctr->err, f_q = qctr->err;
    const uint32_t n_late = prev->n_late;
    const uint32_t o_new_begin = st->o_new_begin;
    if (f_ov | f_e | f_q) return;  // (the step that finds this out runs its own passes / fails like any other)
    CHAIN_STAMP(9);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t lt = lanemask_lt();
    uint32_t dead = 0, ph_unused = 0, n_left = 0, dyn_leave = 0, stat_leave = 0, bad = 0;
    for (uint32_t e = blockIdx.x; e < n_late; e += gridDim.x - 1) {
        const LateEnt le = late[e];
        if (!le.ntotal) continue;
        const uint32_t T0 = le.start / TILE, T1 = (le.start + le.ntotal - 1) / TILE;
        for (uint32_t T = T0 + wave; T <= T1; T += nw) {
            const uint32_t c = T / CHUNK_TILES, t = T % CHUNK_TILES;
            const unsigned long long mv = lane < CHUNK_TILES ? vmask[(size_t)c * CHUNK_TILES + lane] : 0ull;
            const unsigned long long mh = lane < CHUNK_TILES ? hmask[(size_t)c * CHUNK_TILES + lane] : 0ull;
            uint32_t pv = pvl[c] + topv[c >> 10], ph = phl[c] + toph[c >> 10];
            const uint32_t a2 = wave_sum((lane < t) ? ((uint32_t)__popcll(mv) | ((uint32_t)__popcll(mh) << 16)) : 0u);
            pv += a2 & 0xFFFFu;
            ph += a2 >> 16;
            const unsigned long long vm = __shfl(mv, (int)t, 64), hm = __shfl(mh, (int)t, 64);
            const uint32_t pos = T * TILE + lane;
            const bool isdat = pos >= le.start && pos < le.start + le.ndata;
            const bool isph = pos >= le.start + le.ndata && pos < le.start + le.ntotal;
            if (isdat || isph) {
                const float4 p = F[isdat ? pos : pos - le.ndata];
                const bool point = __float_as_uint(p.x) != HOLE_BITS;
                // double dist_square = pow(pt.x - x_criterion, 2) + pow(pt.y - y_criterion, 2)  (OMU.cpp:394)
                const double dx = (double)p.x - xc, dy = (double)p.y - yc;
                const bool in = point && (dx * dx + dy * dy < r2);
                if (isdat) {
                    const uint32_t rank = pv + (uint32_t)__popcll(vm & lt);
                    if (in) {
                        const float4 eg = xform(To2b, p);
                        voi_ego[rank] = eg;
                        voi_key[rank] = bin_key(P, eg.x, eg.y, eg.z, ctr);
                        voi_src[rank] = ph + (uint32_t)__popcll(hm & lt);
                    } else {
                        voi_ego[rank] = make_float4(0.f, 0.f, 0.f, 0.f);
                        voi_key[rank] = (uint32_t)P.B + 1u;
                        voi_src[rank] = 0u;
                        ++dead;
                        if (point && le.ntotal == le.ndata) ++bad;  // a point left a VoI it could not leave
                    }
                } else {
                    const uint32_t lr = (ph - pv) + (uint32_t)__popcll((hm & ~vm) & lt);
                    const uint32_t dst = o_new_begin + lr;
                    if (point && !in) {
                        Oxy[dst] = make_float2(p.x, p.y);
                        Ozi[dst] = make_float2(p.z, p.w);
                        ++n_left;
                        if (is_dynamic_label(p.w)) ++dyn_leave; else ++stat_leave;
                    } else {
                        reinterpret_cast<uint32_t *>(Oxy)[(size_t)dst * 2] = HOLE_BITS;
                        ++ph_unused;
                    }
                }
            }
        }
    }
    dead = wave_sum(dead);
    ph_unused = wave_sum(ph_unused);
    n_left = wave_sum(n_left);
    dyn_leave = wave_sum(dyn_leave);
    stat_leave = wave_sum(stat_leave);
    bad = wave_sum(bad);
    if (lane == 0) {
        if (dead) atomicAdd(&st->n_voi_dead, dead);
        if (ph_unused) atomicAdd(&st->n_leaving, 0u - ph_unused);
        if (n_left) atomicAdd(&st->n_late_leaving, n_left);
        if (dyn_leave) atomicAdd(&st->O_dynamic, (unsigned long long)dyn_leave);
        if (stat_leave) atomicAdd(&st->O_static, (unsigned long long)stat_leave);
        if (bad) atomicMax(&ctr->err, 7u);
    }
}

// ... and its tombstones: the outskirts entries that entered the VoI of an overlapped step, erased once the step is certain (enqueued
// with the step itself, not ahead of it: see k_voi_gather).  One wavefront per outskirts chunk that holds VoI entries; the chunk's
// record counts what stays.
__global__ __launch_bounds__(256) void k_o_commit(float2 *__restrict__ Oxy, const unsigned long long *__restrict__ vmask,
                                                   const uint32_t *__restrict__ cinfo, const DevState *__restrict__ st, uint32_t capO_chunks,
                                                   OMeta *__restrict__ ometa, const Counters *__restrict__ ctr, const Counters *__restrict__ qctr) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    if (st->ov_bad) return;
    const uint32_t nFchunks = (st->nF + CHUNK - 1) / CHUNK, o_chunk0 = st->o_begin / CHUNK;
    CHAIN_STAMP(12);
    // (a step whose scan's voxelisation failed leaves no trace in the store -- its gather did nothing, see k_voi_gather: nor does this)
    if (ctr->err || qctr->err) return;
    const uint32_t nOchunks = capO_chunks - o_chunk0;
    for (uint32_t w = wid; w < nOchunks; w += nwaves) {
        const uint32_t c = nFchunks + w;
        const uint32_t ci = cinfo[c];
        const uint32_t cv = ci & 0xFFFFu, ch = (ci >> 16) & CINFO_HMASK;
        if (!cv) continue;
        const unsigned long long mv = lane < CHUNK_TILES ? vmask[(size_t)c * CHUNK_TILES + lane] : 0ull;
        for (int t = 0; t < CHUNK_TILES; ++t) {
            const unsigned long long vm = __shfl(mv, t, 64);
            if ((vm >> lane) & 1ull) reinterpret_cast<uint32_t *>(Oxy)[(size_t)((o_chunk0 + w) * CHUNK + t * TILE + lane) * 2] = HOLE_BITS;
        }
        if (ometa && lane == 0) ometa[o_chunk0 + w].valid = ch - cv;
    }
}

// ================================================================================================
// generic two-level exclusive scan of uint32 (n read from device memory when n_dev != nullptr)
// ================================================================================================
__global__ __launch_bounds__(256) void k_scan_local(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                     uint32_t *__restrict__ tops, uint32_t n_host, const uint32_t *n_dev) {
    __shared__ uint32_t sm[40];
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    if (blockIdx.x * 1024 >= n && blockIdx.x != 0) return;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0u;
        s += v[j];
    }
    uint32_t tot;
    uint32_t p = block_excl_scan(s, sm, tot);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < n) out[base + j] = p;
        p += v[j];
    }
    if (threadIdx.x == 0) tops[blockIdx.x] = tot;
}
// single block; tops[ntop] receives the grand total
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t *__restrict__ tops, uint32_t n_host, const uint32_t *n_dev,
                                                    uint32_t *total_out) {
    __shared__ uint32_t sm[40];
    __shared__ uint32_t carry;
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t ntop = (n + 1023) / 1024;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ntop; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < ntop ? tops[i] : 0u;
        uint32_t t;
        const uint32_t p = block_excl_scan(v, sm, t);
        const uint32_t c0 = carry;
        if (i < ntop) tops[i] = c0 + p;
        __syncthreads();
        if (threadIdx.x == 0) carry = c0 + t;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tops[ntop] = carry;
        if (total_out) *total_out = carry;
    }
}

// ================================================================================================
// stable LSD radix sort (8-bit digits) of uint32 keys with implicit/explicit uint32 payload.
// Tile = 2048 keys per 256-thread block; each wavefront owns a contiguous 512-key strip so that
// (block, wave, round, lane) order == index order (stability).
// ================================================================================================
static constexpr int RTILE = 2048;
// the lanes of the wavefront that hold the same key as this one (0 for an invalid lane).  Round 4: one round per DISTINCT key of the
// wavefront -- pick the lowest lane not yet matched, read its key (v_readlane, the lane number is wave-uniform), one compare is the
// group's mask -- instead of one ballot and a 64-bit select per key BIT: points that are neighbours in the map are neighbours in the
// grid, a wavefront of 64 consecutive ones holds a handful of distinct bins (the bit-wise form: ~120 instructions per 64 keys of a
// 12-bit grid, 6 us of a scatter workgroup's 13; this one: ~8 per distinct key).  `max_rounds` bounds the loop; what is left then
// (a wavefront with more distinct keys than that) is matched bit by bit.
__device__ __forceinline__ uint64_t match_any(uint32_t k, bool valid, int bits) {
    uint64_t todo = __ballot(valid), mine = 0ull;
    int rounds = 0;
    while (todo != 0ull && rounds < 12) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t kk = __builtin_amdgcn_readlane(k, leader);
        const uint64_t m = __ballot(valid && k == kk);
        if (k == kk) mine = m;
        todo &= ~m;
        ++rounds;
    }
    if (todo != 0ull) {  // (many distinct keys: the rest bit by bit, among the lanes that are still unmatched)
        const bool left = (todo >> (threadIdx.x & 63u)) & 1ull;
        uint64_t p = todo;
        for (int b = 0; b < bits; ++b) {
            const bool bit = (k >> b) & 1u;
            const uint64_t m = __ballot(left && bit);
            p &= bit ? m : ~m;
        }
        if (left) mine = p;
    }
    return valid ? mine : 0ull;
}
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

__global__ __launch_bounds__(256) void k_radix_hist(const uint32_t *__restrict__ keys, uint32_t n_host, const uint32_t *n_dev,
                                                     int shift, uint32_t *__restrict__ hist /* [256][nblk] */, uint32_t *nhist_out) {
    __shared__ uint32_t wcnt[4][256];
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t nblk = (n + RTILE - 1) / RTILE;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nhist_out) *nhist_out = 256u * nblk;  // number of histogram entries to scan
    if (blockIdx.x >= nblk) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint64_t lt = lanemask_lt();
    for (uint32_t i = tid; i < 4 * 256; i += 256) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t strip = blockIdx.x * RTILE + wave * 512;
    for (int r = 0; r < 8; ++r) {
        const uint32_t i = strip + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = valid ? ((keys[i] >> shift) & 0xFFu) : 0u;
        const uint64_t peers = match_digit(d, valid);
        if (valid && (peers & lt) == 0) wcnt[wave][d] += __popcll(peers);
        esort::wave_sync_lds();
    }
    __syncthreads();
    const uint32_t tot = wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
    hist[(size_t)tid * nblk + blockIdx.x] = tot;
}

// writes *cnt_out = 256 * nblk (the number of histogram entries to scan)

__global__ __launch_bounds__(256) void k_radix_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                        uint32_t n_host, const uint32_t *n_dev, int shift,
                                                        const uint32_t *__restrict__ hist_local, const uint32_t *__restrict__ hist_tops,
                                                        uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t wcnt[4][256];
    __shared__ uint32_t wbase[4][256];
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t nblk = (n + RTILE - 1) / RTILE;
    if (blockIdx.x >= nblk) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint64_t lt = lanemask_lt();
    for (uint32_t i = tid; i < 4 * 256; i += 256) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t strip = blockIdx.x * RTILE + wave * 512;
    uint32_t k[8], v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint32_t i = strip + r * 64 + lane;
        const bool valid = i < n;
        k[r] = valid ? keys[i] : 0u;
        v[r] = valid ? (vals ? vals[i] : i) : 0u;
        const uint32_t d = (k[r] >> shift) & 0xFFu;
        const uint64_t peers = match_digit(d, valid);
        if (valid && (peers & lt) == 0) wcnt[wave][d] += __popcll(peers);
        esort::wave_sync_lds();
    }
    __syncthreads();
    {
        const size_t e = (size_t)tid * nblk + blockIdx.x;  // digit = tid
        uint32_t b = hist_local[e] + hist_tops[e >> 10];
        for (int w = 0; w < 4; ++w) {
            wbase[w][tid] = b;
            b += wcnt[w][tid];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint32_t i = strip + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = (k[r] >> shift) & 0xFFu;
        const uint64_t peers = match_digit(d, valid);
        if (valid) {
            const uint32_t pos = wbase[wave][d] + __popcll(peers & lt);
            keys_out[pos] = k[r];
            vals_out[pos] = v[r];
        }
        esort::wave_sync_lds();
        if (valid && (peers & lt) == 0) wbase[wave][d] += __popcll(peers);
        esort::wave_sync_lds();
    }
}


// gather points (and optionally a uint32 side array) into sorted order
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ src, const uint32_t *__restrict__ src_aux,
                                                 const uint32_t *__restrict__ perm, uint32_t n_host, const uint32_t *n_dev,
                                                 float4 *__restrict__ dst, uint32_t *__restrict__ dst_aux) {
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = perm[i];
    dst[i] = src[p];
    if (src_aux) dst_aux[i] = src_aux[p];
}

// off[k] = first sorted position with key >= k, for k in [0, nkeys]; off[nkeys] = n  (nkeys = B+1 buckets -> B+2 entries)
// ---- stable one-digit counting sort of the query voxels by R-POD key ------------------------------
// The key range (B + 1 buckets) fits an LDS table, so the bucketing is one histogram / scan / scatter round with the
// bucket offsets (k_bin_offsets' product) and the point gather (k_gather) folded in: 3 launches instead of 12.
// Tile-major inside a bucket + index order inside a tile == the stable order of the LSD radix path.
static constexpr uint32_t QB_TILE = 1024;
static constexpr uint32_t QB_NB_MAX = 12288;  // buckets (B + 1) the LDS table holds

__device__ __forceinline__ void qb_hist_body(const uint32_t *__restrict__ keys, uint32_t n_host, const uint32_t *n_dev, uint32_t nb,
                                                   uint32_t *__restrict__ hist /* [tile][nb] */, uint32_t *__restrict__ tot /* [nb] */) {
    __shared__ uint32_t cnt[QB_NB_MAX];
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t ntile = (n + QB_TILE - 1) / QB_TILE;
    if (blockIdx.x >= ntile) return;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * QB_TILE + threadIdx.x;
    if (i < n) atomicAdd(&cnt[min(keys[i], nb - 1)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const uint32_t c = cnt[b];
        hist[(size_t)blockIdx.x * nb + b] = c;
        if (c) atomicAdd(&tot[b], c);  // bucket totals (integer: order-independent); zeroed by k_query_begin / k_chunk_scan_top
    }
}
__global__ __launch_bounds__(1024) void k_qb_hist(const uint32_t *__restrict__ keys, uint32_t n_host, const uint32_t *n_dev, uint32_t nb,
                                                   uint32_t *__restrict__ hist /* [tile][nb] */, uint32_t *__restrict__ tot /* [nb] */) { qb_hist_body(keys, n_host, n_dev, nb, hist, tot); }

// single workgroup, one thread per bucket: exclusive scan of the bucket totals -> off[b]; off[nb] = n
__device__ __forceinline__ void qb_scan_body(const uint32_t *__restrict__ tot, uint32_t nb, uint32_t *__restrict__ off) {
    __shared__ uint32_t sm[40];
    constexpr int ROUNDS = QB_NB_MAX / 1024;
    uint32_t s[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t b = r * 1024 + threadIdx.x;
        s[r] = b < nb ? tot[b] : 0u;
    }
    uint32_t carry = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if ((uint32_t)r * 1024 >= nb) break;  // block-uniform
        const uint32_t b = r * 1024 + threadIdx.x;
        uint32_t t;
        const uint32_t ex = block_excl_scan(s[r], sm, t);
        if (b < nb) off[b] = carry + ex;
        carry += t;
    }
    if (threadIdx.x == 0) off[nb] = carry;
}
__global__ __launch_bounds__(1024) void k_qb_scan(const uint32_t *__restrict__ tot, uint32_t nb, uint32_t *__restrict__ off) { qb_scan_body(tot, nb, off); }

__device__ __forceinline__ void qb_scatter_body(const uint32_t *__restrict__ keys, const float4 *__restrict__ src, uint32_t n_host,
                                                      const uint32_t *n_dev, uint32_t nb, int bits, const uint32_t *__restrict__ hist,
                                                      const uint32_t *__restrict__ off, float4 *__restrict__ dst) {
    __shared__ uint32_t cnt[QB_NB_MAX];
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t ntile = (n + QB_TILE - 1) / QB_TILE;
    if (blockIdx.x >= ntile) return;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * QB_TILE + threadIdx.x;
    const bool valid = i < n;
    const uint32_t k = valid ? min(keys[i], nb - 1) : 0u;
    const uint32_t wave = threadIdx.x >> 6;
    // start of (bucket k, this tile): bucket offset + the earlier tiles' share of the bucket
    uint32_t pre = 0;
    if (valid) {
        uint32_t a0 = off[k], a1 = 0, a2 = 0, a3 = 0;
        uint32_t t = 0;
        for (; t + 4 <= blockIdx.x; t += 4) {  // four independent loads in flight
            a0 += hist[(size_t)t * nb + k];
            a1 += hist[(size_t)(t + 1) * nb + k];
            a2 += hist[(size_t)(t + 2) * nb + k];
            a3 += hist[(size_t)(t + 3) * nb + k];
        }
        for (; t < blockIdx.x; ++t) a0 += hist[(size_t)t * nb + k];
        pre = (a0 + a1) + (a2 + a3);
    }
    // lanes of this wave that hold the same key
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool bit = (k >> b) & 1u;
        const uint64_t m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
    }
    const uint64_t lt = lanemask_lt();
    uint32_t r = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {  // waves take their turn in index order
        if (wave == w && valid) {
            r = cnt[k] + (uint32_t)__popcll(peers & lt);
            esort::wave_sync_lds();
            if ((peers >> (threadIdx.x & 63u)) >> 1 == 0) cnt[k] += (uint32_t)__popcll(peers);  // highest lane of the group
        }
        __syncthreads();
    }
    if (valid) dst[pre + r] = src[i];
}
__global__ __launch_bounds__(1024) void k_qb_scatter(const uint32_t *__restrict__ keys, const float4 *__restrict__ src, uint32_t n_host,
                                                      const uint32_t *n_dev, uint32_t nb, int bits, const uint32_t *__restrict__ hist,
                                                      const uint32_t *__restrict__ off, float4 *__restrict__ dst) { qb_scatter_body(keys, src, n_host, n_dev, nb, bits, hist, off, dst); }

__device__ __forceinline__ uint32_t fkey_ord(float f) {  // total-order key for float min/max atomics
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}
// ---- the same one-digit counting sort for the (much longer) VoI list of the map: tiles of 4096 keys keep the
// [tile][bucket] table short (~200 rows for a 0.8 M-point VoI), the per-tile start of a bucket becomes a table look-up
// through a column scan (one wavefront per bucket over the tiles), and the scatter carries the point, its pre-step
// source index and the key along (k_gather folded in)
#ifndef ERASOR_MB_TILE
#define ERASOR_MB_TILE 4096
#endif
static constexpr uint32_t MB_TILE = ERASOR_MB_TILE;
// (round 4: the table's rows are padded to MB_PAD buckets -- `nbs` words per tile -- so that the column scan can take 16 consecutive buckets
// of a tile as ONE fully used 64-byte line)
static constexpr uint32_t MB_PAD = 16;
__host__ __device__ __forceinline__ uint32_t mb_row_stride(uint32_t nb) { return (nb + MB_PAD - 1) / MB_PAD * MB_PAD; }
__global__ __launch_bounds__(1024) void k_mb_hist(const uint32_t *__restrict__ keys, uint32_t n_host, const uint32_t *n_dev, uint32_t nb,
                                                   uint32_t *__restrict__ hist /* [tile][nbs] */, uint32_t *__restrict__ tot /* [nb] */) {
    const uint32_t nbs = mb_row_stride(nb);
    __shared__ uint32_t cnt[QB_NB_MAX];
    CHAIN_STAMP(10);
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t ntile = (n + MB_TILE - 1) / MB_TILE;
    // (the grid is sized from the previous step's VoI, not from the whole map: tiles are walked with a grid stride)
    for (uint32_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) cnt[b] = 0;
        __syncthreads();
        const uint32_t i0 = tile * MB_TILE + threadIdx.x;
        uint32_t k[MB_TILE / 1024];
#pragma unroll
        for (uint32_t r = 0; r < MB_TILE / 1024; ++r) k[r] = i0 + r * 1024 < n ? keys[i0 + r * 1024] : 0xFFFFFFFFu;
        // one LDS add per DISTINCT key of a wavefront (round 4): the VoI arrives in the previous step's bin order, neighbours share their
        // bin, and 64 adds to one counter are served one after the other (config 4: 17.7 us for 14 MB of keys)
        const int bits = 32 - __builtin_clz(nb | 1u);
        const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
        for (uint32_t r = 0; r < MB_TILE / 1024; ++r) {
            const bool valid = k[r] != 0xFFFFFFFFu;
            const uint32_t key = min(k[r], nb - 1);
            const uint64_t peers = match_any(key, valid, bits);
            if (valid && lane == (uint32_t)__builtin_ctzll(peers)) atomicAdd(&cnt[key], (uint32_t)__popcll(peers));
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
            const uint32_t c = cnt[b];
            hist[(size_t)tile * nbs + b] = c;
            if (c) atomicAdd(&tot[b], c);
        }
        __syncthreads();
    }
}

// (the bucket offsets -- exclusive scan of the bucket totals -- are recomputed by every workgroup for its own buckets: cheaper than a
// launch of its own on the main stream's dependency chain; workgroup 0 publishes them)
// Round 4: a workgroup takes MB_PAD = 16 CONSECUTIVE buckets of every tile: a wavefront's load covers four tiles x one 64-byte line each,
// every fetched byte is used (round 3 walked a column with one 4-byte word per cache line: 18.5 MB of traffic for a 1.7 MB table on
// config 2, 88.7 MB on config 4).  The 256 x 16 block is scanned down its columns in LDS (rows padded to 17 words: conflict-free), four
// columns per wavefront, 64 tiles per DPP scan, and goes back the way it came.
__global__ __launch_bounds__(256) void k_mb_colscan(uint32_t *__restrict__ hist /* [tile][nbs] -> start of (bucket, tile) */, uint32_t n_host,
                                                     const uint32_t *n_dev, uint32_t nb, const uint32_t *__restrict__ tot,
                                                     uint32_t *__restrict__ off_out) {
    __shared__ uint32_t sm[40];
    __shared__ uint32_t s_off[MB_PAD];
    __shared__ uint32_t s_blk[256][MB_PAD + 1];
    const uint32_t nbs = mb_row_stride(nb);
    CHAIN_STAMP(11);
    const uint32_t b0 = blockIdx.x * MB_PAD;  // first of this workgroup's buckets
    {
        const uint32_t per = (nb + blockDim.x - 1) / blockDim.x;
        const uint32_t lo = min(threadIdx.x * per, nb), hi = min(lo + per, nb);
        uint32_t s_ = 0;
        for (uint32_t i = lo; i < hi; ++i) s_ += tot[i];
        uint32_t t;
        uint32_t run = block_excl_scan(s_, sm, t);
        for (uint32_t i = lo; i < hi; ++i) {
            if (blockIdx.x == 0) off_out[i] = run;
            if (i >= b0 && i < b0 + MB_PAD) s_off[i - b0] = run;
            run += tot[i];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) off_out[nb] = t;
        __syncthreads();
    }
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t ntile = (n + MB_TILE - 1) / MB_TILE;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t r0 = threadIdx.x >> 4, c = threadIdx.x & 15u;  // this thread's rows r0, r0 + 16, ... of the block, column c
    uint32_t run[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) run[j] = (b0 + wave * 4 + j < nb) ? s_off[wave * 4 + j] : 0u;
    for (uint32_t t0 = 0; t0 < ntile; t0 += 256) {
        uint32_t v[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {  // (all sixteen loads in flight)
            const uint32_t t = t0 + r0 + 16 * k;
            v[k] = t < ntile ? hist[(size_t)t * nbs + b0 + c] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) s_blk[r0 + 16 * k][c] = v[k];
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t col = wave * 4 + j;
#pragma unroll
            for (uint32_t rr = 0; rr < 4; ++rr) {
                const uint32_t row = rr * 64 + lane;
                const uint32_t x = s_blk[row][col];
                const uint32_t inc = esort::wave_incl_scan(x);  // (DPP row shifts)
                s_blk[row][col] = run[j] + inc - x;
                run[j] += __builtin_amdgcn_readlane(inc, 63);
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t t = t0 + r0 + 16 * k;
            if (t < ntile && b0 + c < nb) hist[(size_t)t * nbs + b0 + c] = s_blk[r0 + 16 * k][c];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_mb_scatter(const uint32_t *__restrict__ keys, const float4 *__restrict__ src,
                                                      const uint32_t *__restrict__ src_aux, uint32_t n_host, const uint32_t *n_dev, uint32_t nb,
                                                      int bits, const uint32_t *__restrict__ base /* scanned hist */, float4 *__restrict__ dst,
                                                      uint32_t *__restrict__ dst_aux, uint32_t *__restrict__ dst_keys) {
    __shared__ uint32_t cnt[QB_NB_MAX];
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t ntile = (n + MB_TILE - 1) / MB_TILE;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    for (uint32_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    __syncthreads();
    // cnt[k] starts at the tile's first slot of bucket k and advances as the tile's keys are placed, in index order
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) cnt[b] = base[(size_t)tile * mb_row_stride(nb) + b];
    __syncthreads();
    for (uint32_t r = 0; r < MB_TILE / 1024; ++r) {
        const uint32_t i = tile * MB_TILE + r * 1024 + threadIdx.x;
        const bool valid = i < n;
        const uint32_t k = valid ? min(keys[i], nb - 1) : 0u;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t aux = 0;
        if (valid) {  // fetched before the ranking turns: the loads overlap them
            p = src[i];
            aux = src_aux[i];
        }
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            const bool bit = (k >> b) & 1u;
            const uint64_t m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        uint32_t d = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {  // waves take their turn in index order
            if (wave == w && valid) {
                d = cnt[k] + (uint32_t)__popcll(peers & lt);
                esort::wave_sync_lds();
                if ((peers >> (threadIdx.x & 63u)) >> 1 == 0) cnt[k] += (uint32_t)__popcll(peers);  // highest lane of the group
            }
            __syncthreads();
        }
        if (valid) {
            dst[d] = p;
            dst_aux[d] = aux;
            dst_keys[d] = k;
        }
    }
    }
}

// Scatter for R-POD grids of up to 4096 buckets: every wavefront owns 512 CONSECUTIVE keys of the tile, so index order is
// wavefront-major and the ranks need three block barriers instead of one per wavefront turn: (1) private per-wavefront
// bucket counts, (2) their prefix over the wavefronts (tile-relative, 16 bit), (3) each wavefront ranks its own keys.
static constexpr uint32_t MBW_NB_MAX = 4096;
static constexpr uint32_t MBW_NB_SMALL = 2176;  // round 4: up to this many buckets a workgroup's tables take 78 KB -- TWO workgroups per compute unit
// (NBC: bucket capacity of the LDS tables.  The usual grids -- 20 x 108 + 1 = 2161 buckets -- fit the small instance, whose two
// workgroups per compute unit halve the rounds a dense map's ~800 tiles need on 256 compute units.)
template <uint32_t NBC>
__global__ __launch_bounds__(1024, NBC <= MBW_NB_SMALL ? 8 : 4) void k_mb_scatter_w(const uint32_t *__restrict__ keys, const float4 *__restrict__ src,
                                                        const uint32_t *__restrict__ src_aux, uint32_t n_host, const uint32_t *n_dev, uint32_t nb,
                                                        int bits, const uint32_t *__restrict__ base /* scanned hist */, float4 *__restrict__ dst,
                                                        uint32_t *__restrict__ dst_aux, uint32_t *__restrict__ dst_keys,
                                                        unsigned long long *dbg = nullptr) {
    __shared__ uint16_t wcnt[16][NBC];
    __shared__ uint32_t sbase[NBC];
    CHAIN_STAMP(0);
#define SC_STAMP(i) do { if (dbg && blockIdx.x == 3 && threadIdx.x == 0) dbg[72 + (i)] = wall_clock64(); } while (0)
    SC_STAMP(0);
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t ntile = (n + MB_TILE - 1) / MB_TILE;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t lt = lanemask_lt();
    constexpr uint32_t R = MB_TILE / 16 / 64;  // 64-key strips per wavefront
    for (uint32_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) sbase[b] = base[(size_t)tile * mb_row_stride(nb) + b];
    {   // (only the columns in use: rows are NBC apart)
        const uint32_t nb2 = (nb + 1) / 2;
        for (uint32_t i = threadIdx.x; i < 16 * nb2; i += blockDim.x) reinterpret_cast<uint32_t *>(&wcnt[i / nb2][0])[i % nb2] = 0u;
    }
    __syncthreads();
    SC_STAMP(1);
    const uint32_t i0 = tile * MB_TILE + wave * (MB_TILE / 16) + lane;
    uint32_t k[R];
    uint64_t peers[R];
#pragma unroll
    for (uint32_t r = 0; r < R; ++r) {  // (all key loads in flight: the counting pass used to wait for them one by one)
        const uint32_t i = i0 + r * 64;
        k[r] = i < n ? min(keys[i], nb - 1) : 0u;
    }
#pragma unroll
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t i = i0 + r * 64;
        const bool valid = i < n;
        const uint64_t p = match_any(k[r], valid, bits);
        peers[r] = p;
        if (valid && (p >> lane) >> 1 == 0) wcnt[wave][k[r]] += (uint16_t)__popcll(p);  // highest lane of the group
        esort::wave_sync_lds();
    }
    __syncthreads();
    SC_STAMP(2);
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {  // exclusive prefix over the wavefronts (<= 8192: 16 bit)
        uint32_t run = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) {
            const uint32_t c = wcnt[w][b];
            wcnt[w][b] = (uint16_t)run;
            run += c;
        }
    }
    __syncthreads();
    SC_STAMP(3);
#pragma unroll
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t i = i0 + r * 64;
        if (i < n) {
            const uint32_t d = sbase[k[r]] + wcnt[wave][k[r]] + (uint32_t)__popcll(peers[r] & lt);
            dst[d] = src[i];
            dst_aux[d] = src_aux[i];
            dst_keys[d] = k[r];
        }
        esort::wave_sync_lds();
        if (i < n && (peers[r] >> lane) >> 1 == 0) wcnt[wave][k[r]] += (uint16_t)__popcll(peers[r]);
        esort::wave_sync_lds();
    }
    SC_STAMP(4);
    }
#undef SC_STAMP
}

__global__ __launch_bounds__(256) void k_bin_offsets(const uint32_t *__restrict__ skeys, uint32_t n_host, const uint32_t *n_dev,
                                                      uint32_t nbuckets, uint32_t *__restrict__ off) {
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) {
        if (i <= nbuckets) off[i] = 0;
        return;
    }
    if (i > n) return;
    if (i == n) {
        for (uint32_t b = skeys[n - 1] + 1; b <= nbuckets; ++b) off[b] = n;
        return;
    }
    const uint32_t kcur = skeys[i];
    const uint32_t kprev = i == 0 ? 0xFFFFFFFFu : skeys[i - 1];
    if (i == 0) {
        for (uint32_t b = 0; b <= kcur; ++b) off[b] = 0;
    } else if (kcur != kprev) {
        for (uint32_t b = kprev + 1; b <= kcur; ++b) off[b] = i;
    }
}

// per-bin pseudo-occupancy descriptor (erasor.cpp:87-98): count, min z, max z.  One wavefront per bin.
__device__ __forceinline__ void bin_stats_body(const float4 *__restrict__ spts, const uint32_t *__restrict__ off, uint32_t B,
                                                    uint32_t *__restrict__ cnt, float *__restrict__ minz, float *__restrict__ maxz) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= B) return;
    const uint32_t s = off[b], e = off[b + 1];
    float mn = __int_as_float(0x7F800000), mx = __int_as_float(0xFF800000);
    // bins hold up to a few thousand points: four loads in flight per lane (one load per round made the largest bin's
    // ~50 dependent rounds the kernel's length); the lane's compare order is unchanged
    uint32_t i = s + lane;
    for (; i + 192 < e; i += 256) {
        const float z0 = spts[i].z, z1 = spts[i + 64].z, z2 = spts[i + 128].z, z3 = spts[i + 192].z;
        mn = z0 < mn ? z0 : mn;
        mx = z0 > mx ? z0 : mx;
        mn = z1 < mn ? z1 : mn;
        mx = z1 > mx ? z1 : mx;
        mn = z2 < mn ? z2 : mn;
        mx = z2 > mx ? z2 : mx;
        mn = z3 < mn ? z3 : mn;
        mx = z3 > mx ? z3 : mx;
    }
    for (; i < e; i += 64) {
        const float z = spts[i].z;
        mn = z < mn ? z : mn;
        mx = z > mx ? z : mx;
    }
    mn = wave_min_f(mn);  // (z values are finite: the scan was checked for NaN / Inf, so fminf == the compare chain)
    mx = wave_max_f(mx);
    if (lane == 0) {
        cnt[b] = e - s;
        minz[b] = mn;
        maxz[b] = mx;
    }
}
__global__ __launch_bounds__(256) void k_bin_stats(const float4 *__restrict__ spts, const uint32_t *__restrict__ off, uint32_t B,
                                                    uint32_t *__restrict__ cnt, float *__restrict__ minz, float *__restrict__ maxz) { bin_stats_body(spts, off, B, cnt, minz, maxz); }

// ================================================================================================
// query-scan voxelisation: PCL 1.8 VoxelGrid + label-preserving 1-NN (utils.cpp:80-114; OMU.cpp:238)
// ================================================================================================

// getMinMax3D (dense): plain min/max; -0.0/+0.0 order is irrelevant downstream (only products/floors of it)
__device__ __forceinline__ void bbox_body(const float4 *__restrict__ pts, uint32_t n, uint32_t *bb) {
    __shared__ uint32_t sm[6];
    if (threadIdx.x < 3) sm[threadIdx.x] = 0xFFFFFFFFu;
    if (threadIdx.x >= 3 && threadIdx.x < 6) sm[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        const uint32_t k[3] = {fkey_ord(p.x), fkey_ord(p.y), fkey_ord(p.z)};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = k[a] < mn[a] ? k[a] : mn[a];
            mx[a] = k[a] > mx[a] ? k[a] : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_minmax_u<false>(mn[a]);  // (DPP row shifts: a shuffle ladder is six dependent LDS-crossbar round trips per value)
        mx[a] = wave_minmax_u<true>(mx[a]);
        if ((threadIdx.x & 63u) == 0) {
            atomicMin(&sm[a], mn[a]);
            atomicMax(&sm[3 + a], mx[a]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bb[threadIdx.x], sm[threadIdx.x]);
    if (threadIdx.x >= 3 && threadIdx.x < 6) atomicMax(&bb[threadIdx.x], sm[threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_bbox(const float4 *__restrict__ pts, uint32_t n, uint32_t *bb) { bbox_body(pts, n, bb); }

struct VoxGrid {  // PCL VoxelGrid geometry of one cloud
    int32_t min_b[3], div_b[3];
    float inv_leaf;
    int32_t overflow;
};
__device__ __forceinline__ VoxGrid vox_grid_from_bbox(const float mn[3], const float mx[3], float leaf) {
    VoxGrid g;
    const float inv = 1.0f / leaf;
    g.inv_leaf = inv;
    long long d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = (long long)((mx[a] - mn[a]) * inv) + 1;
    // PCL: (dx*dy*dz) > INT_MAX in int64 -- with saturation, so that extents whose int64 product itself would wrap (undefined
    // behaviour in the reference) report the overflow they stand for instead of a garbage grid
    bool too_many = false;
    long long prod = 1;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (d[a] <= 0 || d[a] > 2147483647LL) too_many = true;
        if (!too_many) {
            prod *= d[a];
            if (prod > 2147483647LL) too_many = true;
        }
    }
    g.overflow = too_many ? 1 : 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g.min_b[a] = (int)floorf(mn[a] * inv);
        const int maxb = (int)floorf(mx[a] * inv);
        g.div_b[a] = maxb - g.min_b[a] + 1;
    }
    return g;
}
__device__ __forceinline__ uint32_t vox_index(const VoxGrid &g, float x, float y, float z) {
    const int i0 = (int)(floorf(x * g.inv_leaf) - (float)g.min_b[0]);
    const int i1 = (int)(floorf(y * g.inv_leaf) - (float)g.min_b[1]);
    const int i2 = (int)(floorf(z * g.inv_leaf) - (float)g.min_b[2]);
    return (uint32_t)(i0 + i1 * g.div_b[0] + i2 * (g.div_b[0] * g.div_b[1]));
}

__device__ __forceinline__ void voxel_keys_body(const float4 *__restrict__ pts, uint32_t n, const uint32_t *__restrict__ bb,
                                                float leaf, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                VoxGrid *gout, Counters *ctr, uint32_t *__restrict__ hkey, uint32_t hsize) {
    // (side job) empty the voxel hash table that k_centroids fills and k_query_nn probes
    for (uint32_t sl = blockIdx.x * blockDim.x + threadIdx.x; sl < hsize; sl += gridDim.x * blockDim.x) hkey[sl] = 0xFFFFFFFFu;
    const float mn[3] = {fkey_inv(bb[0]), fkey_inv(bb[1]), fkey_inv(bb[2])};
    const float mx[3] = {fkey_inv(bb[3]), fkey_inv(bb[4]), fkey_inv(bb[5])};
    VoxGrid g = vox_grid_from_bbox(mn, mx, leaf);
    // A NaN / Inf coordinate anywhere in the cloud shows up in its bounding box (NaN keys order beyond +-Inf).  PCL would
    // drop such points only for clouds flagged !is_dense; here the cloud is refused -- with garbage voxel indices the
    // neighbour search below has no bound on its shells.
    bool finite = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) finite = finite && isfinite(mn[a]) && isfinite(mx[a]);
    if (n && !finite) g.overflow = 2;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        *gout = g;
        if (g.overflow && n) {
            atomicAdd(&ctr->n_voxel_overflow, 1u);
            ctr->err = g.overflow == 2 ? 3 : 2;  // 2: VoxelGrid pass-through (reference returns the input unvoxelised); 3: non-finite input
        }
    }
    if (i >= n) return;
    const float4 p = pts[i];
    keys[i] = g.overflow ? 0u : vox_index(g, p.x, p.y, p.z);
    vals[i] = i;
}
__global__ __launch_bounds__(256) void k_voxel_keys(const float4 *__restrict__ pts, uint32_t n, const uint32_t *__restrict__ bb,
                                                     float leaf, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                     VoxGrid *gout, Counters *ctr, uint32_t *__restrict__ hkey, uint32_t hsize) {
    voxel_keys_body(pts, n, bb, leaf, keys, vals, gout, ctr, hkey, hsize);
}

// ---- VoxelGrid index overflow: PCL warns and returns the input cloud unchanged (utils.cpp:88-91), after which the label
// search of voxelize_preserving_labels gives every point the label of its nearest input point: itself, or -- distance 0,
// "lowest index wins" -- the first exact duplicate.  Duplicates are found through a stable radix sort of a hash of the
// coordinates: within a bucket the points come in index order, the first one with equal coordinates is the answer.
__device__ __forceinline__ uint32_t xyz_hash(float x, float y, float z) {
    // -0.0 == +0.0 under the float comparison that defines "duplicate": one canonical zero
    const uint32_t a = x == 0.f ? 0u : __float_as_uint(x), b = y == 0.f ? 0u : __float_as_uint(y), c = z == 0.f ? 0u : __float_as_uint(z);
    uint32_t h = a * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    return h;
}
__global__ __launch_bounds__(256) void k_xyz_hash_keys(const float4 *__restrict__ pts, uint32_t n, int bits, uint32_t *__restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    keys[i] = xyz_hash(p.x, p.y, p.z) >> (32 - bits);
}
// the pass-through chain's own check that VoxelGrid WOULD overflow for this cloud and leaf (the host only guesses it from
// the previous scan): err 4 = "it would not", the caller falls back to the voxelising chain; err 3 = non-finite input
__global__ void k_passthrough_check(const uint32_t *__restrict__ bb, uint32_t n, float leaf, VoxGrid *gout, Counters *ctr) {
    if (threadIdx.x || blockIdx.x) return;
    const float mn[3] = {fkey_inv(bb[0]), fkey_inv(bb[1]), fkey_inv(bb[2])};
    const float mx[3] = {fkey_inv(bb[3]), fkey_inv(bb[4]), fkey_inv(bb[5])};
    VoxGrid g = vox_grid_from_bbox(mn, mx, leaf);
    bool finite = true;
    for (int a = 0; a < 3; ++a) finite = finite && isfinite(mn[a]) && isfinite(mx[a]);
    if (n && !finite) g.overflow = 2;
    *gout = g;
    if (!n) return;
    if (g.overflow == 2) ctr->err = 3;
    else if (!g.overflow) ctr->err = 4;
    else atomicAdd(&ctr->n_voxel_overflow, 1u);
}
// out[i] = T * (x, y, z, label of the first point with the same coordinates); optionally the R-POD key of the result
__global__ __launch_bounds__(256) void k_dup_label_passthrough(const float4 *__restrict__ pts, uint32_t n, const uint32_t *__restrict__ skeys,
                                                                const uint32_t *__restrict__ sperm, int bits, Xf T, int apply_T, DP P,
                                                                Counters *ctr, float4 *__restrict__ out, uint32_t *__restrict__ qkey) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const uint32_t key = xyz_hash(p.x, p.y, p.z) >> (32 - bits);
    uint32_t lo = 0, hi = n;
    while (lo < hi) {  // first entry of the bucket
        const uint32_t mid = (lo + hi) >> 1;
        if (skeys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    float label = p.w;
    for (uint32_t j = lo; j < n && skeys[j] == key; ++j) {
        const uint32_t c = sperm[j];
        if (c >= i) break;  // stable: index order inside the bucket; nothing below i matched
        const float4 q = pts[c];
        if (q.x == p.x && q.y == p.y && q.z == p.z) {
            label = q.w;
            break;
        }
    }
    float4 o = make_float4(p.x, p.y, p.z, label);
    if (apply_T) o = xform(T, o);
    out[i] = o;
    if (qkey) qkey[i] = bin_key(P, o.x, o.y, o.z, ctr);
}

// ---- exact std::sort of the (idx, point) pairs, global memory -----------------------------------
// Exact std::sort of n <= ESYNC_MAX (key, index) pairs by the whole workgroup, level-synchronous (esort::block_esort_sync): keys are
// produced into registers by key_of(i); scratch = 16384 words of LDS (pool) laid out as pairs[2048] | left stops[2048] | right
// stops[2048] | counts[2048] | cuts[2048]; sorted keys -> K2, their indices -> V2 (LDS; both may lie in pool beyond its first 4096 words).
static constexpr uint32_t ESYNC_MAX = 2048;
// (`cap` >= n lays out the scratch: 8 * cap words of `pool`)
template <bool BIG = false, class KeyFn, class ValFn, class K2P, class V2P>
__device__ __forceinline__ void lds_esort_sync_kv(uint32_t n, KeyFn key_of, ValFn val_of, uint32_t *pool, uint32_t *stab, K2P K2, V2P V2,
                                                  uint32_t *n_fallback, unsigned long long *tstamp = nullptr, int32_t depth_budget = -1,
                                                  uint32_t cap = ESYNC_MAX) {
    uint2 *sKV = reinterpret_cast<uint2 *>(pool), *sLL = reinterpret_cast<uint2 *>(pool + 2 * cap), *sRR = reinterpret_cast<uint2 *>(pool + 4 * cap);
    uint32_t *sPS = pool + 6 * cap, *sCut = pool + 7 * cap;
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    if (n <= bs) {
        uint32_t k[1], v[1];
        k[0] = tid < n ? key_of(tid) : 0u;
        v[0] = tid < n ? val_of(tid) : 0u;
        esort::block_esort_sync<1>(k, v, n, sKV, sLL, sRR, sPS, sCut, stab, K2, V2, n_fallback, tstamp, depth_budget);
    } else {
        uint32_t k[2], v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t i = (uint32_t)e * bs + tid;
            k[e] = i < n ? key_of(i) : 0u;
            v[e] = i < n ? val_of(i) : 0u;
        }
        esort::block_esort_sync<2>(k, v, n, sKV, sLL, sRR, sPS, sCut, stab, K2, V2, n_fallback, tstamp, depth_budget);
    }
}
template <class KeyFn>
__device__ __forceinline__ void lds_esort_sync(uint32_t n, KeyFn key_of, uint32_t *pool, uint32_t *stab, uint32_t *K2, uint32_t *V2,
                                               uint32_t *n_fallback, unsigned long long *tstamp = nullptr) {
    lds_esort_sync_kv<false>(n, key_of, [](uint32_t i) { return i; }, pool, stab, K2, V2, n_fallback, tstamp);
}

static constexpr uint32_t ES_LMAX = 2048;   // segments up to this size are finished inside LDS
struct EsQueues {
    uint32_t cnt[3];      // three rotating level queues: level l reads [l%3], appends to [(l+1)%3], clears [(l+2)%3]
    uint32_t small_cnt;   // segments handed to the final kernel
};
// ---- "wide" partition: the top levels of a big sort, many workgroups per segment ----------------
// Per level three launches, no intra-kernel cross-workgroup dependency (coherence comes from kernel boundaries):
//   mark     : one workgroup per 2048-key tile: tile-local stop lists (in the tile's own index range) + counts
//   swap     : WPARTS workgroups per segment: tile prefix, m by a block-wide multiway search over the two-level
//              stop lists, cut, then the m swaps split between the workgroups
//   children : one workgroup: children -> next wide list (median move done here) / level queue / final queue
#ifndef ESORT_WIDE_MIN
#define ESORT_WIDE_MIN (ES_LMAX + 1)
#endif
static constexpr uint32_t WIDE_MIN = ESORT_WIDE_MIN;  // segments at least this long use the wide path (everything the LDS finisher cannot take)
static constexpr uint32_t WTILE = 2048;      // 256 threads x 8 keys
static constexpr uint32_t WSEG_MAX = 1024;   // wide segments per level (n / WIDE_MIN: ~16 M keys)
static constexpr uint32_t WTILES_MAX = 8192; // tiles per level (all wide segments together) -> n up to ~16 M keys
static constexpr uint32_t WPARTS = 8;        // workgroups sharing one segment's swaps
static constexpr int32_t MED_DONE = 0x40000000;  // flag in Seg::depth: median already moved to `first`

struct WideSeg {
    uint32_t first, last;
    int32_t depth;
    uint32_t tile0, ntiles, cut;
    // round 3: the median move (std::__move_median_to_first) of a wide segment is VIRTUAL while the level is marked -- k_esort_wide_mark
    // decides it from memory, patches the one key it moves into its reads and records it here; k_esort_wide_swap makes it physical
    // together with the partition's swaps.  (A kernel of its own used to do it between two levels: 9 launches per scan.)
    uint32_t pk, pv;    // the pivot: key and value that belong at `first`
    uint32_t kf, vf;    // the old head: key and value that belong at `mpos`
    uint32_t mpos;      // where the median was
    uint32_t mL, mR;    // index of mpos in its tile's left / right stop list (0xFFFFFFFF: not a stop of that kind)
    uint32_t pad0, pad1, pad2;
};
struct WideState {
    uint32_t nseg[2];
    uint32_t ntiles[2];
};

// (the queues of the exact scan sort: also opened by thread 0 of k_voxel_keys -- one launch less on the query chain)
struct EsInit {
    esort::Seg *q0, *smallq;
    EsQueues *qs;
    WideSeg *w0;
    WideState *ws;
};
__device__ __forceinline__ void esort_init(esort::Seg *q0, esort::Seg *smallq, EsQueues *qs, WideSeg *w0, WideState *ws, uint32_t n) {
    qs->cnt[0] = qs->cnt[1] = qs->cnt[2] = 0;
    qs->small_cnt = 0;
    ws->nseg[0] = ws->nseg[1] = 0;
    ws->ntiles[0] = ws->ntiles[1] = 0;
    if (n == 0) return;
    esort::Seg s;
    s.first = 0;
    s.last = n;
    s.depth = 2 * esort::lg2_floor(n);
    const uint32_t nt = (n - 1 + WTILE - 1) / WTILE;
    if (n >= WIDE_MIN && nt <= WTILES_MAX) {
        w0[0].first = 0;  // (its median move is the first level's business, like every wide segment's)
        w0[0].last = n;
        w0[0].depth = s.depth;
        w0[0].tile0 = 0;
        w0[0].ntiles = nt;
        ws->nseg[0] = 1;
        ws->ntiles[0] = nt;
    } else if (n > ES_LMAX) {
        q0[0] = s;
        qs->cnt[0] = 1;
    } else {
        smallq[0] = s;
        qs->small_cnt = 1;
    }
}
__global__ void k_esort_init(uint32_t *K, uint32_t *V, esort::Seg *q0, esort::Seg *smallq, EsQueues *qs, WideSeg *w0, WideState *ws,
                             uint32_t n) {
    esort_init(q0, smallq, qs, w0, ws, n);
}
// the scan's voxel keys AND the opening of the sort's queues (thread 0 of workgroup 0: the queues do not depend on the keys)
__device__ __forceinline__ void voxel_keys_es_body(const float4 *__restrict__ pts, uint32_t n, const uint32_t *__restrict__ bb,
                                                        float leaf, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                        VoxGrid *gout, Counters *ctr, uint32_t *__restrict__ hkey, uint32_t hsize, EsInit es) {
    if (blockIdx.x == 0 && threadIdx.x == 0) esort_init(es.q0, es.smallq, es.qs, es.w0, es.ws, n);
    voxel_keys_body(pts, n, bb, leaf, keys, vals, gout, ctr, hkey, hsize);
}
__global__ __launch_bounds__(256) void k_voxel_keys_es(const float4 *__restrict__ pts, uint32_t n, const uint32_t *__restrict__ bb,
                                                        float leaf, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                        VoxGrid *gout, Counters *ctr, uint32_t *__restrict__ hkey, uint32_t hsize, EsInit es) { voxel_keys_es_body(pts, n, bb, leaf, keys, vals, gout, ctr, hkey, hsize, es); }

__device__ __forceinline__ void esort_wide_mark_body(const uint32_t *__restrict__ K, const uint32_t *__restrict__ V, uint32_t *__restrict__ posL,
                                                          uint32_t *__restrict__ posR, WideSeg *__restrict__ wseg, WideState *ws, int cur,
                                                          uint32_t *__restrict__ tileL, uint32_t *__restrict__ tileR) {
    __shared__ uint32_t sm[40];
    __shared__ uint32_t s_si;
    const uint32_t nseg = min(ws->nseg[cur], WSEG_MAX), ntot = min(ws->ntiles[cur], WTILES_MAX);
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the next level's list is filled by this level's swap kernel (atomics): start it empty
        ws->nseg[cur ^ 1] = 0;
        ws->ntiles[cur ^ 1] = 0;
    }
    for (uint32_t gt = blockIdx.x; gt < ntot; gt += gridDim.x) {
        // which segment owns tile gt?  (slots and tile ranges are handed out by atomics: no ordering to rely on)
        if (threadIdx.x == 0) s_si = 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < nseg; c += blockDim.x) {
            const uint32_t t0 = wseg[c].tile0;
            if (gt >= t0 && gt < t0 + wseg[c].ntiles) s_si = c;
        }
        __syncthreads();
        const uint32_t si = s_si;
        if (si == 0xFFFFFFFFu) continue;  // tile ids of children that were routed to the level queue instead
        const uint32_t first = wseg[si].first, last = wseg[si].last, tile0 = wseg[si].tile0;
        const uint32_t t = gt - tile0;
        // std::__move_median_to_first(first, first + 1, mid, last - 1), decided but not performed
        const uint32_t ma = first + 1, mb = first + (last - first) / 2, mc = last - 1;
        const uint32_t kf = K[first], ka = K[ma], kb = K[mb], kc = K[mc];
        uint32_t mpos, p;
        if (ka < kb) {
            if (kb < kc) { mpos = mb; p = kb; }
            else if (ka < kc) { mpos = mc; p = kc; }
            else { mpos = ma; p = ka; }
        } else if (ka < kc) { mpos = ma; p = ka; }
        else if (kb < kc) { mpos = mc; p = kc; }
        else { mpos = mb; p = kb; }
        if (t == 0 && threadIdx.x == 0) {
            wseg[si].pk = p;
            wseg[si].kf = kf;
            wseg[si].vf = V[first];
            wseg[si].mpos = mpos;
        }
        const uint32_t lo = first + 1 + t * WTILE;
        const uint32_t hi = min(lo + WTILE, last);
        const uint32_t i0 = lo + threadIdx.x * 8;
        uint32_t k[8], fl = 0, fr = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = (i0 + j < hi) ? K[i0 + j] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j == mpos) k[j] = kf;  // (the median's place holds the old head)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool valid = i0 + j < hi;
            fl |= (valid && !(k[j] < p)) ? (1u << j) : 0u;
            fr |= (valid && !(p < k[j])) ? (1u << j) : 0u;
        }
        const uint32_t cl = __popc(fl), cr = __popc(fr);
        uint32_t tl, tr;
        uint32_t oL = lo + block_excl_scan(cl, sm, tl);
        uint32_t oR = lo + block_excl_scan(cr, sm, tr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (i0 + j == mpos && i0 + j < hi) {  // where the median was, as the swap kernel will look for it
                wseg[si].pv = V[mpos];
                wseg[si].mL = (fl & (1u << j)) ? oL - lo : 0xFFFFFFFFu;
                wseg[si].mR = (fr & (1u << j)) ? oR - lo : 0xFFFFFFFFu;
            }
            if (fl & (1u << j)) posL[oL++] = i0 + j;
            if (fr & (1u << j)) posR[oR++] = i0 + j;
        }
        if (threadIdx.x == 0) {
            tileL[gt] = tl;
            tileR[gt] = tr;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_esort_wide_mark(const uint32_t *__restrict__ K, const uint32_t *__restrict__ V, uint32_t *__restrict__ posL,
                                                          uint32_t *__restrict__ posR, WideSeg *__restrict__ wseg, WideState *ws, int cur,
                                                          uint32_t *__restrict__ tileL, uint32_t *__restrict__ tileR) { esort_wide_mark_body(K, V, posL, posR, wseg, ws, cur, tileL, tileR); }

// asc-th entry of a two-level stop list (tile prefix `pre` in LDS, tile-local lists in `list`)
__device__ __forceinline__ uint32_t wide_lookup(const uint32_t *pre, uint32_t ntiles, const uint32_t *list, uint32_t base, uint32_t asc) {
    uint32_t lo = 0, hi = ntiles;  // largest t with pre[t] <= asc
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pre[mid] <= asc) lo = mid; else hi = mid;
    }
    return list[base + lo * WTILE + (asc - pre[lo])];
}

// The level's swaps -- the median move made physical on the way -- and, by the workgroup that owns a segment's first share, the
// routing of its two children: next wide list (slots and tile ranges by atomics on the level state), level queue, finisher queue.
__device__ __forceinline__ void esort_wide_swap_body(uint32_t *K, uint32_t *V, const uint32_t *__restrict__ posL,
                                                          const uint32_t *__restrict__ posR, WideSeg *wseg, WideSeg *wnext, WideState *ws, int cur,
                                                          const uint32_t *__restrict__ tileL, const uint32_t *__restrict__ tileR, esort::Seg *q0,
                                                          esort::Seg *smallq, EsQueues *qs, uint32_t qcap, int last_level, Counters *ctr) {
    __shared__ uint32_t preL[WTILES_MAX + 1], preR[WTILES_MAX + 1];
    __shared__ uint32_t sm[40];
    __shared__ uint32_t s_lo, s_hi;
    const uint32_t nseg = min(ws->nseg[cur], WSEG_MAX);
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    if (nseg == 0) return;
    const uint32_t parts = max(1u, gridDim.x / nseg);  // workgroups sharing one segment's swaps (all of the grid at level 0)
    for (uint32_t w = blockIdx.x; w < nseg * parts; w += gridDim.x) {
        const uint32_t si = w / parts, part = w % parts;
        const WideSeg sg = wseg[si];
        const uint32_t nt = sg.ntiles, base = sg.first + 1;
        __syncthreads();  // (the previous segment's tables are still being read)
        if (nt == 0) continue;  // (a slot that could not get its tiles: see the routing below)
        // tile prefixes (exclusive), entry nt = total
        uint32_t carryL = 0, carryR = 0;
        for (uint32_t t0 = 0; t0 < nt; t0 += bs) {
            const uint32_t t = t0 + tid;
            const uint32_t a = t < nt ? tileL[sg.tile0 + t] : 0u, b = t < nt ? tileR[sg.tile0 + t] : 0u;
            uint32_t ta, tb;
            const uint32_t pa = block_excl_scan(a, sm, ta);
            const uint32_t pb = block_excl_scan(b, sm, tb);
            if (t < nt) {
                preL[t] = carryL + pa;
                preR[t] = carryR + pb;
            }
            carryL += ta;
            carryR += tb;
        }
        if (tid == 0) {
            preL[nt] = carryL;
            preR[nt] = carryR;
        }
        __syncthreads();
        const uint32_t nL = carryL, nR = carryR;
        const uint32_t lim = nL < nR ? nL : nR;
        // m = first k with !(L[k] < R[k]); monotone predicate -> block-wide multiway search
        if (tid == 0) {
            s_lo = 0;
            s_hi = lim;
        }
        __syncthreads();
        for (;;) {
            const uint32_t lo = s_lo, hi = s_hi;
            if (hi <= lo) break;
            const uint32_t step = (hi - lo + bs - 1) / bs;
            const uint32_t k = lo + tid * step;
            bool tr = false;
            if (k < hi) tr = wide_lookup(preL, nt, posL, base, k) < wide_lookup(preR, nt, posR, base, nR - 1 - k);
            uint32_t tot;
            block_excl_scan(tr ? 1u : 0u, sm, tot);  // monotone: the number of true answers is the index of the first false
            __syncthreads();
            if (tid == 0) {
                if (tot == 0) {
                    s_hi = lo;
                } else {
                    const uint32_t ktrue = lo + (tot - 1) * step;
                    s_lo = ktrue + 1;
                    s_hi = min(lo + tot * step, hi);
                }
            }
            __syncthreads();
        }
        const uint32_t m = s_lo;
        if (part == 0 && tid == 0) {
            uint32_t cut = 0xFFFFFFFFu;
            if (m < nL) cut = wide_lookup(preL, nt, posL, base, m);
            if (m > 0) {
                const uint32_t r = wide_lookup(preR, nt, posR, base, nR - m);
                if (r < cut) cut = r;
            }
            wseg[si].cut = cut;
            // ---- the median move, made physical: the pivot goes to `first` (nobody else touches it); the old head goes to the
            // median's place unless that place takes part in a swap -- the pair that holds it then moves the old head along ----
            K[sg.first] = sg.pk;
            V[sg.first] = sg.pv;
            const uint32_t tm = (sg.mpos - base) / WTILE;
            const bool swapped = (sg.mL != 0xFFFFFFFFu && preL[tm] + sg.mL < m) || (sg.mR != 0xFFFFFFFFu && nR - 1u - (preR[tm] + sg.mR) < m);
            if (!swapped) {
                K[sg.mpos] = sg.kf;
                V[sg.mpos] = sg.vf;
            }
            // ---- the two children (what k_esort_wide_children did): queue order is irrelevant, slots by atomics ----
            const esort::Seg ch[2] = {{sg.first, cut, sg.depth - 1}, {cut, sg.last, sg.depth - 1}};
            for (int t = 0; t < 2; ++t) {
                const uint32_t len = ch[t].last - ch[t].first;
                if (len == 0) continue;
                const uint32_t cnt_t = len > 1 ? (len - 1 + WTILE - 1) / WTILE : 0;
                bool wide = ch[t].depth > 0 && len >= WIDE_MIN && !last_level;
                if (wide) {
                    const uint32_t slot = atomicAdd(&ws->nseg[cur ^ 1], 1u);
                    if (slot >= WSEG_MAX) wide = false;  // over capacity: the level queue takes it (the counter only over-counts: clamped where read)
                    else {
                        const uint32_t t0 = atomicAdd(&ws->ntiles[cur ^ 1], cnt_t);
                        WideSeg nx;
                        nx.first = ch[t].first;
                        nx.last = ch[t].last;
                        nx.depth = ch[t].depth;
                        nx.tile0 = t0;
                        nx.ntiles = cnt_t;
                        nx.cut = 0;
                        nx.pk = nx.pv = nx.kf = nx.vf = nx.mpos = 0;
                        nx.mL = nx.mR = 0xFFFFFFFFu;
                        nx.pad0 = nx.pad1 = nx.pad2 = 0;
                        if (t0 + cnt_t > WTILES_MAX) {  // no tiles left: the slot stays, empty, and the level queue takes the segment
                            nx.first = nx.last = 0;
                            nx.ntiles = 0;
                            wide = false;
                        }
                        wnext[slot] = nx;
                    }
                }
                if (!wide) {
                    if (len > ES_LMAX && ch[t].depth > 0) {
                        const uint32_t at = atomicAdd(&qs->cnt[0], 1u);
                        if (at < qcap) q0[at] = ch[t]; else ctr->sort_qoverflow = 1;
                    } else {
                        const uint32_t at = atomicAdd(&qs->small_cnt, 1u);
                        if (at < qcap) smallq[at] = ch[t]; else ctr->sort_qoverflow = 1;
                    }
                }
            }
        }
        const uint32_t k_begin = (uint32_t)(((uint64_t)m * part) / parts), k_end = (uint32_t)(((uint64_t)m * (part + 1)) / parts);
        for (uint32_t k = k_begin + tid; k < k_end; k += bs) {
            const uint32_t a = wide_lookup(preL, nt, posL, base, k), b = wide_lookup(preR, nt, posR, base, nR - 1 - k);
            uint32_t ka = K[a], va = V[a], kb = K[b], vb = V[b];
            if (a == sg.mpos) { ka = sg.kf; va = sg.vf; }  // (the median's place holds the old head)
            if (b == sg.mpos) { kb = sg.kf; vb = sg.vf; }
            K[a] = kb;
            V[a] = vb;
            K[b] = ka;
            V[b] = va;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_esort_wide_swap(uint32_t *K, uint32_t *V, const uint32_t *__restrict__ posL,
                                                          const uint32_t *__restrict__ posR, WideSeg *wseg, WideSeg *wnext, WideState *ws, int cur,
                                                          const uint32_t *__restrict__ tileL, const uint32_t *__restrict__ tileR, esort::Seg *q0,
                                                          esort::Seg *smallq, EsQueues *qs, uint32_t qcap, int last_level, Counters *ctr) { esort_wide_swap_body(K, V, posL, posR, wseg, wnext, ws, cur, tileL, tileR, q0, smallq, qs, qcap, last_level, ctr); }

// one level: each workgroup takes big segments of queue[cur] and performs one partition
__global__ __launch_bounds__(1024) void k_esort_level(uint32_t *K, uint32_t *V, uint32_t *posL, uint32_t *posR, esort::Seg *q0,
                                                       esort::Seg *q1, esort::Seg *q2, esort::Seg *smallq, EsQueues *qs, int level,
                                                       uint32_t qcap, Counters *ctr) {
    __shared__ uint32_t sm[40];
    const int cur = level % 3, nxt = (level + 1) % 3, clr = (level + 2) % 3;
    esort::Seg *qcur = cur == 0 ? q0 : (cur == 1 ? q1 : q2);
    esort::Seg *qnext = nxt == 0 ? q0 : (nxt == 1 ? q1 : q2);
    if (blockIdx.x == 0 && threadIdx.x == 0) qs->cnt[clr] = 0;  // not read or appended to by this launch
    const uint32_t nseg = qs->cnt[cur];
    for (uint32_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const esort::Seg sg = qcur[s];
        if (sg.depth == 0) {  // hand over: the final kernel runs the exact heapsort
            if (threadIdx.x == 0) {
                const uint32_t at = atomicAdd(&qs->small_cnt, 1u);
                if (at < qcap) smallq[at] = sg; else ctr->sort_qoverflow = 2;
            }
            continue;
        }
        const uint32_t cut = esort::block_partition<8>(K, V, posL, posR, sg.first, sg.last, sm);
        if (threadIdx.x == 0) {
            esort::Seg a, b;
            a.first = sg.first; a.last = cut; a.depth = sg.depth - 1;
            b.first = cut; b.last = sg.last; b.depth = sg.depth - 1;
            const esort::Seg ch[2] = {a, b};
            for (int t = 0; t < 2; ++t) {
                const uint32_t len = ch[t].last - ch[t].first;
                if (len == 0) continue;
                if (len > ES_LMAX) {
                    const uint32_t at = atomicAdd(&qs->cnt[nxt], 1u);
                    if (at < qcap) qnext[at] = ch[t]; else ctr->sort_qoverflow = 2;
                } else {
                    const uint32_t at = atomicAdd(&qs->small_cnt, 1u);
                    if (at < qcap) smallq[at] = ch[t]; else ctr->sort_qoverflow = 2;
                }
            }
        }
        __syncthreads();
    }
}

// "mid" kernel: every segment still longer than ES_LMAX after the wide levels is cut down to LDS-finisher size by ONE
// workgroup, depth-first, all of its block-wide partitions in one launch (in LDS when the segment fits: a partition costs
// a few microseconds there, against ~10 us per partition-and-launch through k_esort_level).  Pieces go to the small queue.
static constexpr uint32_t ES_MID_LMAX = 8192;
__device__ __forceinline__ void esort_mid_body(uint32_t *K, uint32_t *V, uint32_t *posL, uint32_t *posR, const esort::Seg *bigq,
                                                     esort::Seg *smallq, EsQueues *qs, int bigcur, uint32_t qcap, Counters *ctr) {
    __shared__ uint32_t sK[ES_MID_LMAX], sV[ES_MID_LMAX], sL[ES_MID_LMAX], sR[ES_MID_LMAX];
    __shared__ uint32_t sm[40];
    __shared__ esort::Seg stk[72];
    __shared__ int sp;
    __shared__ esort::Seg cur;
    const uint32_t nbig = qs->cnt[bigcur];
    for (uint32_t s = blockIdx.x; s < nbig; s += gridDim.x) {
        const esort::Seg sg = bigq[s];
        const uint32_t len = sg.last - sg.first;
        if (sg.depth <= 0 || len <= ES_LMAX) {  // nothing to partition here: straight to the finisher
            if (threadIdx.x == 0) {
                const uint32_t at = atomicAdd(&qs->small_cnt, 1u);
                if (at < qcap) smallq[at] = sg; else ctr->sort_qoverflow = 2;
            }
            continue;
        }
        const bool in_lds = len <= ES_MID_LMAX;
        __syncthreads();
        if (in_lds)
            for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
                sK[i] = K[sg.first + i];
                sV[i] = V[sg.first + i];
            }
        if (threadIdx.x == 0) {
            stk[0] = sg;
            sp = 1;
        }
        for (;;) {
            __syncthreads();
            if (sp == 0) break;  // block-uniform
            __syncthreads();
            if (threadIdx.x == 0) cur = stk[--sp];
            __syncthreads();
            const esort::Seg c = cur;
            uint32_t cut;
            // the LDS copy is indexed relative to the segment's first key
            if (in_lds) cut = esort::block_partition<4>(sK, sV, sL, sR, c.first - sg.first, c.last - sg.first, sm) + sg.first;
            else cut = esort::block_partition<8>(K, V, posL, posR, c.first, c.last, sm);
            if (threadIdx.x == 0) {
                const esort::Seg ch[2] = {{c.first, cut, c.depth - 1}, {cut, c.last, c.depth - 1}};
                for (int t = 0; t < 2; ++t) {
                    const uint32_t l = ch[t].last - ch[t].first;
                    if (l == 0) continue;
                    if (l > ES_LMAX && ch[t].depth > 0 && sp < 72) {
                        stk[sp++] = ch[t];
                    } else {  // finisher size (or depth budget exhausted: the finisher runs the exact heapsort)
                        const uint32_t at = atomicAdd(&qs->small_cnt, 1u);
                        if (at < qcap) smallq[at] = ch[t]; else ctr->sort_qoverflow = 2;
                    }
                }
            }
        }
        __syncthreads();
        if (in_lds)
            for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
                K[sg.first + i] = sK[i];
                V[sg.first + i] = sV[i];
            }
    }
}
__global__ __launch_bounds__(1024) void k_esort_mid(uint32_t *K, uint32_t *V, uint32_t *posL, uint32_t *posR, const esort::Seg *bigq,
                                                     esort::Seg *smallq, EsQueues *qs, int bigcur, uint32_t qcap, Counters *ctr) { esort_mid_body(K, V, posL, posR, bigq, smallq, qs, bigcur, qcap, ctr); }

// final: every remaining segment (any size) is sorted to completion by one workgroup.
// Segments <= ES_LMAX run in LDS; larger ones (only if the level budget ran out) run in place in global memory.
__device__ __forceinline__ void esort_final_body(uint32_t *K, uint32_t *V, uint32_t *posL, uint32_t *posR, uint32_t *head,
                                                      uint32_t *K2, uint32_t *V2, const esort::Seg *smallq, const esort::Seg *bigq,
                                                      EsQueues *qs, int bigcur, Counters *ctr, unsigned long long *dbg) {  // bigcur: queue index (0..2) still holding big segments
    __shared__ uint32_t pool[8 * ESYNC_MAX];  // lds_esort_sync_kv's scratch (pairs | left stops | right stops | counts | cuts)
    __shared__ uint32_t s_stab[68];
    __shared__ esort::Seg qa[ES_LMAX / 16 + 2], qb[ES_LMAX / 16 + 2];  // (the global-memory path's queues)
    __shared__ uint32_t qcnt[2];
    static_assert(ES_LMAX <= ESYNC_MAX, "the finisher's LDS-resident segments go through the level-synchronous sort");
    const uint32_t nsmall = qs->small_cnt, nbig = qs->cnt[bigcur];
    if (dbg && threadIdx.x == 0) atomicMin(&dbg[28], wall_clock64());  // diagnostics: first workgroup start (100 MHz ticks)
    if (dbg && threadIdx.x == 0 && blockIdx.x == 0) {
        dbg[25] = nbig;
        unsigned long long mx = 0;
        for (uint32_t i = 0; i < nbig; ++i) mx = max(mx, (unsigned long long)(bigq[i].last - bigq[i].first));
        dbg[24] = mx;
    }
    for (uint32_t s = blockIdx.x; s < nsmall + nbig; s += gridDim.x) {
        const esort::Seg sg = s < nsmall ? smallq[s] : bigq[s - nsmall];
        const uint32_t len = sg.last - sg.first;
        if (len <= ES_LMAX) {
            // round 3: level-synchronous over the whole workgroup (esort::block_esort_sync), keys straight from global memory into
            // registers, results straight back; the segment brings what is left of its introsort depth budget
            __shared__ unsigned long long s_stamps[24];
            const unsigned long long t_a = clock64();
            const uint32_t *Kg = K + sg.first, *Vg = V + sg.first;
            __syncthreads();  // (the previous segment's ranking may still be reading the pool)
            lds_esort_sync_kv<false>(len, [&](uint32_t i) { return Kg[i]; }, [&](uint32_t i) { return Vg[i]; }, pool, s_stab, K2 + sg.first, V2 + sg.first,
                              &ctr->n_sort_fallback, dbg ? s_stamps : nullptr, sg.depth);
            if (dbg && threadIdx.x == 0) {  // diagnostics: keep the stamps of the slowest segment
                const unsigned long long dur = clock64() - t_a;
                if (atomicMax(&dbg[31], dur) < dur) {
                    for (int i = 0; i < 16; ++i) dbg[i] = s_stamps[i];
                    dbg[30] = ((unsigned long long)len << 32) | (unsigned)sg.depth;
                    dbg[29] = (unsigned long long)(nsmall + nbig);
                }
            }
            __syncthreads();
        } else {
            // in-place global path.  Queue capacity is the LDS queue; a segment this large after the level
            // budget means pathological input; correctness is kept as long as the queue does not overflow.
            esort::block_esort(K, V, posL, posR, head, K2, V2, sg.first, sg.last, sg.depth, qa, qb, qcnt,
                               (uint32_t)(ES_LMAX / 16 + 2), &ctr->n_sort_fallback, &ctr->sort_qoverflow);
        }
        if (dbg && threadIdx.x == 0) atomicMax(&dbg[26], wall_clock64());  // last working workgroup end
    }
    if (dbg && threadIdx.x == 0) atomicMax(&dbg[27], wall_clock64());  // last workgroup (working or not) end
}
__global__ __launch_bounds__(1024) void k_esort_final(uint32_t *K, uint32_t *V, uint32_t *posL, uint32_t *posR, uint32_t *head,
                                                      uint32_t *K2, uint32_t *V2, const esort::Seg *smallq, const esort::Seg *bigq,
                                                      EsQueues *qs, int bigcur, Counters *ctr, unsigned long long *dbg) { esort_final_body(K, V, posL, posR, head, K2, V2, smallq, bigq, qs, bigcur, ctr, dbg); }

// run heads of the sorted voxel keys
__global__ __launch_bounds__(256) void k_run_heads(const uint32_t *__restrict__ skeys, uint32_t n, uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1u : 0u;
}
// run_begin[vid] = i for heads; run_begin[nv] = n
__global__ __launch_bounds__(256) void k_run_begin(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pl,
                                                    const uint32_t *__restrict__ tops, uint32_t n, uint32_t *__restrict__ run_begin,
                                                    uint32_t *nv_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        const uint32_t nv = n ? tops[(n + 1023) / 1024] : 0u;
        run_begin[nv] = n;
        *nv_out = nv;
        return;
    }
    if (flag[i]) run_begin[pl[i] + tops[i >> 10]] = i;
}

// The same in TWO launches instead of four, for up to 1024 tiles of 1024 keys (a scan is ~120 tiles): tile totals, then every
// tile adds up the totals before it (<= 4 loads per thread), recomputes its heads and writes run_begin directly.
__device__ __forceinline__ uint32_t run_heads4(const uint32_t *__restrict__ skeys, uint32_t n, uint32_t base, bool hd[4]) {
    uint32_t prev = (base > 0 && base - 1 < n) ? skeys[base - 1] : 0u, c = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        hd[j] = false;
        if (base + j < n) {
            const uint32_t k = skeys[base + j];
            hd[j] = (base + j == 0) || k != prev;
            prev = k;
            c += hd[j] ? 1u : 0u;
        }
    }
    return c;
}
__device__ __forceinline__ void run_count_body(const uint32_t *__restrict__ skeys, uint32_t n, uint32_t *__restrict__ tops) {
    __shared__ uint32_t sm[40];
    bool hd[4];
    const uint32_t c = run_heads4(skeys, n, blockIdx.x * 1024u + threadIdx.x * 4u, hd);
    uint32_t tot;
    block_excl_scan(c, sm, tot);
    if (threadIdx.x == 0) tops[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_run_count(const uint32_t *__restrict__ skeys, uint32_t n, uint32_t *__restrict__ tops) { run_count_body(skeys, n, tops); }
__device__ __forceinline__ void run_emit_body(const uint32_t *__restrict__ skeys, uint32_t n, const uint32_t *__restrict__ tops,
                                                   uint32_t ntile, uint32_t *__restrict__ run_begin, uint32_t *nv_out) {
    __shared__ uint32_t sm[40];
    uint32_t before = 0, all = 0;
    for (uint32_t i = threadIdx.x; i < ntile; i += blockDim.x) {
        const uint32_t v = tops[i];
        all += v;
        if (i < blockIdx.x) before += v;
    }
    uint32_t off, nv;
    block_excl_scan(before, sm, off);
    block_excl_scan(all, sm, nv);
    bool hd[4];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    const uint32_t c = run_heads4(skeys, n, base, hd);
    uint32_t tot;
    uint32_t p = off + block_excl_scan(c, sm, tot);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (hd[j]) run_begin[p++] = base + j;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        run_begin[nv] = n;
        *nv_out = nv;
    }
}
__global__ __launch_bounds__(256) void k_run_emit(const uint32_t *__restrict__ skeys, uint32_t n, const uint32_t *__restrict__ tops,
                                                   uint32_t ntile, uint32_t *__restrict__ run_begin, uint32_t *nv_out) { run_emit_body(skeys, n, tops, ntile, run_begin, nv_out); }

__device__ __forceinline__ uint32_t vox_hash(uint32_t key, int hbits) { return (key * 2654435761u) >> (32 - hbits); }

// CentroidPoint<PointXYZI>: float32 running sums in sorted order, each / (float)count
__device__ __forceinline__ void centroids_body(const float4 *__restrict__ pts, const uint32_t *__restrict__ skeys,
                                                    const uint32_t *__restrict__ sperm, const uint32_t *__restrict__ run_begin,
                                                    const uint32_t *nv_dev, float4 *__restrict__ cent, uint32_t *__restrict__ ukeys,
                                                    uint32_t *__restrict__ hkey, uint32_t *__restrict__ hval, int hbits) {
    // Eight lanes per voxel: they FETCH eight points of the run at once (index, then point: two dependent loads per
    // batch instead of two per point); the float32 sums stay a strictly sequential chain in run order -- every lane
    // replays it from the batch's registers.
    constexpr uint32_t SUB = 8;
    const uint32_t v = (blockIdx.x * blockDim.x + threadIdx.x) / SUB, sub = threadIdx.x & (SUB - 1);
    if (v >= *nv_dev) return;  // the eight lanes of a voxel leave together
    const int lane0 = (int)((threadIdx.x & 63u) & ~(SUB - 1));
    const uint32_t s = run_begin[v], e = run_begin[v + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (uint32_t base = s; base < e; base += SUB) {
        const uint32_t li = base + sub;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (li < e) p = pts[sperm[li]];
        const uint32_t cnt = min(SUB, e - base);
#pragma unroll
        for (uint32_t j = 0; j < SUB; ++j) {
            const float px = __shfl(p.x, lane0 + (int)j, 64), py = __shfl(p.y, lane0 + (int)j, 64);
            const float pz = __shfl(p.z, lane0 + (int)j, 64), pw = __shfl(p.w, lane0 + (int)j, 64);
            if (j < cnt) {
                sx += px;
                sy += py;
                sz += pz;
                si += pw;
            }
        }
    }
    if (sub != 0) return;
    const float c = (float)(e - s);
    cent[v] = make_float4(sx / c, sy / c, sz / c, si / c);
    const uint32_t key = skeys[s];
    ukeys[v] = key;
    // voxel key -> voxel id, open addressing (load factor <= 1/2; keys are unique, so the slot is ours once claimed)
    uint32_t sl = vox_hash(key, hbits);
    for (;;) {
        const uint32_t old = atomicCAS(&hkey[sl], 0xFFFFFFFFu, key);
        if (old == 0xFFFFFFFFu || old == key) break;
        sl = (sl + 1) & ((1u << hbits) - 1u);
    }
    hval[sl] = v;
}
__global__ __launch_bounds__(256) void k_centroids(const float4 *__restrict__ pts, const uint32_t *__restrict__ skeys,
                                                    const uint32_t *__restrict__ sperm, const uint32_t *__restrict__ run_begin,
                                                    const uint32_t *nv_dev, float4 *__restrict__ cent, uint32_t *__restrict__ ukeys,
                                                    uint32_t *__restrict__ hkey, uint32_t *__restrict__ hval, int hbits) { centroids_body(pts, skeys, sperm, run_begin, nv_dev, cent, ukeys, hkey, hval, hbits); }

// FLANN L2_Simple: ((0 + dx*dx) + dy*dy) + dz*dz in float32
__device__ __forceinline__ float l2_simple(float ax, float ay, float az, float bx, float by, float bz) {
    float r = 0.f;
    const float d0 = ax - bx;
    r += d0 * d0;
    const float d1 = ay - by;
    r += d1 * d1;
    const float d2 = az - bz;
    r += d2 * d2;
    return r;
}

// exact 1-NN of every centroid among the input points (lowest index on float ties), searched through
// the sorted voxel runs; then tf_lidar2body and the query's R-POD key (OMU.cpp:240; erasor.cpp:100-115).
// EIGHT LANES PER VOXEL: the points of a cell and the cells of a shell are dealt out to the lanes, neighbour cells are
// found through the voxel hash (one or two probes instead of a 15-step binary search), and the lanes' candidates are
// merged by (distance, index) -- the order in which candidates are looked at cannot change that minimum.
static constexpr uint32_t NN_SUB = 8;
__device__ __forceinline__ void nn_take(float dd, uint32_t pi, float &best, uint32_t &best_i) {
    if (dd < best || (dd == best && pi < best_i)) {
        best = dd;
        best_i = pi;
    }
}
__device__ __forceinline__ void nn_merge(float &best, uint32_t &best_i) {
#pragma unroll
    for (int o = NN_SUB / 2; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const uint32_t oi = __shfl_xor(best_i, o, 64);
        nn_take(ob, oi, best, best_i);
    }
}
__device__ __forceinline__ void query_nn_body(const float4 *__restrict__ pts, const uint32_t *__restrict__ sperm,
                                                   const uint32_t *__restrict__ run_begin, const uint32_t *__restrict__ ukeys,
                                                   const float4 *__restrict__ cent, const uint32_t *nv_dev, const VoxGrid *gp, Xf Tl2b,
                                                   DP P, Counters *ctr, float4 *__restrict__ query, uint32_t *__restrict__ qkey,
                                                   const uint32_t *__restrict__ hkey, const uint32_t *__restrict__ hval, int hbits) {
    const uint32_t v = (blockIdx.x * blockDim.x + threadIdx.x) / NN_SUB, sub = threadIdx.x & (NN_SUB - 1);
    const uint32_t nv = *nv_dev;
    if (v >= nv) return;  // the eight lanes of a voxel leave together
    const VoxGrid g = *gp;
    if (g.overflow) return;  // the cloud was refused (k_voxel_keys): the step fails, nothing downstream is used
    const float4 c = cent[v];
    const uint32_t key = ukeys[v];
    const int dx = g.div_b[0], dy = g.div_b[1], dz = g.div_b[2];
    const int ci = (int)(key % (uint32_t)dx), cj = (int)((key / (uint32_t)dx) % (uint32_t)dy), ck = (int)(key / ((uint32_t)dx * (uint32_t)dy));
    const double L = 1.0 / (double)g.inv_leaf;
    const double cc[3] = {(double)c.x, (double)c.y, (double)c.z};
    const int cidx[3] = {ci, cj, ck};
    float best = __int_as_float(0x7F800000);
    uint32_t best_i = 0xFFFFFFFFu;
    const int maxrho = max(dx, max(dy, dz));
    // stage 0: the voxel's own points.  If the best of them is closer than the centroid's distance to the walls of its
    // own cell, no point of any other cell can beat or tie it (single-point voxels end here with distance 0).
    for (uint32_t li = run_begin[v] + sub; li < run_begin[v + 1]; li += NN_SUB) {
        const uint32_t pi = sperm[li];
        const float4 p = pts[pi];
        nn_take(l2_simple(c.x, c.y, c.z, p.x, p.y, p.z), pi, best, best_i);
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
    const uint32_t hmask = (1u << hbits) - 1u;
    // one cell of a shell: pruned by its squared distance d2c from the centroid (a cell whose nearest corner / face is farther than
    // the current best cannot hold a closer or tied point), found through the voxel hash, its points taken
    auto visit = [&](int ii, int jj, int kk, double d2c, float shell_best) {
        if (d2c > (double)shell_best) return;
        const uint32_t q = (uint32_t)ii + (uint32_t)jj * (uint32_t)dx + (uint32_t)kk * (uint32_t)dx * (uint32_t)dy;
        uint32_t sl = vox_hash(q, hbits), hk;
        while ((hk = hkey[sl]) != q && hk != 0xFFFFFFFFu) sl = (sl + 1) & hmask;
        if (hk != q) return;  // empty cell
        const uint32_t w = hval[sl];
        for (uint32_t li = run_begin[w]; li < run_begin[w + 1]; ++li) {
            const uint32_t pi = sperm[li];
            const float4 p = pts[pi];
            nn_take(l2_simple(c.x, c.y, c.z, p.x, p.y, p.z), pi, best, best_i);
        }
    };
    // squared distance from the centroid to the slab of cells `cell_a` along axis a (the summand of d2c)
    auto slab_d2 = [&](int a, int cell_a) -> double {
        const double lo_a = (double)(g.min_b[a] + cell_a) * L, hi_a = (double)(g.min_b[a] + cell_a + 1) * L;
        const double margin = 1e-3 * L + 1e-6 * fabs(cc[a]);
        const double da = fmax(0.0, fmax(lo_a - cc[a], cc[a] - hi_a) - margin);
        return da * da;
    };
    for (int rho = 1; !done; ++rho) {
        const float shell_best = best;  // pruning bound for this shell (the same in all eight lanes; a looser bound only prunes less)
        if (rho == 1) {
            // round 3: the first shell (where nearly every search ends) without the triple loop: the 26 neighbours are dealt to the
            // eight lanes by their number (which lane takes which cell does not matter: candidates merge by (distance, index)), and
            // the three summands of d2c come from a 3 x 3 table of slab distances instead of 24 float64 operations per cell
            double tab[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int o = 0; o < 3; ++o) tab[a][o] = slab_d2(a, cidx[a] + o - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = (int)sub + 8 * r;  // cell number: (kk - ck + 1) * 9 + (jj - cj + 1) * 3 + (ii - ci + 1)
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
                        if (!shell_jk && abs(ii - ci) < rho) continue;  // interior cells were visited by earlier stages (rho-1, ..., 0)
                        if ((turn++ & (NN_SUB - 1)) != sub) continue;   // this cell belongs to another lane
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
            const double t = fmin(cc[a] - lo, hi - cc[a]) - margin;
            gmin = fmin(gmin, t);
        }
        if (best_i != 0xFFFFFFFFu && gmin > 0.0 && (double)best <= gmin * gmin) break;
    }
    if (sub != 0) return;
    float4 o = c;
    o.w = pts[best_i].w;  // utils.cpp:109
    const float4 b = xform(Tl2b, o);
    query[v] = b;
    qkey[v] = bin_key(P, b.x, b.y, b.z, ctr);
}
__global__ __launch_bounds__(256) void k_query_nn(const float4 *__restrict__ pts, const uint32_t *__restrict__ sperm,
                                                   const uint32_t *__restrict__ run_begin, const uint32_t *__restrict__ ukeys,
                                                   const float4 *__restrict__ cent, const uint32_t *nv_dev, const VoxGrid *gp, Xf Tl2b,
                                                   DP P, Counters *ctr, float4 *__restrict__ query, uint32_t *__restrict__ qkey,
                                                   const uint32_t *__restrict__ hkey, const uint32_t *__restrict__ hval, int hbits) { query_nn_body(pts, sperm, run_begin, ukeys, cent, nv_dev, gp, Tl2b, P, ctr, query, qkey, hkey, hval, hbits); }

// query handed over already voxelised and in the body frame (ERASOR::set_inputs used directly, erasor.cpp:57-73)
__global__ __launch_bounds__(256) void k_query_direct(const float4 *__restrict__ src, uint32_t n, DP P, Counters *ctr,
                                                       float4 *__restrict__ query, uint32_t *__restrict__ qkey) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i];
    query[i] = p;
    qkey[i] = bin_key(P, p.x, p.y, p.z, ctr);
}

// ================================================================================================
// Scan Ratio Test + bin selection (v3: erasor.cpp:438-563; v2: erasor.cpp:332-427).  Single block.
// Keys are theta-major (key = sector*R + ring), so key order == the reference's loop order.
// ================================================================================================
__device__ __forceinline__ double std_min_d(double a, double b) { return (b < a) ? b : a; }

// per-bin pieces of the Scan Ratio Test, shared by the two kernel variants below
struct BinStat {
    uint32_t mc, cc;
    double mmaxh, mminh, cmaxh, cminh;
};
__device__ __forceinline__ BinStat srt_load(int key, const uint32_t *mcnt, const float *mmin, const float *mmax, const uint32_t *ccnt,
                                            const float *cmin, const float *cmax) {
    BinStat b;
    b.mc = mcnt[key];
    b.cc = ccnt[key];
    b.mmaxh = b.mc ? (double)mmax[key] : -INF_H;
    b.mminh = b.mc ? (double)mmin[key] : INF_H;
    b.cmaxh = b.cc ? (double)cmax[key] : -INF_H;
    b.cminh = b.cc ? (double)cmin[key] : INF_H;
    return b;
}
// first pass of v3 (erasor.cpp:467-501): status from the pseudo-occupancy ratio
__device__ __forceinline__ uint8_t srt_first(const DP &P, const BinStat &b) {
    uint8_t s = ST_LITTLE;
    if (P.version == 3) {
        if (b.mc == 0) {
            s = ST_LITTLE;
        } else if ((long long)b.cc < (long long)P.min_pts) {
            s = ST_LITTLE;
        } else {
            const double md = b.mmaxh - b.mminh, cd = b.cmaxh - b.cminh;
            const double ratio = std_min_d(md / cd, cd / md);
            if (b.cc > 0 && b.mc > 0) {
                if (ratio < P.srt_thr) {
                    if (md >= cd) s = ST_MAP;
                    else if (md <= cd) s = ST_CURR;
                } else {
                    s = ST_MERGE;
                }
            }
        }
    }
    return s;
}
// k_bin_stats of the MAP side with the Scan Ratio Test's first pass folded in (round 3): the wavefront that has a bin's map
// statistics also reads the query's (its chain is done: the main stream has waited for it) and leaves the bin's first-pass status in
// st1 -- bit 7: the map bin is taller than 0.5 m (the v3 revert gate, erasor.cpp:511).  540 workgroups share the float64
// divisions that k_srt4's ONE workgroup used to run for all 2160 bins (2700 instructions on one compute unit, 18 us).
__global__ __launch_bounds__(256) void k_bin_stats_srt(DP P, const float4 *__restrict__ spts, const uint32_t *__restrict__ off, uint32_t B,
                                                        uint32_t *__restrict__ cnt, float *__restrict__ minz, float *__restrict__ maxz,
                                                        const uint32_t *__restrict__ ccnt, const float *__restrict__ cmin,
                                                        const float *__restrict__ cmax, uint8_t *__restrict__ st1) {
    const uint32_t lane = threadIdx.x & 63u;
    CHAIN_STAMP(1);
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= B) return;
    const uint32_t s = off[b], e = off[b + 1];
    float mn = __int_as_float(0x7F800000), mx = __int_as_float(0xFF800000);
    uint32_t i = s + lane;
    for (; i + 192 < e; i += 256) {
        const float z0 = spts[i].z, z1 = spts[i + 64].z, z2 = spts[i + 128].z, z3 = spts[i + 192].z;
        mn = z0 < mn ? z0 : mn;
        mx = z0 > mx ? z0 : mx;
        mn = z1 < mn ? z1 : mn;
        mx = z1 > mx ? z1 : mx;
        mn = z2 < mn ? z2 : mn;
        mx = z2 > mx ? z2 : mx;
        mn = z3 < mn ? z3 : mn;
        mx = z3 > mx ? z3 : mx;
    }
    for (; i < e; i += 64) {
        const float z = spts[i].z;
        mn = z < mn ? z : mn;
        mx = z > mx ? z : mx;
    }
    mn = wave_min_f(mn);  // (z values are finite: the scan was checked for NaN / Inf, so fminf == the compare chain)
    mx = wave_max_f(mx);
    if (lane == 0) {
        cnt[b] = e - s;
        minz[b] = mn;
        maxz[b] = mx;
        BinStat bsx;  // exactly what srt_load would read back
        bsx.mc = e - s;
        bsx.cc = ccnt[b];
        bsx.mmaxh = bsx.mc ? (double)mx : -INF_H;
        bsx.mminh = bsx.mc ? (double)mn : INF_H;
        bsx.cmaxh = bsx.cc ? (double)cmax[b] : -INF_H;
        bsx.cminh = bsx.cc ? (double)cmin[b] : INF_H;
        const uint8_t s1 = srt_first(P, bsx);
        st1[b] = (uint8_t)(s1 | (((bsx.mmaxh - bsx.mminh) > 0.5) ? 0x80u : 0u));
    }
}

// second pass (erasor.cpp:503-560 for v3, :332-434 for v2): final status and the action of the bin.  ST1: st1[] accessor
template <class ST1>
__device__ __forceinline__ void srt_second(const DP &P, int key, const BinStat &b, uint8_t s, ST1 st1, uint8_t &fs, uint8_t &act) {
    fs = ST_LITTLE;
    act = 0;
    if (P.version == 3) {
        if (s == ST_MAP) {
            if ((b.mmaxh - b.mminh) > 0.5) {  // erasor.cpp:511
                fs = ST_MAP;
                act = 1;
            } else {
                fs = ST_NOT_ASSIGNED;
            }
        } else if (s == ST_CURR) {
            fs = ST_CURR;
        } else if (s == ST_MERGE) {
            // is_dynamic_obj_close(r, theta, 1, 1): erasor.cpp:573-595 (theta wrap uses num_rings)
            const int r_t = key % P.R, th_t = key / P.R;
            bool close = false;
            for (int j = th_t - 1; j <= th_t + 1; ++j) {
                int th = j;
                if (j < 0) th = j + P.R;
                else if (j >= P.S) th = j - P.R;
                if (th < 0 || th >= P.S) continue;  // reference: out-of-bounds read when num_rings > num_sectors
                for (int r = max(0, r_t - 1); r <= min(r_t + 1, P.R - 1); ++r) {
                    if (r == r_t && th == th_t) continue;
                    if (st1[th * P.R + r] == ST_CURR) close = true;
                }
            }
            fs = close ? ST_BLOCKED : ST_MERGE;
        }
    } else {  // version 2
        if ((long long)b.cc < (long long)P.min_pts) {
            act = 0;
        } else if (b.cc > 0 && b.mc > 0) {
            const double md = b.mmaxh - b.mminh, cd = b.cmaxh - b.cminh;
            const double ratio = std_min_d(md / cd, cd / md);
            if (ratio < P.srt_thr) {
                if (md >= cd) {
                    fs = ST_MAP;
                    act = (b.mmaxh > P.th_bin_max_h) ? 1 : 0;
                } else if (md <= cd) {
                    fs = ST_CURR;
                    act = (b.cmaxh > P.th_bin_max_h) ? 4 : 0;  // 4: keep map bin, curr points -> curr_rejected
                }
            } else {
                fs = ST_MERGE;
                act = 2;
            }
        } else if (b.cc > 0) {
            act = 3;
        }
    }
}

// General variant: any number of bins, one key per thread and round.
__global__ __launch_bounds__(1024) void k_srt(DP P, const uint32_t *__restrict__ mcnt, const float *__restrict__ mmin,
                                               const float *__restrict__ mmax, const uint32_t *__restrict__ ccnt,
                                               const float *__restrict__ cmin, const float *__restrict__ cmax, uint8_t *__restrict__ st1,
                                               uint8_t *__restrict__ status, uint8_t *__restrict__ action, uint32_t *__restrict__ rev_idx,
                                               uint32_t *__restrict__ rev_list, uint32_t *__restrict__ vox_off, DevState *st) {
    // action: 0 keep map bin, 1 revert (curr + ground(map)), 2 v2 merge (curr then map), 3 v2 curr only
    __shared__ uint32_t sm[40];
    __shared__ uint32_t carry[2];
    const int B = P.B;
    for (int key = threadIdx.x; key < B; key += blockDim.x) st1[key] = srt_first(P, srt_load(key, mcnt, mmin, mmax, ccnt, cmin, cmax));
    __syncthreads();
    for (int key = threadIdx.x; key < B; key += blockDim.x) {
        uint8_t fs, act;
        srt_second(P, key, srt_load(key, mcnt, mmin, mmax, ccnt, cmin, cmax), st1[key], (const uint8_t *)st1, fs, act);
        status[key] = fs;
        action[key] = act;
    }
    __syncthreads();
    // reverted list in key order + voxel scratch offsets
    if (threadIdx.x == 0) carry[0] = carry[1] = 0;
    __syncthreads();
    for (int base = 0; base < B; base += blockDim.x) {
        const int key = base + threadIdx.x;
        const bool rv = key < B && action[key] == 1;
        const uint32_t cap = rv ? (mcnt[key] + ccnt[key]) : 0u;
        uint32_t t0, t1;
        const uint32_t p0 = block_excl_scan(rv ? 1u : 0u, sm, t0);
        const uint32_t p1 = block_excl_scan(cap, sm, t1);
        const uint32_t c0 = carry[0], c1 = carry[1];
        if (key < B) rev_idx[key] = rv ? (c0 + p0) : 0xFFFFFFFFu;
        if (rv) {
            rev_list[c0 + p0] = (uint32_t)key;
            vox_off[c0 + p0] = c1 + p1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            carry[0] = c0 + t0;
            carry[1] = c1 + t1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st->n_rev = carry[0];
        st->vox_scratch_total = carry[1];
    }
}

// Up to 4096 bins: every thread owns FOUR CONSECUTIVE bins.  The statistics are loaded once (all loads in flight
// together), the first-pass status lives in LDS, and the reverted list needs two block scans in all.  This kernel sits on
// the main stream's dependency chain, where the general variant's nine dependent rounds of global accesses cost 20 us.
static constexpr int SRT_KPT = 4;
template <class S1P>
__device__ __forceinline__ void srt4_body(const DP &P, uint32_t *sm, S1P s_st1, const uint32_t *__restrict__ mcnt, const float *__restrict__ mmin,
                                                const float *__restrict__ mmax, const uint32_t *__restrict__ ccnt,
                                                const float *__restrict__ cmin, const float *__restrict__ cmax, uint8_t *__restrict__ st1,
                                                uint8_t *__restrict__ status, uint8_t *__restrict__ action, uint32_t *__restrict__ rev_idx,
                                                uint32_t *__restrict__ rev_list, uint32_t *__restrict__ vox_off, DevState *st,
                                                // the part of the output layout (k_layout4) that does not depend on R-GPF / the per-bin
                                                // voxelisation: offsets with the reverted bins counted as empty, how many reverted bins precede a
                                                // bin, curr_rejected offsets (final).  nullptr: not wanted (k_layout4 runs later)
                                                uint32_t *__restrict__ out_off0, uint32_t *__restrict__ rev_before,
                                                uint32_t *__restrict__ crej_off,
                                                // v3: the first pass has been done bin by bin in k_bin_stats_srt (status | 0x80 if the map bin is taller
                                                // than 0.5 m); nullptr: done here
                                                const uint8_t *__restrict__ st1_in,
                                                // fused launch (k_revert_bins_srt): a reverted bin's share of the voxel scratch begins at
                                                // moff[key] + qoff[key] (disjoint ranges of mc + cc entries, like the prefix of capacities, but
                                                // a per-bin workgroup knows it without a scan); nullptr: the prefix
                                                const uint32_t *__restrict__ moff_pos = nullptr, const uint32_t *__restrict__ qoff_pos = nullptr,
                                                // round 5, RESERVED layout (v3): every reverted bin keeps room for all it could become -- mc + cc
                                                // voxels, mc ground points --, so the offsets of everything else are known HERE, before R-GPF has
                                                // run: out_offR[key], gres_off[rk] (within the ground part), the late table (LateEnt: entry rk =
                                                // the bin's range, entry n_rev + rk = its ground range) and the extents in *st.  A bin whose points
                                                // may lie outside the NEXT step's VoI circle (outer radius beyond leave_lim) reserves twice: the
                                                // second half gives its points places in the outskirts' order.  nullptr: the dense layout only
                                                uint32_t *__restrict__ out_offR = nullptr, uint32_t *__restrict__ gres_off = nullptr,
                                                LateEnt *__restrict__ late = nullptr, double leave_lim = -1.0, const uint32_t *__restrict__ moff_all = nullptr) {
    const int B = P.B;
    const int k0 = threadIdx.x * SRT_KPT;
    BinStat bs[SRT_KPT];
    // (round 6: what the tail of this function used to fetch behind the scans -- a reverted bin's place in the voxel scratch, the size of
    // the complement -- is fetched HERE, beside the statistics: two dependent round trips less for a launch that is one workgroup of them)
    uint32_t vpos[SRT_KPT];
#pragma unroll
    for (int j = 0; j < SRT_KPT; ++j) vpos[j] = (moff_pos && k0 + j < B) ? moff_pos[k0 + j] + qoff_pos[k0 + j] : 0u;
    const uint32_t ncompl_pre = (out_offR && moff_all) ? moff_all[B + 1] - moff_all[B] : 0u;
    if (st1_in && P.version == 3) {
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j)
            if (k0 + j < B) {
                const uint8_t v = st1_in[k0 + j];
                bs[j].mc = mcnt[k0 + j];
                bs[j].cc = ccnt[k0 + j];
                bs[j].mmaxh = (v & 0x80u) ? 1.0 : 0.0;  // (all the second pass asks of the heights: is the map bin taller than 0.5 m?)
                bs[j].mminh = 0.0;
                bs[j].cmaxh = bs[j].cminh = 0.0;
                s_st1[k0 + j] = (uint8_t)(v & 0x7Fu);
            }
    } else {
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j)
            if (k0 + j < B) bs[j] = srt_load(k0 + j, mcnt, mmin, mmax, ccnt, cmin, cmax);
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j)
            if (k0 + j < B) {
                const uint8_t s = srt_first(P, bs[j]);
                s_st1[k0 + j] = s;
                st1[k0 + j] = s;
            }
    }
    __syncthreads();
    uint8_t act[SRT_KPT];
    uint32_t nrv = 0, ncap = 0;
#pragma unroll
    for (int j = 0; j < SRT_KPT; ++j) {
        act[j] = 0;
        if (k0 + j < B) {
            uint8_t fs;
            srt_second(P, k0 + j, bs[j], s_st1[k0 + j], s_st1, fs, act[j]);
            status[k0 + j] = fs;
            action[k0 + j] = act[j];
            if (act[j] == 1) {
                ++nrv;
                ncap += bs[j].mc + bs[j].cc;
            }
        }
    }
    // everything the prefixes are taken over, then ONE pass of six scans (round 6)
    uint32_t sz[SRT_KPT], cr[SRT_KPT], szR[SRT_KPT], gR[SRT_KPT];
    bool ml[SRT_KPT];
    uint32_t sums[6] = {nrv, ncap, 0u, 0u, 0u, 0u};  // reverted bins, their voxel scratch, dense sizes, curr_rejected, reserved sizes, reserved ground
    const double zb = fmax(fabs(P.min_h), fabs(P.max_h));
#pragma unroll
    for (int j = 0; j < SRT_KPT; ++j) {
        sz[j] = cr[j] = szR[j] = gR[j] = 0;
        ml[j] = false;
        if (k0 + j < B) {
            const uint32_t mc = bs[j].mc, cc = bs[j].cc;
            if (out_off0) {
                if (act[j] == 1) sz[j] = 0;             // curr + ground, voxelised (v3) or not (v2): known after R-GPF
                else if (act[j] == 2) sz[j] = cc + mc;  // merge_bins: curr then map (erasor.cpp:296-307)
                else if (act[j] == 3) sz[j] = cc;
                else {
                    sz[j] = mc;
                    if (act[j] == 4) cr[j] = cc;
                }
                sums[2] += sz[j];
                sums[3] += cr[j];
            }
            if (out_offR) {
                if (act[j] == 1) {
                    // can a point of this bin lie outside the next VoI circle?  Its egocentric radius is below the ring's outer edge (a
                    // voxel centroid is a mean of such points), |z| below zb; the sensor moves by what leave_lim has been reduced by
                    const int ring = (k0 + j) % P.R;
                    const double rho = (ring + 1 >= P.R) ? P.max_r : (double)(ring + 1) * P.ring_size;
                    ml[j] = !(sqrt(rho * rho + zb * zb) * (1.0 + 1e-5) < leave_lim);
                    szR[j] = cc > 0 ? (mc + cc) * (ml[j] ? 2u : 1u) : 0u;
                    gR[j] = mc * (ml[j] ? 2u : 1u);
                } else {
                    szR[j] = mc;  // (v3: a bin that is not reverted keeps its map points)
                }
                sums[4] += szR[j];
                sums[5] += gR[j];
            }
        }
    }
    uint32_t pre[6], tot[6];
    block_excl_scan_n<6>(sums, sm, pre, tot);
    {
        uint32_t p0 = pre[0], p1 = pre[1];
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j)
            if (k0 + j < B) {
                const bool rv = act[j] == 1;
                rev_idx[k0 + j] = rv ? p0 : 0xFFFFFFFFu;
                if (rv) {
                    rev_list[p0] = (uint32_t)(k0 + j);
                    vox_off[p0] = moff_pos ? vpos[j] : p1;
                    ++p0;
                    p1 += bs[j].mc + bs[j].cc;
                }
            }
    }
    if (threadIdx.x == 0) {
        st->n_rev = tot[0];
        st->vox_scratch_total = tot[1];
    }
    if (out_off0) {
        uint32_t p2 = pre[2], p3 = pre[3], pr = pre[0];
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j)
            if (k0 + j < B) {
                out_off0[k0 + j] = p2;
                crej_off[k0 + j] = p3;
                rev_before[k0 + j] = pr;
                p2 += sz[j];
                p3 += cr[j];
                pr += act[j] == 1 ? 1u : 0u;
            }
        if (threadIdx.x == 0) {
            st->total_bins0 = tot[2];
            st->n_curr_rejected = tot[3];
        }
    }
    if (out_offR) {
        const uint32_t t4 = tot[4], t5 = tot[5], t6 = tot[0];
        uint32_t p4 = pre[4], p5 = pre[5], pr = pre[0];
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j)
            if (k0 + j < B) {
                out_offR[k0 + j] = p4;
                if (act[j] == 1) {
                    const uint32_t mc = bs[j].mc, cc = bs[j].cc;
                    gres_off[pr] = p5;
                    LateEnt e;
                    e.start = p4;
                    e.ndata = cc > 0 ? mc + cc : 0u;
                    e.ntotal = szR[j];
                    e.actual = 0u;
                    late[pr] = e;
                    e.start = t4 + p5;
                    e.ndata = mc;
                    e.ntotal = gR[j];
                    late[t6 + pr] = e;
                    ++pr;
                }
                p4 += szR[j];
                p5 += gR[j];
            }
        if (threadIdx.x == 0) {
            const uint32_t ncompl = ncompl_pre;
            st->total_binsR = t4;
            st->ground_res = t5;
            st->n_late = 2u * t6;
            st->n_compl = ncompl;
            st->nF_new = t4 + t5 + ncompl;  // (an EXTENT; nF_valid_new follows when the reverted bins are through: k_assemble_late)
        }
    }
}
__global__ __launch_bounds__(1024) void k_srt4(DP P, const uint32_t *__restrict__ mcnt, const float *__restrict__ mmin,
                                                const float *__restrict__ mmax, const uint32_t *__restrict__ ccnt,
                                                const float *__restrict__ cmin, const float *__restrict__ cmax, uint8_t *__restrict__ st1,
                                                uint8_t *__restrict__ status, uint8_t *__restrict__ action, uint32_t *__restrict__ rev_idx,
                                                uint32_t *__restrict__ rev_list, uint32_t *__restrict__ vox_off, DevState *st,
                                                uint32_t *__restrict__ out_off0, uint32_t *__restrict__ rev_before,
                                                uint32_t *__restrict__ crej_off, const uint8_t *__restrict__ st1_in,
                                                // round 5: the bins' shares of the voxel scratch by position (the per-bin launch finds its bins
                                                // itself, see k_revert_bins_srt) and the RESERVED layout (see srt4_body); nullptr: neither
                                                const uint32_t *__restrict__ moff_pos = nullptr, const uint32_t *__restrict__ qoff_pos = nullptr,
                                                uint32_t *__restrict__ out_offR = nullptr, uint32_t *__restrict__ gres_off = nullptr,
                                                LateEnt *__restrict__ late = nullptr, double leave_lim = -1.0) {
    __shared__ uint32_t sm[96];  // (six scans' wavefront totals, see block_excl_scan_n)
    __shared__ uint8_t s_st1[1024 * SRT_KPT];
    CHAIN_STAMP(3);
    srt4_body(P, sm, s_st1, mcnt, mmin, mmax, ccnt, cmin, cmax, st1, status, action, rev_idx, rev_list, vox_off, st, out_off0, rev_before, crej_off,
              st1_in, moff_pos, qoff_pos, out_offR, gres_off, late, leave_lim, moff_pos);
}

// Round 4: which bin is entry `rk` of the reverted list -- WITHOUT the list.  In v3 a bin is reverted iff its first-pass status
// (k_bin_stats_srt: st1b) is MAP_IS_HIGHER and the map bin is taller than 0.5 m (erasor.cpp:510-511): a decision local to the bin.  Every
// workgroup of the fused per-bin launch therefore finds its own bin from the 2160 status bytes (every thread owns SRT_KPT consecutive
// bins, two block scans) while ONE extra workgroup of the same launch does k_srt4's work for the kernels behind it (status / action /
// layout prefixes): the Scan Ratio Test's second pass leaves the main stream's dependency chain (14 us + a kernel boundary).
// sel[0] = bin key, sel[2] = number of reverted bins (the bin's share of the voxel scratch begins at moff[key] + qoff[key]).
// Round 6: sel[4 .. 7] = the bin's range of the bucketed map (begin, points) and of the bucketed scan -- the few threads that own a reverted
// bin fetch their bins' offsets BESIDE the block scan, so the workgroup does not pay a round trip of its own for them behind it.
__shared__ uint32_t g_sel[8];
__shared__ uint32_t g_selsm[40];
// (out of line, like the sort: a register allocation of its own instead of a share of the per-bin kernel's 128 VGPRs)
__device__ __attribute__((noinline)) void rev_select_call(int B, const uint8_t *__restrict__ st1b, uint32_t rk, const uint32_t *__restrict__ moff,
                                                          const uint32_t *__restrict__ qoff) {
    uint32_t *sm = g_selsm, *sel = g_sel;
    const int k0 = threadIdx.x * SRT_KPT;
    uint32_t rvm = 0;
    if (k0 < B) {
        uint32_t w = 0;
        if (k0 + SRT_KPT <= B) w = *reinterpret_cast<const uint32_t *>(st1b + k0);  // (k0 is a multiple of four: one aligned load)
        else
            for (int j = 0; k0 + j < B; ++j) w |= (uint32_t)st1b[k0 + j] << (8 * j);
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j) {
            const uint32_t v = (w >> (8 * j)) & 0xFFu;
            if (k0 + j < B && (v & 0x7Fu) == ST_MAP && (v & 0x80u)) rvm |= 1u << j;
        }
    }
    const uint32_t nrv = (uint32_t)__popc(rvm);
    uint32_t mo[SRT_KPT + 1], qo[SRT_KPT + 1];
#pragma unroll
    for (int j = 0; j <= SRT_KPT; ++j) mo[j] = qo[j] = 0u;
    if (rvm) {  // (in flight during the scan)
#pragma unroll
        for (int j = 0; j <= SRT_KPT; ++j) {
            const int k = min(k0 + j, B);
            mo[j] = moff[k];
            qo[j] = qoff[k];
        }
    }
    uint32_t t0;
    uint32_t p0 = block_excl_scan(nrv, sm, t0);
    if (threadIdx.x == 0) sel[2] = t0;
    if (rk >= p0 && rk < p0 + nrv) {
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j)
            if ((rvm >> j) & 1u) {
                if (p0 == rk) {
                    sel[0] = (uint32_t)(k0 + j);
                    sel[4] = mo[j];
                    sel[5] = mo[j + 1] - mo[j];
                    sel[6] = qo[j];
                    sel[7] = qo[j + 1] - qo[j];
                }
                ++p0;
            }
    }
    __syncthreads();
}

// ================================================================================================
// R-GPF (extract_ground, erasor.cpp:233-294; estimate_plane_ :183-198; seeds :204-231).
// One workgroup per reverted bin.
// ================================================================================================
struct Rot {
    float c, s;
};
__device__ __forceinline__ Rot make_jacobi(float x, float y, float z) {  // Eigen 3.3 JacobiRotation::makeJacobi
    Rot r;
    const float deno = 2.0f * fabsf(y);
    if (deno < 1.17549435e-38f) {
        r.c = 1.f;
        r.s = 0.f;
    } else {
        const float tau = (x - z) / deno;
        const float w = sqrtf(tau * tau + 1.0f);
        // (round 3: ONE division for t -- the denominator is selected first, the quotient is the same either way -- and none for
        // y / |y|, which is copysign(1, y) for every finite y that gets here (|y| >= 2^-127: no zero, no denormal quotient) and NaN
        // for an infinite or NaN y like the quotient: an IEEE division is ~12 dependent instructions of the one lane that runs this)
        const float t = 1.0f / (tau > 0.f ? tau + w : tau - w);
        const float sign_t = t > 0.f ? 1.0f : -1.0f;
        const float n = 1.0f / sqrtf(t * t + 1.0f);
        const float unit_y = (fabsf(y) < __int_as_float(0x7F800000)) ? copysignf(1.0f, y) : __int_as_float(0x7FC00000);
        r.s = -sign_t * unit_y * fabsf(t) * n;
        r.c = n;
    }
    return r;
}
__device__ __forceinline__ void rot_apply(float &x, float &y, float c, float s) {
    const float xi = x, yi = y;
    x = c * xi + s * yi;
    y = -s * xi + c * yi;
}
// Eigen 3.3 JacobiSVD<MatrixXf>(3x3, ComputeFullU): U (row-major) and singular values.
// One (p, q) step of a sweep with COMPILE-TIME indices: W and U are nine scalars each and stay in registers (with run-time
// indices the two arrays lived in scratch memory: every access a ~1 us round trip, 6.5 us per decomposition on one lane).
template <int PP, int QQ>
__device__ __forceinline__ void jacobi_step(float (&W)[9], float (&U)[9], float &maxDiag, bool &finished) {
    const float precision = 2.0f * 1.1920929e-07f;
    const float considerAsZero = 1.17549435e-38f;
    const float threshold = fmaxf(considerAsZero, precision * maxDiag);
    if (fabsf(W[PP * 3 + QQ]) > threshold || fabsf(W[QQ * 3 + PP]) > threshold) {
        finished = false;
        float m00 = W[PP * 3 + PP], m01 = W[PP * 3 + QQ], m10 = W[QQ * 3 + PP], m11 = W[QQ * 3 + QQ];
        Rot rot1;
        const float t = m00 + m11;
        const float d = m10 - m01;
        if (fabsf(d) < 1.17549435e-38f) {
            rot1.s = 0.f;
            rot1.c = 1.f;
        } else {
            const float u = t / d;
            const float tmp = sqrtf(1.0f + u * u);
            rot1.s = 1.0f / tmp;
            rot1.c = u / tmp;
        }
        if (!(rot1.c == 1.f && rot1.s == 0.f)) {
            rot_apply(m00, m10, rot1.c, rot1.s);
            rot_apply(m01, m11, rot1.c, rot1.s);
        }
        const Rot jr = make_jacobi(m00, m01, m11);
        const float oc = jr.c, os = -jr.s;
        Rot jl;
        jl.c = rot1.c * oc - rot1.s * os;
        jl.s = rot1.c * os + rot1.s * oc;
        if (!(jl.c == 1.f && jl.s == 0.f)) {
#pragma unroll
            for (int col = 0; col < 3; ++col) rot_apply(W[PP * 3 + col], W[QQ * 3 + col], jl.c, jl.s);
#pragma unroll
            for (int row = 0; row < 3; ++row) rot_apply(U[row * 3 + PP], U[row * 3 + QQ], jl.c, jl.s);
        }
        if (!(jr.c == 1.f && -jr.s == 0.f)) {
#pragma unroll
            for (int row = 0; row < 3; ++row) rot_apply(W[row * 3 + PP], W[row * 3 + QQ], jr.c, -jr.s);
        }
        maxDiag = fmaxf(maxDiag, fmaxf(fabsf(W[PP * 3 + PP]), fabsf(W[QQ * 3 + QQ])));
    }
}
template <int A, int B>
__device__ __forceinline__ void svd_swap_cols(float (&sv)[3], float (&U)[9]) {
    const float ts = sv[A];
    sv[A] = sv[B];
    sv[B] = ts;
#pragma unroll
    for (int row = 0; row < 3; ++row) {
        const float tu = U[row * 3 + A];
        U[row * 3 + A] = U[row * 3 + B];
        U[row * 3 + B] = tu;
    }
}
__device__ __forceinline__ void jacobi_svd3(const float (&cov)[9], float (&U)[9], float (&sv)[3]) {
    float scale = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) scale = fmaxf(scale, fabsf(cov[k]));
    if (scale == 0.f) scale = 1.f;
    float W[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) W[k] = cov[k] / scale;
#pragma unroll
    for (int k = 0; k < 9; ++k) U[k] = (k % 4 == 0) ? 1.f : 0.f;
    float maxDiag = fmaxf(fabsf(W[0]), fmaxf(fabsf(W[4]), fabsf(W[8])));
    bool finished = false;
    int guard = 0;
    while (!finished && guard++ < 1000) {  // sweeps: (p, q) = (1, 0), (2, 0), (2, 1)
        finished = true;
        jacobi_step<1, 0>(W, U, maxDiag, finished);
        jacobi_step<2, 0>(W, U, maxDiag, finished);
        jacobi_step<2, 1>(W, U, maxDiag, finished);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float a = W[i * 3 + i];
        sv[i] = fabsf(a);
        if (a < 0.f) {
#pragma unroll
            for (int row = 0; row < 3; ++row) U[row * 3 + i] = -U[row * 3 + i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    // singular values in decreasing order (Eigen: selection by maxCoeff over the tail, first maximum wins; stops at a zero maximum)
    {
        int pos = 0;
        float mx = sv[0];
        if (sv[1] > mx) { mx = sv[1]; pos = 1; }
        if (sv[2] > mx) { mx = sv[2]; pos = 2; }
        if (mx == 0.f) return;
        if (pos == 1) svd_swap_cols<0, 1>(sv, U);
        else if (pos == 2) svd_swap_cols<0, 2>(sv, U);
    }
    {
        float mx = sv[1];
        bool second = false;
        if (sv[2] > mx) { mx = sv[2]; second = true; }
        if (mx == 0.f) return;
        if (second) svd_swap_cols<1, 2>(sv, U);
    }
}

static constexpr uint32_t RG_LMAX = 4096;
static constexpr uint32_t RG_CH = 1024;  // covariance products are staged in LDS in chunks of this many list elements

// After the exact z-sort: seeds, gf_iter x (plane fit, classification).  sortedV: sorted order (bin-local indices);
// glist: scratch for the current ground list.  Templated so that LDS callers get ds_* instructions.
template <class SV, class GL>
__device__ __forceinline__ void rgpf_after_sort(const DP &P, const float4 *__restrict__ pts, uint32_t M, uint32_t o0, uint32_t rk, SV sortedV, GL glist,
                                                uint32_t *sm, float *sProd, float *s_n, double *s_thp, double *s_lprp, uint32_t *s_carryp,
                                                uint8_t *__restrict__ gflag, uint32_t *__restrict__ grank, uint32_t *__restrict__ glist_out,
                                                uint32_t *__restrict__ ng_out, float *__restrict__ plane_n, double *__restrict__ plane_d,
                                                Counters *ctr) {
    const uint32_t tid = threadIdx.x, bs = blockDim.x, lane = tid & 63u, wave = tid >> 6;
    double &s_th = *s_thp;
    double &s_lpr = *s_lprp;
    uint32_t &s_carry = *s_carryp;
    // --- drop leading z < min_h (erasor.cpp:242-251); monotone in sorted order ---
    uint32_t cnt = 0;
    for (uint32_t k = tid; k < M; k += bs) cnt += ((double)pts[sortedV[k]].z < P.min_h) ? 1u : 0u;
    {
        uint32_t tot;
        block_excl_scan(cnt, sm, tot);
        cnt = tot;
    }
    const uint32_t drop = cnt, Ms = M - drop;
    // --- extract_initial_seeds_ (erasor.cpp:204-231) ---
    {   // the (at most gf_lpr) lowest z values are FETCHED in parallel (a dependent global load each: ~0.7 us apiece when one
        // thread walks them), then added by one thread in the reference's order
        uint32_t cl = 0;
        if (P.num_lowest >= 0 && Ms > (uint32_t)P.num_lowest && P.gf_lpr > 0) cl = min((uint32_t)P.gf_lpr, Ms - (uint32_t)P.num_lowest);
        cl = min(cl, 9u * RG_CH);
        for (uint32_t t = tid; t < cl; t += bs) sProd[t] = pts[sortedV[drop + (uint32_t)P.num_lowest + t]].z;
        __syncthreads();
        if (tid == 0) {
            double sum = 0;
            for (uint32_t t = 0; t < cl; ++t) sum += (double)sProd[t];
            s_lpr = cl != 0 ? sum / (int)cl : 0;
        }
    }
    __syncthreads();
    const double seed_thr = s_lpr + P.gf_seeds_h;
    cnt = 0;
    for (uint32_t k = tid; k < Ms; k += bs) cnt += ((double)pts[sortedV[drop + k]].z < seed_thr) ? 1u : 0u;
    {
        uint32_t tot;
        block_excl_scan(cnt, sm, tot);
        cnt = tot;
    }
    uint32_t ng = cnt;  // seeds = the first ng of the sorted points (predicate is monotone in z)
    __syncthreads();
    for (uint32_t k = tid; k < ng; k += bs) glist[k] = sortedV[drop + k];
    __threadfence_block();
    __syncthreads();
    for (int it = 0; it < P.gf_iter; ++it) {
        // --- estimate_plane_: pcl::computeMeanAndCovarianceMatrix, nine float32 accumulators in list order ---
        // The nine products of every list element are formed by ALL threads (parallel, order-free) into LDS; then lane a
        // of wave 0 adds row a strictly in list order — the only part that has to be sequential (float32 addition order
        // is what PCL's result depends on).  x*x etc. are single IEEE multiplies either way.
        float acc = 0.f;
        for (uint32_t cb = 0; cb < ng; cb += RG_CH) {
            const uint32_t cn = min(RG_CH, ng - cb);
            for (uint32_t t = tid; t < cn; t += bs) {
                const float4 q = pts[glist[cb + t]];
                sProd[0 * RG_CH + t] = q.x * q.x;
                sProd[1 * RG_CH + t] = q.x * q.y;
                sProd[2 * RG_CH + t] = q.x * q.z;
                sProd[3 * RG_CH + t] = q.y * q.y;
                sProd[4 * RG_CH + t] = q.y * q.z;
                sProd[5 * RG_CH + t] = q.z * q.z;
                sProd[6 * RG_CH + t] = q.x;
                sProd[7 * RG_CH + t] = q.y;
                sProd[8 * RG_CH + t] = q.z;
            }
            __syncthreads();
            if (wave == 0 && lane < 9) {
                const float *row = sProd + lane * RG_CH;
#pragma unroll 8
                for (uint32_t k = 0; k < cn; ++k) acc += row[k];
            }
            __syncthreads();
        }
        if (wave == 0) {
            float a[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) a[k] = __shfl(acc, k, 64);
            if (lane == 0) {
                float cov[9], mean[3], U[9], sv[3];
                if (ng == 0) {
                    for (int k = 0; k < 9; ++k) cov[k] = 0.f;
                    mean[0] = mean[1] = mean[2] = 0.f;
                    atomicAdd(&ctr->n_degenerate, 1u);
                } else {
                    const float fn = (float)ng;
                    for (int k = 0; k < 9; ++k) a[k] /= fn;
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
                const float n0 = U[2], n1 = U[5], n2 = U[8];
                const float dot = (n0 * mean[0] + n1 * mean[1]) + n2 * mean[2];
                const double d = -dot;
                s_n[0] = n0;
                s_n[1] = n1;
                s_n[2] = n2;
                s_th = P.gf_dist - d;
                plane_n[((size_t)rk * P.gf_iter + it) * 3 + 0] = n0;
                plane_n[((size_t)rk * P.gf_iter + it) * 3 + 1] = n1;
                plane_n[((size_t)rk * P.gf_iter + it) * 3 + 2] = n2;
                plane_d[(size_t)rk * P.gf_iter + it] = d;
            }
        }
        __syncthreads();
        // --- points * normal_ < th_dist_d_ in source order (erasor.cpp:265-281) ---
        const float n0 = s_n[0], n1 = s_n[1], n2 = s_n[2];
        const double th = s_th;
        const bool last = it == P.gf_iter - 1;
        if (tid == 0) s_carry = 0;
        __syncthreads();
        for (uint32_t base = 0; base < M; base += bs) {
            const uint32_t i = base + tid;
            bool g = false;
            if (i < M) {
                const float4 p = pts[i];
                const float res = (p.x * n0 + p.y * n1) + p.z * n2;
                g = (double)res < th;
            }
            uint32_t tot;
            const uint32_t pre = block_excl_scan(g ? 1u : 0u, sm, tot);
            const uint32_t c0 = s_carry;
            if (i < M) {
                if (g) glist[c0 + pre] = i;
                if (last) {
                    gflag[o0 + i] = g ? 1 : 0;
                    grank[o0 + i] = g ? (c0 + pre) : (i - (c0 + pre));  // rank among ground / among rejected
                }
            }
            __syncthreads();
            if (tid == 0) s_carry = c0 + tot;
            __syncthreads();
        }
        ng = s_carry;
        __threadfence_block();
        __syncthreads();
    }
    for (uint32_t k = tid; k < ng; k += bs) glist_out[o0 + k] = glist[k];
    if (tid == 0) ng_out[rk] = ng;
}

// ================================================================================================
// per-bin voxelize_preserving_labels(curr points + reverted ground, /erasor/map_voxel_size) —
// erasor.cpp:523-528.  One workgroup per reverted bin.
// ================================================================================================
static constexpr uint32_t BV_LMAX = 2048;

// per-bin voxelisation for clouds beyond the LDS-resident size (global scratch), templated on the scratch pointers so that the LDS instantiation compiles to ds_* instructions
template <class KP, class VP, class K2P, class V2P, class CP, class PP, class HP>
__device__ __forceinline__ void binvox_core(const DP &P, uint32_t m, uint32_t nc, const float4 *__restrict__ sqb, const float4 *__restrict__ sptb,
                                            const uint32_t *__restrict__ glb, KP K, VP V, K2P K2, V2P V2, CP C, PP posL, PP posR, HP head,
                                            esort::Seg *qa, esort::Seg *qb, uint32_t *qcnt, uint32_t *sm, uint32_t *sbb, uint32_t *s_carryp,
                                            float4 *__restrict__ vout, uint32_t *__restrict__ nvox_slot, Counters *ctr) {
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    uint32_t &s_carry = *s_carryp;
    // input cloud of this call: curr bin points (scan order) then the reverted ground (source order)
    for (uint32_t j = tid; j < m; j += bs) C[j] = j < nc ? sqb[j] : sptb[glb[j - nc]];
    if (tid < 3) sbb[tid] = 0xFFFFFFFFu;
    if (tid >= 3 && tid < 6) sbb[tid] = 0u;
    __threadfence_block();
    __syncthreads();
    {
        uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
        for (uint32_t j = tid; j < m; j += bs) {
            const float4 p = C[j];
            const uint32_t k3[3] = {fkey_ord(p.x), fkey_ord(p.y), fkey_ord(p.z)};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                mn[a] = k3[a] < mn[a] ? k3[a] : mn[a];
                mx[a] = k3[a] > mx[a] ? k3[a] : mx[a];
            }
        }
        // wavefront reduction first: one LDS atomic per wavefront and extreme (1024 threads hitting six LDS words serialise)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t0 = __shfl_down(mn[a], o, 64), t1 = __shfl_down(mx[a], o, 64);
                mn[a] = t0 < mn[a] ? t0 : mn[a];
                mx[a] = t1 > mx[a] ? t1 : mx[a];
            }
            if ((tid & 63u) == 0) {
                atomicMin(&sbb[a], mn[a]);
                atomicMax(&sbb[3 + a], mx[a]);
            }
        }
    }
    __syncthreads();
    const float mn[3] = {fkey_inv(sbb[0]), fkey_inv(sbb[1]), fkey_inv(sbb[2])};
    const float mx[3] = {fkey_inv(sbb[3]), fkey_inv(sbb[4]), fkey_inv(sbb[5])};
    const VoxGrid g = vox_grid_from_bbox(mn, mx, P.leaf_map);
    if (g.overflow) {  // VoxelGrid returns the input unchanged; not supported on device -> flagged, host fails the step
        if (tid == 0) {
            atomicAdd(&ctr->n_voxel_overflow, 1u);
            ctr->err = 1;
            *nvox_slot = 0;
        }
        return;
    }
    for (uint32_t j = tid; j < m; j += bs) {
        const float4 p = C[j];
        K[j] = vox_index(g, p.x, p.y, p.z);
        V[j] = j;
    }
    __threadfence_block();
    __syncthreads();
    esort::block_esort(K, V, posL, posR, head, K2, V2, 0u, m, 2 * esort::lg2_floor(m), qa, qb, qcnt, (uint32_t)(BV_LMAX / 16 + 2),
                       &ctr->n_sort_fallback, &ctr->sort_qoverflow);
    __threadfence_block();
    __syncthreads();
    // phase A: runs -> centroids (CentroidPoint float sums in sorted order).  K / V are dead: reuse their storage for the
    // centroid x / y of voxel v (v <= position of its run head, and slots below the current tile are no longer read);
    // cz is parked in the output slot.
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < m; base += bs) {
        const uint32_t i = base + tid;
        const bool hd = i < m && (i == 0 || (uint32_t)K2[i] != (uint32_t)K2[i - 1]);
        uint32_t tot;
        const uint32_t pre = block_excl_scan(hd ? 1u : 0u, sm, tot);
        const uint32_t c0 = s_carry;
        if (hd) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
            uint32_t e = i;
            const uint32_t kk = K2[i];
            while (e < m && (uint32_t)K2[e] == kk) {
                const float4 p = C[V2[e]];
                sx += p.x;
                sy += p.y;
                sz += p.z;  // the averaged intensity is overwritten by the nearest input point's label (utils.cpp:109)
                ++e;
            }
            const float c = (float)(e - i);
            const uint32_t v = c0 + pre;
            K[v] = __float_as_uint(sx / c);
            V[v] = __float_as_uint(sy / c);
            vout[v].z = sz / c;
        }
        __syncthreads();
        if (tid == 0) s_carry = c0 + tot;
        __syncthreads();
    }
    const uint32_t nv = s_carry;
    __threadfence_block();
    __syncthreads();
    // phase B: exact 1-NN of every centroid over all inputs of this call (lowest index on float ties)
    // (a few hundred voxels against ~1000 points: up to eight lanes share a voxel's candidate list, merged by (distance, index))
    uint32_t S = 1;
    while (S < 8 && (uint64_t)nv * S * 2 <= bs) S <<= 1;
    for (uint32_t v0 = 0; v0 < nv; v0 += bs / S) {
        const uint32_t v = v0 + tid / S, part = tid % S;
        if (v >= nv) continue;  // (all S lanes of a voxel take the same branch)
        const float cx = __uint_as_float(K[v]), cy = __uint_as_float(V[v]), cz = vout[v].z;
        float best = __int_as_float(0x7F800000);
        uint32_t best_j = 0xFFFFFFFFu;
        for (uint32_t j = part; j < m; j += S) {
            const float4 p = C[j];
            const float dd = l2_simple(cx, cy, cz, p.x, p.y, p.z);
            if (dd < best) {
                best = dd;
                best_j = j;
            }
        }
        for (uint32_t o = S >> 1; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, (int)o, 64);
            const uint32_t oj = __shfl_xor(best_j, (int)o, 64);
            if (ob < best || (ob == best && oj < best_j)) {
                best = ob;
                best_j = oj;
            }
        }
        if (part == 0) {
            const float4 pb = C[best_j < m ? best_j : 0u];
            vout[v] = make_float4(cx, cy, cz, pb.w);
        }
    }
    if (tid == 0) *nvox_slot = nv;
}

#include "revert_bins.hip.h"

// ================================================================================================
// output layout of the new F region (get_static_estimate erasor.cpp:612-626; OMU.cpp:281-290):
//   [selected bins, theta-major | ground_viz (reverted bins' ground, again) | complement]
// ================================================================================================
__global__ __launch_bounds__(1024) void k_layout(DP P, const uint8_t *__restrict__ action, const uint32_t *__restrict__ rev_idx,
                                                  const uint32_t *__restrict__ mcnt, const uint32_t *__restrict__ ccnt,
                                                  const uint32_t *__restrict__ moff, const uint32_t *__restrict__ nvox,
                                                  const uint32_t *__restrict__ ng_arr, uint32_t *__restrict__ out_off,
                                                  uint32_t *__restrict__ ground_off, uint32_t *__restrict__ rej_off,
                                                  uint32_t *__restrict__ crej_off, DevState *st) {
    __shared__ uint32_t sm[40];
    __shared__ uint32_t carry[4];
    const int B = P.B;
    if (threadIdx.x == 0) carry[0] = carry[1] = carry[2] = carry[3] = 0;
    __syncthreads();
    for (int base = 0; base < B; base += blockDim.x) {
        const int key = base + threadIdx.x;
        uint32_t sz = 0, g = 0, rj = 0, cr = 0;
        if (key < B) {
            const uint8_t act = action[key];
            const uint32_t mc = mcnt[key], cc = ccnt[key];
            if (act == 1) {
                const uint32_t rk = rev_idx[key];
                g = ng_arr[rk];
                rj = mc - g;
                if (P.version == 3) sz = cc > 0 ? nvox[rk] : 0u;
                else sz = cc > 0 ? (cc + g) : 0u;  // v2: curr points then ground, no voxelisation (erasor.cpp:384-392)
            } else if (act == 2) {
                sz = cc + mc;  // merge_bins: curr then map (erasor.cpp:296-307)
            } else if (act == 3) {
                sz = cc;
            } else {
                sz = mc;
                if (act == 4) cr = cc;
            }
        }
        uint32_t t0, t1, t2, t3;
        const uint32_t p0 = block_excl_scan(sz, sm, t0);
        const uint32_t p1 = block_excl_scan(g, sm, t1);
        const uint32_t p2 = block_excl_scan(rj, sm, t2);
        const uint32_t p3 = block_excl_scan(cr, sm, t3);
        const uint32_t c0 = carry[0], c1 = carry[1], c2 = carry[2], c3 = carry[3];
        if (key < B) {
            out_off[key] = c0 + p0;
            crej_off[key] = c3 + p3;
            if (action[key] == 1) {
                const uint32_t rk = rev_idx[key];
                ground_off[rk] = c1 + p1;  // relative to total_bins
                rej_off[rk] = c2 + p2;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            carry[0] = c0 + t0;
            carry[1] = c1 + t1;
            carry[2] = c2 + t2;
            carry[3] = c3 + t3;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t ncompl = moff[B + 1] - moff[B];
        st->total_bins = carry[0];
        st->n_ground = carry[1];
        st->n_rejected = carry[2];
        st->n_curr_rejected = carry[3];
        st->n_static_est = carry[0] + carry[1];
        st->n_compl = ncompl;
        st->nF_new = carry[0] + carry[1] + ncompl;
        st->nF_valid_new = carry[0] + carry[1] + ncompl;
    }
}

// Up to 4096 bins: four consecutive bins per thread, four block scans in all (see k_srt4)
__global__ __launch_bounds__(1024) void k_layout4(DP P, const uint8_t *__restrict__ action, const uint32_t *__restrict__ rev_idx,
                                                   const uint32_t *__restrict__ mcnt, const uint32_t *__restrict__ ccnt,
                                                   const uint32_t *__restrict__ moff, const uint32_t *__restrict__ nvox,
                                                   const uint32_t *__restrict__ ng_arr, uint32_t *__restrict__ out_off,
                                                   uint32_t *__restrict__ ground_off, uint32_t *__restrict__ rej_off,
                                                   uint32_t *__restrict__ crej_off, DevState *st) {
    __shared__ uint32_t sm[40];
    const int B = P.B;
    const int k0 = threadIdx.x * SRT_KPT;
    uint8_t act[SRT_KPT];
    uint32_t rk[SRT_KPT], sz[SRT_KPT], g[SRT_KPT], rj[SRT_KPT], cr[SRT_KPT];
    uint32_t ssz = 0, sg = 0, srj = 0, scr = 0;
#pragma unroll
    for (int j = 0; j < SRT_KPT; ++j) {
        act[j] = 0;
        rk[j] = 0;
        sz[j] = g[j] = rj[j] = cr[j] = 0;
        if (k0 + j < B) {
            const int key = k0 + j;
            act[j] = action[key];
            const uint32_t mc = mcnt[key], cc = ccnt[key];
            if (act[j] == 1) {
                rk[j] = rev_idx[key];
                g[j] = ng_arr[rk[j]];
                rj[j] = mc - g[j];
                if (P.version == 3) sz[j] = cc > 0 ? nvox[rk[j]] : 0u;
                else sz[j] = cc > 0 ? (cc + g[j]) : 0u;  // v2: curr points then ground, no voxelisation (erasor.cpp:384-392)
            } else if (act[j] == 2) {
                sz[j] = cc + mc;  // merge_bins: curr then map (erasor.cpp:296-307)
            } else if (act[j] == 3) {
                sz[j] = cc;
            } else {
                sz[j] = mc;
                if (act[j] == 4) cr[j] = cc;
            }
            ssz += sz[j];
            sg += g[j];
            srj += rj[j];
            scr += cr[j];
        }
    }
    uint32_t t0, t1, t2, t3;
    uint32_t p0 = block_excl_scan(ssz, sm, t0);
    uint32_t p1 = block_excl_scan(sg, sm, t1);
    uint32_t p2 = block_excl_scan(srj, sm, t2);
    uint32_t p3 = block_excl_scan(scr, sm, t3);
#pragma unroll
    for (int j = 0; j < SRT_KPT; ++j)
        if (k0 + j < B) {
            out_off[k0 + j] = p0;
            crej_off[k0 + j] = p3;
            if (act[j] == 1) {
                ground_off[rk[j]] = p1;  // relative to total_bins
                rej_off[rk[j]] = p2;
            }
            p0 += sz[j];
            p1 += g[j];
            p2 += rj[j];
            p3 += cr[j];
        }
    if (threadIdx.x == 0) {
        const uint32_t ncompl = moff[B + 1] - moff[B];
        st->total_bins = t0;
        st->n_ground = t1;
        st->n_rejected = t2;
        st->n_curr_rejected = t3;
        st->n_static_est = t0 + t1;
        st->n_compl = ncompl;
        st->nF_new = t0 + t1 + ncompl;
        st->nF_valid_new = t0 + t1 + ncompl;
    }
}

// one thread per sorted VoI point.  XFORM: apply tf_body2origin_ (map write-back) or keep egocentric
// coordinates (the clouds get_static_estimate / get_outliers hand out).
// parse_dynamic_obj as counters (utils.cpp:57-78) over the points a workgroup wrote to Fnew: per-thread tallies (d, s),
// one pair of device-scope atomics per workgroup (same-address atomics serialise: keep them few)
__device__ __forceinline__ void block_commit_labels(uint32_t d, uint32_t s, unsigned long long *cnt /* [16][8]: {static, dynamic, pad} */) {
    __shared__ uint32_t sd[16], ss[16];
    if (!cnt) return;
    d = wave_sum(d);
    s = wave_sum(s);
    if ((threadIdx.x & 63u) == 0) {
        sd[threadIdx.x >> 6] = d;
        ss[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t td = 0, ts = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
            td += sd[w];
            ts += ss[w];
        }
        // 16 slots, one cache line each: same-address atomics serialise (~6 ns apiece), k_step_end adds the slots up
        unsigned long long *slot = cnt + (blockIdx.x & 15u) * 8u;
        if (td) atomicAdd(slot + 1, (unsigned long long)td);
        if (ts) atomicAdd(slot, (unsigned long long)ts);
    }
}

// FOLD: the output layout is finished HERE instead of in a launch of its own (k_layout4, 8.8 us on the dependency chain): k_srt4 has
// left the offsets with the reverted bins counted as empty (out_off0), the number of reverted bins before every bin (rev_before)
// and the size of that part (st->total_bins0); every workgroup adds what R-GPF / the per-bin voxelisation have produced since --
// exclusive prefixes over the (few) reverted bins of their output sizes, their ground and their rejected points, in LDS -- and
// workgroup 0 also leaves the final tables and totals for the getters and k_step_end.
static constexpr uint32_t ASM_RVMAX = 1024;  // reverted bins whose prefixes fit the LDS tables (more: summed on the fly, correct and slow)
// RES (round 5): the EARLY half of a write-back in the reserved layout (see srt4_body): everything that does not wait for the per-bin launch
// -- the bins that are not reverted, the complement -- at the offsets `out_off0` (= out_offR here) and the extents in *st; the reverted
// bins' voxels, ground and rejected points follow in k_assemble_late.  No FOLD, no tail.
template <bool XFORM, bool FOLD, bool RES = false>
__global__ __launch_bounds__(256) void k_assemble_map(DP P, Xf Tb2o, const uint8_t *__restrict__ action,
                                                       const uint32_t *__restrict__ rev_idx, const uint32_t *__restrict__ skeys,
                                                       const float4 *__restrict__ spts, const uint32_t *__restrict__ ssrc,
                                                       const uint32_t *__restrict__ moff, const uint32_t *__restrict__ ccnt,
                                                       const uint8_t *__restrict__ gflag, const uint32_t *__restrict__ grank,
                                                       uint32_t *__restrict__ out_off, uint32_t *__restrict__ ground_off,
                                                       uint32_t *__restrict__ rej_off, DevState *st,
                                                       float4 *__restrict__ Fnew, float4 *__restrict__ rejected,
                                                       uint32_t *__restrict__ rejected_src, unsigned long long *cnt,
                                                       // v3 only: the last `tail` workgroups write the voxelised reverted bins (k_assemble_bins' v3
                                                       // branch) from the reverted-bin list -- one launch less on the dependency chain
                                                       uint32_t tail, const uint32_t *__restrict__ rev_list, const uint32_t *__restrict__ qoff,
                                                       const uint32_t *__restrict__ nvox, const uint32_t *__restrict__ vox_off,
                                                       const float4 *__restrict__ vox_out,
                                                       // FOLD only
                                                       const uint32_t *__restrict__ out_off0, const uint32_t *__restrict__ rev_before,
                                                       const uint32_t *__restrict__ ng_arr) {
    __shared__ uint32_t s_cv[FOLD ? ASM_RVMAX + 1 : 1], s_cg[FOLD ? ASM_RVMAX + 1 : 1], s_cr[FOLD ? ASM_RVMAX + 1 : 1];
    __shared__ uint32_t s_sm[40];
    __shared__ uint32_t s_carry[3];
    uint32_t total_bins = 0, n_static_est = 0, n_rev_all = 0;
    CHAIN_STAMP(4);
    // size of reverted bin rk in the output, its ground and its rejected points
    auto rev_sizes = [&](uint32_t rk, uint32_t &sz, uint32_t &g, uint32_t &rj) {
        const uint32_t key = rev_list[rk];
        const uint32_t cc = ccnt[key], mc = moff[key + 1] - moff[key];
        g = ng_arr[rk];
        rj = mc - g;
        sz = cc > 0 ? (P.version == 3 ? nvox[rk] : cc + g) : 0u;  // an unoccupied bin_curr: r_pod2pc skips the bin (erasor.cpp:313)
    };
    if constexpr (FOLD) {
        n_rev_all = st->n_rev;
        if (n_rev_all < 64u) {  // the usual handful: one wavefront, three DPP scans, one barrier
            if (threadIdx.x < 64u) {
                const uint32_t rk = threadIdx.x;
                uint32_t sz = 0, g = 0, rj = 0;
                if (rk < n_rev_all) rev_sizes(rk, sz, g, rj);
                const uint32_t i0 = esort::wave_incl_scan(sz), i1 = esort::wave_incl_scan(g), i2 = esort::wave_incl_scan(rj);
                if (rk <= n_rev_all) {
                    s_cv[rk] = i0 - sz;
                    s_cg[rk] = i1 - g;
                    s_cr[rk] = i2 - rj;
                }
                if (rk == 63u) {
                    s_carry[0] = i0;
                    s_carry[1] = i1;
                    s_carry[2] = i2;
                }
            }
            __syncthreads();
        } else {
            if (threadIdx.x < 3) s_carry[threadIdx.x] = 0;
            __syncthreads();
            for (uint32_t base = 0; base < n_rev_all; base += blockDim.x) {
                const uint32_t rk = base + threadIdx.x;
                uint32_t sz = 0, g = 0, rj = 0;
                if (rk < n_rev_all) rev_sizes(rk, sz, g, rj);
                uint32_t t0, t1, t2;
                const uint32_t p0 = block_excl_scan(sz, s_sm, t0);
                const uint32_t p1 = block_excl_scan(g, s_sm, t1);
                const uint32_t p2 = block_excl_scan(rj, s_sm, t2);
                const uint32_t c0 = s_carry[0], c1 = s_carry[1], c2 = s_carry[2];
                if (rk < n_rev_all && rk < ASM_RVMAX) {
                    s_cv[rk] = c0 + p0;
                    s_cg[rk] = c1 + p1;
                    s_cr[rk] = c2 + p2;
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    s_carry[0] = c0 + t0;
                    s_carry[1] = c1 + t1;
                    s_carry[2] = c2 + t2;
                }
                __syncthreads();
            }
            if (threadIdx.x == 0 && n_rev_all <= ASM_RVMAX) {
                s_cv[n_rev_all] = s_carry[0];
                s_cg[n_rev_all] = s_carry[1];
                s_cr[n_rev_all] = s_carry[2];
            }
            __syncthreads();
        }
        total_bins = st->total_bins0 + s_carry[0];
        n_static_est = total_bins + s_carry[1];
    } else if constexpr (RES) {
        total_bins = st->total_binsR;
        n_static_est = total_bins + st->ground_res;
    } else {
        total_bins = st->total_bins;
        n_static_est = st->n_static_est;
    }
    // exclusive prefix (over the reverted bins before rk) of: 0 output sizes, 1 ground, 2 rejected
    auto cum = [&](uint32_t rk, int which) -> uint32_t {
        if (n_rev_all <= ASM_RVMAX) return which == 0 ? s_cv[rk] : (which == 1 ? s_cg[rk] : s_cr[rk]);
        uint32_t a = 0;  // (more reverted bins than the tables hold: walk them)
        for (uint32_t r = 0; r < rk; ++r) {
            uint32_t sz, g, rj;
            rev_sizes(r, sz, g, rj);
            a += which == 0 ? sz : (which == 1 ? g : rj);
        }
        return a;
    };
    auto OUT_OFF = [&](uint32_t key) -> uint32_t {
        if constexpr (FOLD) return out_off0[key] + cum(rev_before[key], 0);
        else if constexpr (RES) return out_off0[key];
        else return out_off[key];
    };
    auto GROUND_OFF = [&](uint32_t rk) -> uint32_t {
        if constexpr (FOLD) return cum(rk, 1);
        else return ground_off[rk];
    };
    auto REJ_OFF = [&](uint32_t rk) -> uint32_t {
        if constexpr (FOLD) return cum(rk, 2);
        else return rej_off[rk];
    };
    if constexpr (FOLD) {
        if (blockIdx.x == 0) {  // the final tables and totals, for the getters and k_step_end
            for (uint32_t key = threadIdx.x; key < (uint32_t)P.B; key += blockDim.x) out_off[key] = OUT_OFF(key);
            for (uint32_t rk = threadIdx.x; rk < n_rev_all; rk += blockDim.x) {
                ground_off[rk] = GROUND_OFF(rk);
                rej_off[rk] = REJ_OFF(rk);
            }
            if (threadIdx.x == 0) {
                const uint32_t ncompl = moff[P.B + 1] - moff[P.B];
                st->total_bins = total_bins;
                st->n_ground = s_carry[1];
                st->n_rejected = s_carry[2];
                st->n_static_est = n_static_est;
                st->n_compl = ncompl;
                st->nF_new = n_static_est + ncompl;
                st->nF_valid_new = n_static_est + ncompl;
            }
        }
    }
    uint32_t nd = 0, nst = 0;  // label tallies of the copies this thread wrote to Fnew
    const uint32_t gmap = gridDim.x - tail;
    if (blockIdx.x >= gmap) {
        const uint32_t n_rev = st->n_rev;
        for (uint32_t rk = blockIdx.x - gmap; rk < n_rev; rk += tail) {
            const uint32_t key = rev_list[rk];
            if (qoff[key + 1] == qoff[key]) continue;  // selected = an unoccupied bin_curr: r_pod2pc skips it
            const uint32_t nv = nvox[rk], vo = vox_off[rk], oo = OUT_OFF(key);
            for (uint32_t v = threadIdx.x; v < nv; v += blockDim.x) {
                const float4 p = vox_out[vo + v];
                Fnew[oo + v] = XFORM ? xform(Tb2o, p) : p;
                if (is_dynamic_label(p.w)) ++nd; else ++nst;
            }
        }
        block_commit_labels(nd, nst, cnt);
        return;
    }
    const uint32_t n_act = st->voi_total;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_act; i += gmap * blockDim.x) {
        uint32_t nwr = 0;
        const uint32_t key = skeys[i];
        if (key > (uint32_t)P.B) continue;  // (round 5: a VoI-order slot kept for a late point that never came, bucket B + 1)
        const float4 p = spts[i];
        const float4 w = XFORM ? xform(Tb2o, p) : p;
        if (key == (uint32_t)P.B) {
            Fnew[n_static_est + (i - moff[P.B])] = w;
            nwr = 1;
        } else {
            const uint8_t act = action[key];
            const uint32_t r = i - moff[key];
            if (RES && act == 1) {
                // (k_assemble_late, once the bin's ground is known)
            } else if (act == 1) {
                const uint32_t rk = rev_idx[key];
                const uint32_t gr = grank[i];
                if (gflag[i]) {
                    Fnew[total_bins + GROUND_OFF(rk) + gr] = w;  // ground_viz copy
                    nwr = 1;
                    if (P.version == 2 && ccnt[key] > 0) {
                        Fnew[OUT_OFF(key) + ccnt[key] + gr] = w;  // v2: inside the bin too
                        nwr = 2;
                    }
                } else if (rejected) {
                    rejected[REJ_OFF(rk) + gr] = xform(Tb2o, p);  // map_rejected_ is handed out in the map frame (OMU.cpp:287)
                    rejected_src[REJ_OFF(rk) + gr] = ssrc[i];
                }
            } else if (act == 2) {
                Fnew[OUT_OFF(key) + ccnt[key] + r] = w;  // merged bin: curr points first
                nwr = 1;
            } else if (act == 3) {
                // map bin empty by construction
            } else {
                Fnew[OUT_OFF(key) + r] = w;
                nwr = 1;
            }
        }
        if (is_dynamic_label(p.w)) nd += nwr; else nst += nwr;
    }
    block_commit_labels(nd, nst, cnt);
}

// Round 6 (second half): the EARLY write-back of a step in the reserved layout WITHOUT a k_srt4 launch in front of it.  On config 2 the early
// stream's chain (k_srt4 15.5 us -> early write-back 9.5 -> next split -> chunk scan -> gather) is the longer branch of an overlapped step in
// three steps of four, and k_srt4 -- one workgroup, two rounds of global accesses, ten output arrays -- is a sixth of it.  All the write-back
// needs of it is, per bin, "reverted or not" and the bin's reserved offset (srt4_body's out_offR) and the two extents, and in v3 both follow
// from the first-pass status bytes and the two counts (erasor.cpp:510-511: the decision is local to a bin): EVERY workgroup builds that table
// itself in LDS -- one round trip for 3 x 2160 values, one pass of two block scans: the arithmetic of srt4_body's reserved branch, value for
// value -- and the launch's LAST workgroup is k_srt4 (srt4_body, unchanged) for everything behind it: the next split's late table, the late
// write-back, the getters.  One launch, one kernel boundary and ~13 us less on the early chain; same results.
struct EarlyArgs {
    // srt4_body's (the launch's last workgroup)
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
    uint32_t *out_offR, *gres_off;
    LateEnt *late;
    double leave_lim;
    // the write-back's
    const uint32_t *skeys;
    const float4 *spts;
    float4 *Fnew;
    unsigned long long *cnt;
};
__global__ __launch_bounds__(1024) void k_assemble_early(DP P, Xf Tb2o, EarlyArgs a) {
    __shared__ uint32_t sm[96];
    __shared__ uint32_t s_base[1024 * SRT_KPT];  // a bin's reserved offset minus its offset in VoI order: point i of the bin goes to s_base[key] + i
    __shared__ uint8_t s_st1[1024 * SRT_KPT];    // the k_srt4 workgroup: first-pass statuses; the others: 1 = reverted (k_assemble_late writes the bin)
    if (blockIdx.x == 0) {  // (the FIRST workgroup: dispatched with the first round, whatever the grid)
        if (g_stamps_on && threadIdx.x == 0) g_stamps[3] = wall_clock64();
        srt4_body(P, sm, s_st1, a.mcnt, a.mmin, a.mmax, a.ccnt, a.cmin, a.cmax, a.st1, a.status, a.action, a.rev_idx, a.rev_list, a.vox_off, a.st, a.out_off0,
                  a.rev_before, a.crej_off, a.st1_in, a.moff, a.qoff, a.out_offR, a.gres_off, a.late, a.leave_lim, a.moff);
        return;
    }
    if (g_stamps_on && blockIdx.x == 1 && threadIdx.x == 0) g_stamps[4] = wall_clock64();
    const int B = P.B;
    const int k0 = threadIdx.x * SRT_KPT;
    const uint32_t n_act = a.st->voi_total;  // (the step's own state, not k_srt4's)
    uint32_t szR[SRT_KPT], mo[SRT_KPT];
    bool rv[SRT_KPT];
    uint32_t sums[2] = {0u, 0u};  // reserved sizes, reserved ground: srt4_body's sums[4], sums[5]
    const double zb = fmax(fabs(P.min_h), fabs(P.max_h));
#pragma unroll
    for (int j = 0; j < SRT_KPT; ++j) {
        szR[j] = 0;
        mo[j] = 0;
        rv[j] = false;
        if (k0 + j < B) {
            const uint8_t v = a.st1_in[k0 + j];
            const uint32_t mc = a.mcnt[k0 + j], cc = a.ccnt[k0 + j];
            mo[j] = a.moff[k0 + j];
            rv[j] = (v & 0x7Fu) == ST_MAP && (v & 0x80u);  // srt_second, v3: MAP_IS_HIGHER and the map bin taller than 0.5 m
            if (rv[j]) {
                const int ring = (k0 + j) % P.R;
                const double rho = (ring + 1 >= P.R) ? P.max_r : (double)(ring + 1) * P.ring_size;
                const bool ml = !(sqrt(rho * rho + zb * zb) * (1.0 + 1e-5) < a.leave_lim);
                szR[j] = cc > 0 ? (mc + cc) * (ml ? 2u : 1u) : 0u;
                sums[1] += mc * (ml ? 2u : 1u);
            } else {
                szR[j] = mc;
            }
            sums[0] += szR[j];
        }
    }
    const uint32_t mo_compl = a.moff[B];
    uint32_t pre[2], tot[2];
    block_excl_scan_n<2>(sums, sm, pre, tot);
    {
        uint32_t p4 = pre[0];
#pragma unroll
        for (int j = 0; j < SRT_KPT; ++j)
            if (k0 + j < B) {
                s_base[k0 + j] = p4 - mo[j];
                s_st1[k0 + j] = rv[j] ? 1 : 0;
                p4 += szR[j];
            }
    }
    __syncthreads();
    const uint32_t base_compl = tot[0] + tot[1] - mo_compl;  // the complement follows the bins and the ground part (n_static_est)
    uint32_t nd = 0, nst = 0;
    const uint32_t gmap = gridDim.x - 1u;
    // (a workgroup of 1024 threads at 70 VGPRs has a compute unit to itself: the grid is one round of workgroups -- see the launch -- and
    // every thread keeps four points in flight)
    const uint32_t stride = gmap * blockDim.x;
    auto place = [&](uint32_t i, uint32_t key, const float4 &p) {
        uint32_t nwr = 0;
        if (key == (uint32_t)B) {
            a.Fnew[base_compl + i] = xform(Tb2o, p);
            nwr = 1;
        } else if (!s_st1[key]) {
            a.Fnew[s_base[key] + i] = xform(Tb2o, p);
            nwr = 1;
        }
        if (is_dynamic_label(p.w)) nd += nwr; else nst += nwr;
    };
    uint32_t i = (blockIdx.x - 1u) * blockDim.x + threadIdx.x;
    for (; i + 3u * stride < n_act; i += 4u * stride) {
        uint32_t key[4];
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = a.skeys[i + (uint32_t)u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = a.spts[i + (uint32_t)u * stride];  // (a slot of the dead bucket holds zeros: read, not used)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (key[u] <= (uint32_t)B) place(i + (uint32_t)u * stride, key[u], p[u]);  // (> B: a VoI-order slot kept for a late point that never came)
    }
    for (; i < n_act; i += stride) {
        const uint32_t key = a.skeys[i];
        if (key > (uint32_t)B) continue;
        place(i, key, a.spts[i]);
    }
    block_commit_labels(nd, nst, a.cnt);
}

// scan-side contributions: v3 voxelised reverted bins; v2 curr points of reverted / merged / curr-only bins
template <bool XFORM>
__global__ __launch_bounds__(256) void k_assemble_bins(DP P, Xf Tb2o, const uint8_t *__restrict__ action,
                                                        const uint32_t *__restrict__ rev_idx, const uint32_t *__restrict__ qoff,
                                                        const float4 *__restrict__ sq, const uint32_t *__restrict__ nvox,
                                                        const uint32_t *__restrict__ vox_off, const float4 *__restrict__ vox_out,
                                                        const uint32_t *__restrict__ out_off, const uint32_t *__restrict__ crej_off,
                                                        float4 *__restrict__ Fnew, float4 *__restrict__ curr_rejected, unsigned long long *cnt) {
    const int key = blockIdx.x;
    const uint8_t act = action[key];
    if (act == 0) return;  // (block-uniform exits: no thread reaches the tally's barrier)
    const uint32_t qo = qoff[key], cc = qoff[key + 1] - qo;
    uint32_t nd = 0, nst = 0;
    if (act == 1 && P.version == 3) {
        if (cc == 0) return;
        const uint32_t rk = rev_idx[key];
        const uint32_t nv = nvox[rk], vo = vox_off[rk], oo = out_off[key];
        for (uint32_t v = threadIdx.x; v < nv; v += blockDim.x) {
            const float4 p = vox_out[vo + v];
            Fnew[oo + v] = XFORM ? xform(Tb2o, p) : p;
            if (is_dynamic_label(p.w)) ++nd; else ++nst;
        }
    } else if (act == 4) {
        if (curr_rejected)
            for (uint32_t j = threadIdx.x; j < cc; j += blockDim.x) curr_rejected[crej_off[key] + j] = xform(Tb2o, sq[qo + j]);
    } else {  // v2: act 1 (with curr occupied), 2, 3: curr points lead the bin
        const uint32_t oo = out_off[key];
        for (uint32_t j = threadIdx.x; j < cc; j += blockDim.x) {
            const float4 p = sq[qo + j];
            Fnew[oo + j] = XFORM ? xform(Tb2o, p) : p;
            if (is_dynamic_label(p.w)) ++nd; else ++nst;
        }
    }
    block_commit_labels(nd, nst, cnt);
}

// Round 5: the LATE half of a write-back in the reserved layout -- what had to wait for the per-bin launch.  One workgroup per reverted bin
// (grid-stride over the list): the bin's voxels (erasor.cpp:521-528) into its reserved range, its ground into its range of the ground_viz
// part (erasor.cpp:527, 616), holes (x = HOLE_BITS) into whatever of both stays empty, its rejected points into map_rejected
// (erasor.cpp:528) with their pre-step map indices; workgroup 0 leaves the DENSE tables and totals the getters and the step's end read
// (what k_layout4 / k_assemble_map<., true> leave), the late table's `actual` column and the holes before every entry.
// src_late / src_holes / n_src_late: the late table of the step that WROTE the region this step read (its source indices count reserved
// slots: idx - holes before idx = the logical index), nullptr / 0: that region was dense.
template <class TP, class HP>
__device__ __forceinline__ uint32_t late_logical(uint32_t src, TP tab, HP holes, uint32_t n) {
    if (!n) return src;
    uint32_t lo = 0, hi = n;  // first entry whose range ends beyond src
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tab[mid].start + tab[mid].ntotal <= src) lo = mid + 1;
        else hi = mid;
    }
    return src - holes[lo];  // (src lies before or inside entry lo: inside, its data slots are dense from the range's start)
}
__global__ __launch_bounds__(256) void k_assemble_late(DP P, Xf Tb2o, const uint32_t *__restrict__ rev_list, const float4 *__restrict__ spts,
                                                        const uint32_t *__restrict__ ssrc, const uint32_t *__restrict__ moff,
                                                        const uint32_t *__restrict__ qoff, const uint8_t *__restrict__ gflag,
                                                        const uint32_t *__restrict__ grank, const uint32_t *__restrict__ ng_arr,
                                                        const uint32_t *__restrict__ nvox, const uint32_t *__restrict__ vox_off,
                                                        const float4 *__restrict__ vox_out, const uint32_t *__restrict__ out_offR,
                                                        const uint32_t *__restrict__ gres_off, const uint32_t *__restrict__ out_off0,
                                                        const uint32_t *__restrict__ rev_before, LateEnt *__restrict__ late,
                                                        uint32_t *__restrict__ late_holes, uint32_t *__restrict__ out_off,
                                                        uint32_t *__restrict__ ground_off, uint32_t *__restrict__ rej_off, DevState *st,
                                                        float4 *__restrict__ Fnew, float4 *__restrict__ rejected,
                                                        uint32_t *__restrict__ rejected_src, unsigned long long *cnt,
                                                        const LateEnt *__restrict__ src_late, const uint32_t *__restrict__ src_holes,
                                                        uint32_t n_src_late) {
    __shared__ uint32_t s_sm[40];
    __shared__ uint32_t s_carry[4];
    __shared__ uint32_t s_cvl[ASM_RVMAX + 1];
    // (round 6, second half: the late table of the region that was read -- a few dozen entries -- is staged in LDS while the bin's own loads are
    // on their way: every rejected point's source index is converted by a binary search in it, five or six DEPENDENT reads, which from
    // global memory were ~4 us of this launch's chain)
    constexpr uint32_t SRC_LATE_LDS = 512;
    __shared__ uint32_t s_src_end[SRC_LATE_LDS], s_src_holes[SRC_LATE_LDS + 1];
    const bool src_in_lds = n_src_late != 0 && n_src_late <= SRC_LATE_LDS;
    if (src_in_lds) {
        for (uint32_t i = threadIdx.x; i <= n_src_late; i += blockDim.x) {
            if (i < n_src_late) {
                const LateEnt e = src_late[i];
                s_src_end[i] = e.start + e.ntotal;
            }
            s_src_holes[i] = src_holes[i];
        }
    }
    const uint32_t n_rev = st->n_rev;
    CHAIN_STAMP(8);
    const uint32_t total_binsR = st->total_binsR;
    const float4 hole = make_float4(__uint_as_float(HOLE_BITS), 0.f, 0.f, 0.f);
    auto sizes = [&](uint32_t rk, uint32_t &sz, uint32_t &g, uint32_t &rj) {
        const uint32_t key = rev_list[rk];
        const uint32_t cc = qoff[key + 1] - qoff[key], mc = moff[key + 1] - moff[key];
        g = ng_arr[rk];
        rj = mc - g;
        sz = cc > 0 ? nvox[rk] : 0u;  // an unoccupied bin_curr: r_pod2pc skips the bin (erasor.cpp:313)
    };
    // (rejected points before reverted bin rk -- the dense map_rejected -- are summed by the workgroup that needs them: below)
    uint32_t nd = 0, nst = 0;
    // (the launch's LAST workgroup leaves the tables and totals, the others take the bins: side by side)
    const uint32_t nbin_wg = gridDim.x - 1u;
    for (uint32_t rk = blockIdx.x; rk < n_rev && blockIdx.x < nbin_wg; rk += nbin_wg) {
        // (round 6: everything that depends on rk alone -- the bin's key, its two table entries, its place in the voxel scratch, its counts --
        // is fetched in ONE round trip, ahead of the prefix of rejected points: they used to follow it one behind the other)
        const uint32_t key = rev_list[rk];
        const LateEnt eb = late[rk], eg = late[n_rev + rk];
        const uint32_t vo = vox_off[rk];
        const uint32_t ng = ng_arr[rk], nvx = nvox[rk];
        // (round 6, second half: the earlier bins' keys and ground counts -- the prefix of rejected points -- travel in the SAME two round
        // trips as the bin's own key and range: thread r takes bin r; behind them they were two more)
        const uint32_t r_first = threadIdx.x;
        uint32_t key_r = 0, ng_r = 0;
        if (r_first < rk) {
            key_r = rev_list[r_first];
            ng_r = ng_arr[r_first];
        }
        const uint32_t o0 = moff[key], mc = moff[key + 1] - o0;
        const uint32_t nv = (qoff[key + 1] - qoff[key]) > 0 ? nvx : 0u;  // an unoccupied bin_curr: r_pod2pc skips the bin (erasor.cpp:313)
        uint32_t rej0;
        {
            uint32_t a = 0;
            if (r_first < rk) a = (moff[key_r + 1] - moff[key_r]) - ng_r;
            for (uint32_t r = threadIdx.x + blockDim.x; r < rk; r += blockDim.x) {
                uint32_t sz, g, rj;
                sizes(r, sz, g, rj);
                a += rj;
            }
            (void)block_excl_scan(a, s_sm, rej0);
        }
        for (uint32_t v = threadIdx.x; v < eb.ntotal; v += blockDim.x) {
            float4 w = hole;
            if (v < nv) {
                const float4 p = vox_out[vo + v];
                w = xform(Tb2o, p);
                if (is_dynamic_label(p.w)) ++nd; else ++nst;
            }
            Fnew[eb.start + v] = w;
        }
        for (uint32_t j = threadIdx.x; j < mc; j += blockDim.x) {
            const uint32_t i = o0 + j;
            const float4 p = spts[i];
            const uint32_t gr = grank[i];
            if (gflag[i]) {
                Fnew[eg.start + gr] = xform(Tb2o, p);  // ground_viz copy (the bin itself holds the voxelised curr + ground)
                if (is_dynamic_label(p.w)) ++nd; else ++nst;
            } else {
                rejected[rej0 + gr] = xform(Tb2o, p);  // map_rejected_ is handed out in the map frame (OMU.cpp:287)
                const uint32_t sidx = ssrc[i];
                uint32_t lg;
                if (src_in_lds) {  // (late_logical's search, the table in LDS)
                    uint32_t lo = 0, hi = n_src_late;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_src_end[mid] <= sidx) lo = mid + 1;
                        else hi = mid;
                    }
                    lg = sidx - s_src_holes[lo];
                } else {
                    lg = late_logical(sidx, src_late, src_holes, n_src_late);
                }
                rejected_src[rej0 + gr] = lg;
            }
        }
        for (uint32_t v = ng + threadIdx.x; v < eg.ntotal; v += blockDim.x) Fnew[eg.start + v] = hole;
        if (threadIdx.x == 0) {
            late[rk].actual = nv;
            late[n_rev + rk].actual = ng;
        }
        __syncthreads();
    }
    (void)gres_off;
    (void)total_binsR;
    if (blockIdx.x == nbin_wg) {
        // dense tables and totals (k_layout4's): prefixes over the reverted bins of their output sizes, ground, rejected points
        if (threadIdx.x < 4) s_carry[threadIdx.x] = 0;
        __syncthreads();
        // (round 6, second half: the five prefixes in ONE pass of scans -- they were five pairs of barriers behind four dependent round
        // trips; the ground entries' holes follow the bins' in the table, so their prefix is lifted by the bins' total once that is known)
        if (threadIdx.x == 0) s_carry[3] = 0;
        __shared__ uint32_t s_carry4;
        __shared__ uint32_t s_smn[80];
        if (threadIdx.x == 0) s_carry4 = 0;
        __syncthreads();
        for (uint32_t base = 0; base < n_rev; base += blockDim.x) {
            const uint32_t rk = base + threadIdx.x;
            uint32_t v[5] = {0u, 0u, 0u, 0u, 0u};  // output size, ground, rejected, holes of the bin's range, holes of its ground range
            if (rk < n_rev) {
                const uint32_t nt_b = late[rk].ntotal, nt_g = late[n_rev + rk].ntotal;
                sizes(rk, v[0], v[1], v[2]);
                v[3] = nt_b - v[0];
                v[4] = nt_g - v[1];
            }
            uint32_t pre[5], tot[5];
            block_excl_scan_n<5>(v, s_smn, pre, tot);
            const uint32_t c0 = s_carry[0], c1 = s_carry[1], c2 = s_carry[2], c3 = s_carry[3], c4 = s_carry4;
            if (rk < n_rev) {
                // (late_holes[i]: holes in the ranges before entry i; the bins' entries first, the ground entries follow below)
                late_holes[rk] = c3 + pre[3];
                late_holes[n_rev + rk] = c4 + pre[4];  // (+ the bins' holes in all: below)
                ground_off[rk] = c1 + pre[1];
                rej_off[rk] = c2 + pre[2];
                if (rk < ASM_RVMAX) s_cvl[rk] = c0 + pre[0];  // voxels of the reverted bins before rk (out_off by key, below)
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                s_carry[0] = c0 + tot[0];
                s_carry[1] = c1 + tot[1];
                s_carry[2] = c2 + tot[2];
                s_carry[3] = c3 + tot[3];
                s_carry4 = c4 + tot[4];
            }
            __syncthreads();
        }
        const uint32_t sum_sz = s_carry[0], sum_g = s_carry[1], sum_rj = s_carry[2], holes_bins = s_carry[3];
        if (threadIdx.x == 0 && n_rev < ASM_RVMAX + 1) s_cvl[n_rev] = sum_sz;
        for (uint32_t base = 0; base < n_rev; base += blockDim.x) {  // (every thread lifts what it wrote itself)
            const uint32_t rk = base + threadIdx.x;
            if (rk < n_rev) late_holes[n_rev + rk] += holes_bins;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry[3] = holes_bins + s_carry4;
        __syncthreads();
        if (threadIdx.x == 0) late_holes[2 * n_rev] = s_carry[3];  // (all of them: what a source index behind the last range is reduced by)
        // dense out_off by key: out_off0[key] + the voxels of the reverted bins before it
#pragma unroll 4
        for (uint32_t key = threadIdx.x; key < (uint32_t)P.B; key += blockDim.x) {
            const uint32_t rb = rev_before[key];
            uint32_t a = 0;
            if (rb <= ASM_RVMAX && n_rev <= ASM_RVMAX) a = s_cvl[rb];
            else
                for (uint32_t r = 0; r < rb; ++r) {  // (more reverted bins than the table holds: walk them)
                    uint32_t sz, g, rj;
                    sizes(r, sz, g, rj);
                    a += sz;
                }
            out_off[key] = out_off0[key] + a;
        }
        if (threadIdx.x == 0) {
            const uint32_t total_bins = st->total_bins0 + sum_sz;
            st->total_bins = total_bins;
            st->n_ground = sum_g;
            st->n_rejected = sum_rj;
            st->n_static_est = total_bins + sum_g;
            st->nF_valid_new = total_bins + sum_g + st->n_compl;
        }
    }
    block_commit_labels(nd, nst, cnt);
}

// label counters over a float4 cloud (parse_dynamic_obj as counters, utils.cpp:57-78)
__global__ __launch_bounds__(256) void k_count_labels4(const float4 *__restrict__ pts, uint32_t n_host, const uint32_t *n_dev,
                                                        unsigned long long *n_static, unsigned long long *n_dynamic) {
    __shared__ uint32_t sd[4], ss[4];
    const uint32_t n = n_dev ? *n_dev : n_host;
    uint32_t d = 0, s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (is_dynamic_label(pts[i].w)) ++d; else ++s;
    }
    for (int o = 32; o > 0; o >>= 1) {
        d += __shfl_down(d, o, 64);
        s += __shfl_down(s, o, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        sd[threadIdx.x >> 6] = d;
        ss[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one device-scope atomic per block and counter
        const uint32_t td = sd[0] + sd[1] + sd[2] + sd[3], ts = ss[0] + ss[1] + ss[2] + ss[3];
        if (td) atomicAdd(n_dynamic, (unsigned long long)td);
        if (ts) atomicAdd(n_static, (unsigned long long)ts);
    }
}

// empty kernel: bracketed by HIP events exactly like k_voi_split to measure the bracket's own overhead
__global__ void k_null() {}
// ALU-only delay (measurement aid: ERASOR_HIP_QPAD_US / ERASOR_HIP_MPAD_US lengthen a chain by a known amount to see which one bounds the step)
__global__ void k_pad(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

// start of a scan's query chain (its own stream): counters, bounding box, bucket totals, voxel count of this query side
__device__ __forceinline__ void query_begin_body(Counters *qctr, uint32_t *bb, uint32_t *qb_tot, uint32_t qb_n, uint32_t *nvox, uint32_t nvox_init) {
    for (uint32_t b = threadIdx.x; b < qb_n; b += blockDim.x) qb_tot[b] = 0;  // bucket totals of the query counting sort
    if (threadIdx.x == 0) {
        qctr->n_neg_sector = qctr->n_ambiguous = qctr->n_degenerate = qctr->n_voxel_overflow = qctr->n_sort_fallback = 0;
        qctr->sort_qoverflow = qctr->err = 0;
        *nvox = nvox_init;  // (pre-voxelised input: the count is known; otherwise k_run_begin writes it)
    }
    if (threadIdx.x < 3) bb[threadIdx.x] = 0xFFFFFFFFu;
    if (threadIdx.x >= 3 && threadIdx.x < 6) bb[threadIdx.x] = 0u;
}
__global__ void k_query_begin(Counters *qctr, uint32_t *bb, uint32_t *qb_tot, uint32_t qb_n, uint32_t *nvox, uint32_t nvox_init) { query_begin_body(qctr, bb, qb_tot, qb_n, nvox, nvox_init); }
// ---- mapgen (src/mapgen/mapgen.hpp:198-257): per-scan preparation ----------------------------------------------
// keep flag of the self-filter (mapgen.hpp:219-228): a point is dropped when pow(x,2) + pow(y,2) (double) is below
// max_dist_square, a FLOAT holding pow(CAR_BODY_SIZE, 2)
__global__ __launch_bounds__(256) void k_mapgen_flag(const float4 *__restrict__ pts, uint32_t n, float max_dist_square,
                                                      uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const double dx = (double)p.x, dy = (double)p.y;
    const double dist_square = dx * dx + dy * dy;  // squares of floats are exact in double, like pow(x, 2)
    flag[i] = (dist_square < (double)max_dist_square) ? 0u : 1u;
}
// stable compaction of the kept points, then lidar->origin-of-body and the pose (two pcl::transformPointCloud, :231-237)
__global__ __launch_bounds__(256) void k_mapgen_scatter(const float4 *__restrict__ pts, uint32_t n, const uint32_t *__restrict__ flag,
                                                         const uint32_t *__restrict__ pl, const uint32_t *__restrict__ tops, Xf T1, Xf T2,
                                                         float4 *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    out[pl[i] + tops[i >> 10]] = xform(T2, xform(T1, pts[i]));
}

// ---- map store maintenance ---------------------------------------------------------------------
// split an AoS float4 cloud into the outskirts layout at O[dst0 ...]
__global__ __launch_bounds__(256) void k_store_outskirts(const float4 *__restrict__ src, uint32_t n, float2 *__restrict__ Oxy,
                                                          float2 *__restrict__ Ozi, uint32_t dst0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i];
    Oxy[dst0 + i] = make_float2(p.x, p.y);
    Ozi[dst0 + i] = make_float2(p.z, p.w);
}
// valid flags of the outskirts region (for compaction / read-back)
__global__ __launch_bounds__(256) void k_o_valid(const float2 *__restrict__ Oxy, uint32_t o_begin, uint32_t o_cap,
                                                  uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= o_cap - o_begin) return;
    flag[i] = (__float_as_uint(Oxy[o_begin + i].x) != HOLE_BITS) ? 1u : 0u;
}
// stable compaction of the valid outskirts entries into an AoS float4 buffer
__global__ __launch_bounds__(256) void k_o_compact(const float2 *__restrict__ Oxy, const float2 *__restrict__ Ozi, uint32_t o_begin,
                                                    uint32_t o_cap, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pl,
                                                    const uint32_t *__restrict__ tops, float4 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= o_cap - o_begin) return;
    if (!flag[i]) return;
    const float2 a = Oxy[o_begin + i], b = Ozi[o_begin + i];
    dst[pl[i] + tops[i >> 10]] = make_float4(a.x, a.y, b.x, b.y);
}

// round 5: the same for the VoI-resident region when it holds reserved slots nothing filled (x == HOLE_BITS)
__global__ __launch_bounds__(256) void k_f_valid(const float4 *__restrict__ F, uint32_t n, uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (__float_as_uint(F[i].x) != HOLE_BITS) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_voi_live(const uint32_t *__restrict__ voi_key, uint32_t n, uint32_t B, uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = voi_key[i] <= B ? 1u : 0u;  // (B + 1: a place kept for a late point that did not come, see k_late_gather)
}
__global__ __launch_bounds__(256) void k_f_compact(const float4 *__restrict__ F, uint32_t n, const uint32_t *__restrict__ flag,
                                                    const uint32_t *__restrict__ pl, const uint32_t *__restrict__ tops, float4 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) dst[pl[i] + tops[i >> 10]] = F[i];
}

// ---- large-scale mode: set_submap (OMU.cpp:360-379) ------------------------------------------------
// flag[i] = 1 iff |x - pt.x| < submap_size && |y - pt.y| < submap_size (doubles, strict)
__global__ __launch_bounds__(256) void k_box_flag(const float4 *__restrict__ g, uint32_t n, double x, double y, double S,
                                                   uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = g[i];
    const double dx = fabs(x - (double)p.x), dy = fabs(y - (double)p.y);
    flag[i] = ((dx < S) && (dy < S)) ? 1u : 0u;
}
// stable partition: submap points -> outskirts layout at O[dst0 + rank], the rest -> C[i - rank]
__global__ __launch_bounds__(256) void k_box_partition(const float4 *__restrict__ g, uint32_t n, const uint32_t *__restrict__ flag,
                                                        const uint32_t *__restrict__ pl, const uint32_t *__restrict__ tops,
                                                        float2 *__restrict__ Oxy, float2 *__restrict__ Ozi, uint32_t dst0,
                                                        float4 *__restrict__ C) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = g[i];
    const uint32_t r = pl[i] + tops[i >> 10];
    if (flag[i]) {
        Oxy[dst0 + r] = make_float2(p.x, p.y);
        Ozi[dst0 + r] = make_float2(p.z, p.w);
    } else {
        C[i - r] = p;
    }
}
// label counters over the {z,intensity} half of the outskirts layout
__global__ __launch_bounds__(256) void k_count_labels_zi(const float2 *__restrict__ zi, uint32_t n, unsigned long long *n_static,
                                                          unsigned long long *n_dynamic) {
    __shared__ uint32_t sd[4], ss[4];
    uint32_t d = 0, s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (is_dynamic_label(zi[i].y)) ++d; else ++s;
    }
    for (int o = 32; o > 0; o >>= 1) {
        d += __shfl_down(d, o, 64);
        s += __shfl_down(s, o, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        sd[threadIdx.x >> 6] = d;
        ss[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t td = sd[0] + sd[1] + sd[2] + sd[3], ts = ss[0] + ss[1] + ss[2] + ss[3];
        if (td) atomicAdd(n_dynamic, (unsigned long long)td);
        if (ts) atomicAdd(n_static, (unsigned long long)ts);
    }
}

// device libm probe: sqrt / div / atan2 in double, as the binning uses them (tests pin these vs host libm)
// test hook: the bin key both ways (float32 decision with float64 fallback | float64 only) + how many points the fast path decided
__global__ void k_probe_bin_keys(DP P, const float4 *pts, uint32_t n, uint32_t *k_fast, uint32_t *k_exact, Counters *ca, Counters *cb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = pts[i];
    k_fast[i] = bin_key(P, q.x, q.y, q.z, ca);
    k_exact[i] = bin_key_exact(P, q.x, q.y, q.z, cb);
}
__global__ void k_probe_math(const double *x, const double *y, uint32_t n, double *o_sqrt, double *o_div, double *o_atan2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_sqrt[i] = sqrt(x[i] * x[i] + y[i] * y[i]);
    o_div[i] = x[i] / y[i];
    o_atan2[i] = atan2(y[i], x[i]);
}

// ================================================================================================
// round 6: the query chains of SEVERAL announced scans as ONE set of launches (OMU.cpp:237-241 per node; the offline driver knows
// its nodes ahead, main_in_your_env.cpp:92-123).  Every launch of a chain is bound by latency, not by work (a sort level moves 1 MB
// in 6-10 us on a chip that is 95 % idle), so a launch that serves the same stage of two to four scans costs what it costs for one:
// the per-scan queue time of a chain halves to quarters.  Nothing is concatenated: blockIdx.y selects the scan, every scan keeps its
// own query side (buffers, counters, queues) and the bodies are the single-scan kernels' bodies, unchanged.
// ================================================================================================
static constexpr int QBATCH_MAX = 4;
struct QSideDev {  // one scan's query side as the batched kernels see it (device pointers of erasor_hip.hip's QSide)
    const float4 *scan_in;
    float4 *cent, *query, *sq;
    uint32_t *bb, *qk_a, *qk_b, *qv_a, *qv_b, *qposL, *qposR, *qtops, *run_begin, *ukeys, *qkey, *qhead, *wtileL, *wtileR, *hkey, *hval;
    uint32_t *qb_tot, *qb_hist, *qoff, *ccnt, *d_nvox;
    float *cmin, *cmax;
    esort::Seg *esq0, *esq1, *essmall;
    EsQueues *esqs;
    WideSeg *wseg0, *wseg1;
    WideState *wstate;
    VoxGrid *qgrid;
    Counters *d_qctr;
    Xf Tl;
    uint32_t n;
    int32_t hbits;
};
struct QBatch {
    QSideDev s[QBATCH_MAX];
};
#define QB_SIDE const QSideDev &q = b.s[blockIdx.y]
__global__ __launch_bounds__(256) void k_query_begin_b(QBatch b, uint32_t qb_n) {
    QB_SIDE;
    query_begin_body(q.d_qctr, q.bb, q.qb_tot, qb_n, q.d_nvox, 0u);
}
__global__ __launch_bounds__(256) void k_bbox_b(QBatch b) {
    QB_SIDE;
    bbox_body(q.scan_in, q.n, q.bb);
}
__global__ __launch_bounds__(256) void k_voxel_keys_es_b(QBatch b, float leaf) {
    QB_SIDE;
    EsInit es;
    es.q0 = q.esq0;
    es.smallq = q.essmall;
    es.qs = q.esqs;
    es.w0 = q.wseg0;
    es.ws = q.wstate;
    voxel_keys_es_body(q.scan_in, q.n, q.bb, leaf, q.qk_a, q.qv_a, q.qgrid, q.d_qctr, q.hkey, 1u << q.hbits, es);
}
__global__ __launch_bounds__(256) void k_esort_wide_mark_b(QBatch b, int cur) {
    QB_SIDE;
    esort_wide_mark_body(q.qk_a, q.qv_a, q.qposL, q.qposR, cur ? q.wseg1 : q.wseg0, q.wstate, cur, q.wtileL, q.wtileR);
}
__global__ __launch_bounds__(256) void k_esort_wide_swap_b(QBatch b, int cur, int last_level) {
    QB_SIDE;
    esort_wide_swap_body(q.qk_a, q.qv_a, q.qposL, q.qposR, cur ? q.wseg1 : q.wseg0, cur ? q.wseg0 : q.wseg1, q.wstate, cur, q.wtileL, q.wtileR, q.esq0,
                         q.essmall, q.esqs, 65536u, last_level, q.d_qctr);
}
__global__ __launch_bounds__(1024) void k_esort_mid_b(QBatch b) {
    QB_SIDE;
    esort_mid_body(q.qk_a, q.qv_a, q.qposL, q.qposR, q.esq0, q.essmall, q.esqs, 0, 65536u, q.d_qctr);
}
__global__ __launch_bounds__(1024) void k_esort_final_b(QBatch b) {  // (behind k_esort_mid queue 0 is spent: the empty queue 1 stands for "no big segments")
    QB_SIDE;
    esort_final_body(q.qk_a, q.qv_a, q.qposL, q.qposR, q.qhead, q.qk_b, q.qv_b, q.essmall, q.esq1, q.esqs, 1, q.d_qctr, (unsigned long long *)nullptr);
}
__global__ __launch_bounds__(256) void k_run_count_b(QBatch b) {
    QB_SIDE;
    if (blockIdx.x * 1024u >= q.n && blockIdx.x) return;
    run_count_body(q.qk_b, q.n, q.qtops);
}
__global__ __launch_bounds__(256) void k_run_emit_b(QBatch b) {
    QB_SIDE;
    const uint32_t ntile = q.n ? (q.n + 1023u) / 1024u : 1u;  // (k_run_count's grid for THIS scan; the launch is sized for the longest)
    if (blockIdx.x >= ntile) return;
    run_emit_body(q.qk_b, q.n, q.qtops, ntile, q.run_begin, q.d_nvox);
}
__global__ __launch_bounds__(256) void k_centroids_b(QBatch b) {
    QB_SIDE;
    centroids_body(q.scan_in, q.qk_b, q.qv_b, q.run_begin, q.d_nvox, q.cent, q.ukeys, q.hkey, q.hval, q.hbits);
}
__global__ __launch_bounds__(256) void k_query_nn_b(QBatch b, DP P) {
    QB_SIDE;
    query_nn_body(q.scan_in, q.qv_b, q.run_begin, q.ukeys, q.cent, q.d_nvox, q.qgrid, q.Tl, P, q.d_qctr, q.query, q.qkey, q.hkey, q.hval, q.hbits);
}
__global__ __launch_bounds__(1024) void k_qb_hist_b(QBatch b, uint32_t nb) {
    QB_SIDE;
    qb_hist_body(q.qkey, q.n, q.d_nvox, nb, q.qb_hist, q.qb_tot);
}
__global__ __launch_bounds__(1024) void k_qb_scan_b(QBatch b, uint32_t nb) {
    QB_SIDE;
    qb_scan_body(q.qb_tot, nb, q.qoff);
}
__global__ __launch_bounds__(1024) void k_qb_scatter_b(QBatch b, uint32_t nb, int bits) {
    QB_SIDE;
    qb_scatter_body(q.qkey, q.query, q.n, q.d_nvox, nb, bits, q.qb_hist, q.qoff, q.sq);
}
__global__ __launch_bounds__(256) void k_bin_stats_b(QBatch b, uint32_t B) {
    QB_SIDE;
    bin_stats_body(q.sq, q.qoff, B, q.ccnt, q.cmin, q.cmax);
}
#undef QB_SIDE

}  // namespace ek
#endif
