// erasor_hip.hip — host side of liberasor_hip.so: handle, HBM map store, step orchestration.
// C ABI declared in include/erasor_hip.h.  All arithmetic happens in kernels.hip.h; this file only
// sizes buffers, launches kernels on the handle's stream and copies small result structs back.
//
// There is NO CPU fallback: without a usable HIP device every entry point that needs the GPU
// returns ERASOR_E_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/erasor_hip.h"
#ifndef ERASOR_REV_GRID
#define ERASOR_REV_GRID 128u  // workgroups of the per-bin launch (one reverted bin each; more bins than workgroups: a grid-stride loop)
#endif
#ifndef ERASOR_EARLY_GMAP
#define ERASOR_EARLY_GMAP 200u  // workgroups of k_assemble_early that write the map back (build options: A/B with tools/ab_dirs.sh, EXPERIMENTS r06-8)
#endif
#include "kernels.hip.h"

using namespace ek;

namespace {

struct ProfEntry {
    double ms = 0;
    uint64_t launches = 0;
};

struct PendingEvt {
    int name_id;
    hipEvent_t a, b;
};

template <class T>
struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
};

}  // namespace

// The query side of a step (see erasor_hip_handle::q)
// layout of a HOST scan's records: float x, y, z at byte 0, float intensity at byte `ioff`, one record every `stride` bytes
// (the library's own XYZI rows: 16 / 12; pcl::PointXYZI, what OMU.cpp:237 pcl::fromROSMsg leaves: 32 / 16)
struct RowFmt {
    uint32_t stride = 16, ioff = 12;
};
static constexpr int NSIDE = 8;  // the scan being stepped + up to seven announced ahead (round 6: chains of announced scans share launches, see flush_held)
static constexpr int MAX_AHEAD = NSIDE - 1;
struct QSide {
    uint32_t capS = 0;
    DBuf<float4> scan, cent, query, sq;
    const float4 *scan_in = nullptr;  // scan being voxelised (the caller's device buffer, or scan)
    DBuf<uint32_t> bb, qk_a, qk_b, qv_a, qv_b, qposL, qposR, qflag, qpl, qtops, run_begin, ukeys, qkey;
    DBuf<uint32_t> qhead;
    DBuf<esort::Seg> esq0, esq1, esq2, essmall;
    DBuf<EsQueues> esqs;
    DBuf<WideSeg> wseg0, wseg1;
    DBuf<WideState> wstate;
    DBuf<uint32_t> wtileL, wtileR;
    DBuf<VoxGrid> qgrid;
    DBuf<uint32_t> hkey, hval;        // voxel key -> voxel id hash of the scan voxelisation (k_centroids fills, k_query_nn probes)
    int hbits = 10;
    DBuf<uint32_t> qb_tot;            // [B + 1] bucket totals of the query counting sort
    DBuf<uint32_t> qb_hist;           // [tiles][B + 1] histogram of the same
    DBuf<uint32_t> qoff, ccnt;        // bins of the query: offsets, counts, min / max z
    DBuf<float> cmin, cmax;
    DBuf<uint32_t> d_nvox;            // voxels of this scan (device word every downstream launch reads)
    DBuf<Counters> d_qctr;            // counters / error flags of this scan's query chain (folded into the step's by k_step_end)
    float4 *stage = nullptr;          // pinned host staging of a HOST scan (hipHostMalloc): memcpy in, hipMemcpyAsync out -- a pageable
    size_t stage_cap = 0;             // hipMemcpy of 2 MB blocked the caller for milliseconds while chains were in flight
    hipEvent_t ev_h2d = nullptr;      // the staged scan has reached q.scan (recorded on the copy stream)
    bool h2d_pending = false;         // the chain that reads q.scan has to wait for ev_h2d
    hipEvent_t ev_keys = nullptr;     // voxel keys (and the VoxelGrid overflow flag) are final
    hipEvent_t ev_done = nullptr;     // the whole query chain of the scan is done
    bool held = false;                // the chain is set up but its launches wait for a set shared with the next announcements (flush_held)
    // bookkeeping of a chain that has been enqueued (erasor_hip_prefetch_scan) but not yet consumed by a step
    const void *src = nullptr;
    size_t src_n = 0;
    float Tl[16] = {0};
    uint32_t ns = 0;
    bool src_dev = false;             // src is a device pointer (read in place; not fingerprinted)
    uint64_t fp = 0;                  // host scans are copied when they are announced: hash of EVERY record of that copy ...
    bool fp_valid = false;            // ... taken when somebody asks for it (round 6: a scan that is stepped by its ticket is never hashed)
    uint64_t ticket = 0;              // the announcement's ticket (erasor_hip_prefetch_node_rows -> erasor_hip_step_ticket)
    RowFmt fmt;                       // record layout of src (host scans)
    bool used = false;                // ev_done has been recorded at least once
    bool pose_valid = false;          // the scan was announced together with its pose (erasor_hip_prefetch_node)
    double pose_x = 0, pose_y = 0;    // T_body2origin translation (OMU.cpp:246-247): all fetch_VoI needs
    bool to_valid = false;            // ... and with T_origin2body (erasor_hip_announce_origin2body): the next step's gather can go ahead too
    float To[16] = {0};
};

struct erasor_hip_handle {
    erasor_params P;
    DP dp;
    int device = 0;
    hipStream_t stream = nullptr;   // main stream: step begin, SRT .. write-back
    // query chains alternate between two streams, so that two consecutive scans' chains overlap (three streams in all; the
    // runtime multiplexes streams onto 4 hardware queues by default -- a fifth stream shares one and serialises behind it)
    static constexpr int NQS_MAX = 4;
    hipStream_t qstream[NQS_MAX] = {nullptr, nullptr, nullptr, nullptr};
    int nqs = 2;  // query streams in use (ERASOR_HIP_QSTREAMS, 1..4)
    hipStream_t cstream = nullptr;  // host scans on their way into a query side (staged, asynchronous)
    // round 5: the stream of the EARLY write-back and of the next step's passes that run beside the per-bin launch (k_srt4, k_assemble_map<RES>,
    // k_voi_split / k_chunk_scan_* / k_voi_gather in overlap mode), and the events the two streams meet at
    hipStream_t bstream = nullptr;
    bool bstream_own = false;
    hipEvent_t ev_stats = nullptr, ev_srt4 = nullptr, ev_asm = nullptr, ev_early = nullptr, ev_scan = nullptr;
    unsigned n_chain = 0;
    // round 6: chains of announced scans are held back until `batch_n` of them can share one set of launches -- unless fewer than
    // `batch_lead` chains are in their queues in front of them (then latency matters more than queue time)
    int batch_n = 2, batch_lead = 3;
    unsigned long long n_batches = 0, n_batched_chains = 0;
    hipStream_t cur = nullptr;      // stream LAUNCH() currently targets
    int pend[MAX_AHEAD] = {0};  // query sides with a prefetched chain in flight (or held back for a shared set of launches), oldest first
    int npend = 0;
    // a scan announced by erasor_hip_prefetch_scan whose chain is not enqueued yet (the step in flight goes first)
    struct {
        bool valid = false, is_device = false;
        const void *src = nullptr;
        size_t n = 0;
        float Tl[16] = {0};
        int side = 0;
        uint64_t fp = 0;
        bool fp_valid = false;
        uint64_t ticket = 0;
        RowFmt fmt;
        bool pose_valid = false;
        double pose_x = 0, pose_y = 0;
        bool to_valid = false;
        float To[16] = {0};
    } ann;
    int last_ann_side = -1;  // the side of the most recent announcement (erasor_hip_announce_origin2body attaches to it)
    int step_side = -1;      // the side of the step in flight / of the last step collected: its arrays are read by that step's launches and by the getters
    uint64_t next_ticket = 1;
    // the NEXT step's VoI split, launched ahead (behind this step's k_step_end) when the next scan was announced with its pose
    struct {
        bool valid = false;
        double x = 0, y = 0;
        bool scan = false;               // the chunk scan was launched ahead as well (buffers below)
        const void *pvl = nullptr, *phl = nullptr;
        uint32_t scan_cap = 0;
        unsigned long long seq = 0;      // h->step_seq when it was launched
        unsigned long long epoch = 0;    // h->store_epoch then
        int curF = 0;                    // the F buffer it read
        uint32_t cap_chunks = 0;
        const void *vmask = nullptr, *hmask = nullptr, *cinfo = nullptr;
    } spec;
    // round 5: what was launched for the NEXT step beside this step's per-bin launch (OVERLAPPED steps): split, chunk scan and gather
    // without the late table's slots on bstream, then -- behind the per-bin launch -- k_late_gather (which also ends the step in
    // flight), the bucket histogram and its column scan.  The step that follows takes all of it if it is exactly the step assumed here.
    struct {
        bool valid = false;
        unsigned long long seq = 0, epoch = 0;
        int curF = 0, qside = -1;
        double x = 0, y = 0;
        float To[16] = {0};
        uint32_t cap_chunks = 0, cap_voi = 0, nbk = 0;
        bool stats_done = false;  // ... and the stable scatter + bin statistics + first Scan Ratio Test pass (the query chain was through)
        const void *vmask = nullptr, *hmask = nullptr, *lmask = nullptr, *cinfo = nullptr, *pvl = nullptr, *phl = nullptr, *voi_ego = nullptr,
                   *mb_hist = nullptr;
    } ov;
    unsigned long long n_ov_launched = 0, n_ov_used = 0;
    bool ov_mode = false;  // the last step used the reserved layout / the early stream: query chains keep to TWO streams then
    // round 5: a step is ~45 launches, ~30 of them the query chain of a scan announced ahead -- with the steps overlapped the HOST had
    // become the bound (180 us of launch calls per step against a 120 us chain on the device).  The chain's launches go to a worker
    // thread of the handle: the caller's thread dispatches a job (allocations, staging, bookkeeping stay with it) and goes on with the
    // main chain; whoever needs the chain's events first waits until the worker has recorded them (chain_wait).
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::function<int()>> jobs;
        bool stop = false;
        std::atomic<int> busy[NSIDE];  // jobs of a side that are not enqueued yet
        std::atomic<int> err{0};
    } *worker = nullptr;
    unsigned long long store_epoch = 0;  // bumped by everything that rewrites the map store outside a step
    unsigned long long n_spec_used = 0, n_spec_launched = 0;
    unsigned n_split_launch = 0;         // k_voi_split launches seen by the sampled roofline measurement (erasor_hip_profiling(h, 3))
    // mapgen state (mapgen.hpp:27-46): cloud_curr, cloud_map, the finished submaps (cloud_maps, concatenated)
    DBuf<float4> mg_curr, mg_map, mg_done, mg_tmp;
    uint64_t mg_ncurr = 0, mg_nmap = 0, mg_ndone = 0;
    double mg_leaf = 0.05;
    bool mg_large = false, mg_initial = true, mg_active = false;
    uint64_t mg_cnt_voxel = 0, mg_accum = 0;
    DBuf<uint32_t> mb_hist, mb_tot;   // the map's bucketing as a counting sort: [tiles][B + 1] table, [B + 1] totals
    DBuf<unsigned long long> lab_slots;  // [16][8] label tallies of the assemble kernels (one cache line per slot)
    double tm_span = 0, tm_gap = 0, tm_period = 0;  // erasor_hip_chain_timing: sums in microseconds
    unsigned long long tm_last_open = 0;
    uint64_t tm_n = 0, tm_ngap = 0, tm_nper = 0;
    // ERASOR_HIP_OVERLAP unset: the handle decides by measurement whether consecutive steps overlap (round 5 had a threshold fitted to the
    // five bench workloads here).  The period of a step -- end to end on the device's own clock, consecutive steps only -- is sampled in
    // blocks: plain, overlapped, overlapped, plain (OVA_W samples each; a sequence drifts -- its first steps revert more bins --, and
    // this order cancels a linear drift; the first OVA_SKIP samples after a change of mode are not counted), and the mode with the
    // shorter mean period runs on (plain has to lose by 2 %).  Measured again every OVA_AGAIN steps.  Results never depend on it.
    struct OvAuto {
        int mode = 0;            // 1: overlapped, 0: plain -- what the next step uses
        int block = 0;           // 0..3: the block being sampled (modes 0 1 1 0); 4: decided
        int skip = 4;            // samples still to be ignored (start-up, a change of mode)
        int n = 0;               // samples of the current block
        double sum[2] = {0, 0};
        double mx[2] = {0, 0};   // the largest sample of each mode: left out of its mean (one hiccup in eight samples must not decide 600 steps)
        int cnt[2] = {0, 0};
        double est[2] = {0, 0};  // the last decision's mean periods (0: not measured)
        unsigned long long since_decision = 0;
    } ova;
    unsigned long long tm_last_end = 0, tm_last_seq = 0;
    unsigned long long step_seq = 0;  // steps issued so far (k_step_end echoes it into the pinned block)
    // a step that has been enqueued (erasor_hip_step_async) and not collected yet (erasor_hip_step_wait)
    struct {
        bool active = false;
        unsigned long long seq = 0;
        uint64_t n_map_in = 0;
        uint32_t ns = 0;
        const uint32_t *sm_keys = nullptr;
        int flags = 0;
        const void *scan_src = nullptr;
        size_t n_scan = 0;
        bool src_is_device = false;
        float Tl[16], Tb[16], To[16];
        uint32_t nchunks = 0;     // chunk count of the step's own VoI pass (grid hint of a split launched ahead later)
        bool spec_launched = false;
        bool reserved = false;    // the write-back uses the reserved layout (round 5)
        bool deep = false;        // the caller had announced at least batch_lead nodes beyond this step (OvAuto samples only such steps)
    } fly;
    HostOut *pin = nullptr;         // pinned host block k_step_end reports into
    int bank = 0;                   // scratch bank of scan/radix helpers (0: query chains, 1: map chain)
    std::string err;
    bool have_map = false, have_step = false;
    bool poisoned = false;        // a step failed after the map store had been touched: only set_map makes the handle usable again
    bool q_passthrough = false;   // the last scan overflowed PCL's VoxelGrid indices (utils.cpp:88-91): start the next chains in pass-through mode

    // ---- map store ----
    uint32_t capMap = 0, capF = 0, capO = 0, capV = 0, capG = 0;
    uint32_t n_grow = 0;  // times the map-sized scratch was enlarged (grow_map_scratch)
    uint32_t B = 0;
    DBuf<float4> F[2];
    int curF = 0;
    DBuf<float2> Oxy, Ozi;
    uint32_t nF = 0, o_begin = 0;  // host mirrors (nF: the EXTENT of the VoI-resident region)
    uint32_t nFv = 0;              // ... the POINTS in it (round 5: a region written in the reserved layout holds holes, see DevState::nF_valid)
    // round 5, reserved layout (srt4_body / k_assemble_late): offsets with the reverted bins at full size, the late tables of the region in
    // F[curF] (late_idx) and of the one being written (late_idx ^ 1), the holes before every entry
    DBuf<uint32_t> out_offR, gres_off, late_holes[2];
    DBuf<LateEnt> late[2];
    int late_idx = 0;
    uint32_t n_late_F = 0;         // entries of late[late_idx] (0: F[curF] is dense)
    // large-scale mode (OMU.cpp:332-379): everything outside the current submap
    DBuf<float4> Cbuf;
    uint32_t nC = 0;
    bool submap_not_initialized = true;
    double submap_cx = 0, submap_cy = 0;
    uint32_t n_reassign = 0;
    uint64_t o_valid = 0;          // live outskirts entries
    // ---- VoI split ----
    DBuf<unsigned long long> vmask, hmask, lmask;  // (lmask: the late table's slots, overlapped steps)
    DBuf<uint32_t> cinfo, pvl, phl, topv, toph, topr;
    DBuf<OMeta> ometa;  // one record per outskirts chunk (bounding box, valid count): chunks outside the VoI circle are not read
    bool use_ometa = true;
    // ---- VoI-order arrays ----
    DBuf<float4> voi_ego, spts, rejected;
    DBuf<uint32_t> voi_key, voi_src, ssrc, rejected_src, grank, glist;
    DBuf<uint8_t> gflag;
    // ---- radix ----
    DBuf<uint32_t> rk_a, rk_b, rv_a, rv_b, hist, hist_l, hist_t, hist2, hist2_l, hist2_t, dn;
    // ---- bins ----
    DBuf<uint32_t> moff, mcnt, rev_idx, rev_list, vox_off, nvox, ng, out_off, ground_off, rej_off, crej_off;
    DBuf<uint32_t> out_off0, rev_before;  // the layout's R-GPF-independent part (k_srt4 -> k_assemble_map<., true>)
    DBuf<float> mmin, mmax, plane_n;
    DBuf<double> plane_d;
    DBuf<uint8_t> st1, st1b, status, action;  // (st1b: first-pass status | 0x80 written by k_bin_stats_srt)
    // ---- query side: everything the voxelisation / bucketing of ONE scan owns.  Two sets, so that the query chain of the
    // next scan (erasor_hip_prefetch_scan) can run while the current step is in its map-side stages ----
    QSide q[NSIDE];
    int qi = 0;                      // the set the calls below work on
    DBuf<float4> curr_rejected;
    DBuf<unsigned long long> dbg_stamps;  // optional cycle stamps of the first finished segment (ERASOR_HIP_SORT_STAMPS)
    // ---- per-bin scratch (R-GPF / bin voxelise global paths) ----
    DBuf<uint32_t> gsK, gsV, gsL, gsR, gsK2, gsV2;
    DBuf<uint32_t> gsH;
    DBuf<float4> gsC, vox_out;
    // ---- round 5: what consecutive steps must not share (an overlapped step's early passes run while the step in flight still reads its
    // own): VoI-order arrays, bin offsets, state, counters, label tallies.  The names above / below are the CURRENT step's set; `alt` is
    // the other one, where the passes launched ahead write; the two change places when a step is enqueued (swap_sides)
    struct {
        DBuf<float4> voi_ego, spts;
        DBuf<uint32_t> voi_key, voi_src, moff, ssrc, rk_a, mcnt;
        DBuf<float> mmin, mmax;
        DBuf<uint8_t> st1b;
        DBuf<DevState> d_st;
        DBuf<Counters> d_ctr;
        DBuf<unsigned long long> lab_slots;
    } alt;
    // ---- state ----
    DBuf<DevState> d_st, d_st_get;  // (d_st_get: a copy of the last finished step's state for the getters, see assemble_egocentric)
    DBuf<Counters> d_ctr;
    DevState st;
    Counters ctr;
    Xf Tl2b, Tb2o, To2b;
    uint32_t last_n_voi = 0, last_nq = 0, last_n_scan = 0;
    const uint32_t *last_skeys = nullptr;  // sorted VoI keys of the last step
    erasor_step_result last_res;
    // ---- profiling ----
    int prof = 0;  // 0 off, 1 every kernel, 2 only the roofline kernel (voi_split)
    std::vector<std::string> prof_names;
    std::map<std::string, int> prof_ids;
    std::vector<ProfEntry> prof_tab;
    std::vector<PendingEvt> pending;
    std::vector<hipEvent_t> evt_pool;
};

// (round 5: the worker thread launches a query chain with its OWN target stream and query side -- the caller's thread goes on using
// the handle's --: LAUNCH() and Q() look here first)
struct TlCtx {
    const erasor_hip_handle *h = nullptr;
    hipStream_t cur = nullptr;
    int qi = 0;
};
static thread_local TlCtx g_tl;
#define Q(h) ((h)->q[g_tl.h == (h) ? g_tl.qi : (h)->qi])
#define CUR(h) (g_tl.h == (h) ? g_tl.cur : (h)->cur)
// between erasor_hip_step_async and erasor_hip_step_wait the handle takes no other call: the step's scratch, query side and
// host mirror are in use
#define NOFLY(h)                                                                                              \
    do {                                                                                                      \
        if ((h) && (h)->fly.active) {                                                                         \
            (h)->err = "a step is in flight (erasor_hip_step_async): call erasor_hip_step_wait first";        \
            return ERASOR_E_STATE;                                                                            \
        }                                                                                                     \
    } while (0)

namespace {

const bool g_debug_sync = getenv("ERASOR_HIP_DEBUG_SYNC") != nullptr;  // bring-up aid: sync + log every launch

#define HIPC(h, call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return ERASOR_E_NO_DEVICE;                                                        \
        }                                                                                     \
    } while (0)

template <class T>
int ensure(erasor_hip_handle *h, DBuf<T> &b, size_t n, bool keep = false) {
    if (n <= b.cap) return 0;
    T *np = nullptr;
    size_t ncap = n + n / 8 + 64;
    HIPC(h, hipMalloc((void **)&np, ncap * sizeof(T)));
    if (keep && b.p) HIPC(h, hipMemcpyAsync(np, b.p, b.cap * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    if (b.p) {
        HIPC(h, hipStreamSynchronize(h->stream));
        if (h->bstream) HIPC(h, hipStreamSynchronize(h->bstream));  // (passes launched ahead of the next step may be writing the old block)
        (void)hipFree(b.p);
    }
    b.p = np;
    b.cap = ncap;
    return 0;
}
template <class T>
void release(DBuf<T> &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

hipEvent_t get_evt(erasor_hip_handle *h) {
    if (!h->evt_pool.empty()) {
        hipEvent_t e = h->evt_pool.back();
        h->evt_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
int prof_id(erasor_hip_handle *h, const char *name) {
    auto it = h->prof_ids.find(name);
    if (it != h->prof_ids.end()) return it->second;
    const int id = (int)h->prof_names.size();
    h->prof_names.push_back(name);
    h->prof_tab.push_back(ProfEntry());
    h->prof_ids[name] = id;
    return id;
}
// fold finished event pairs into the per-name totals.  Pairs of launches that are still in flight (query chains of scans
// announced ahead) stay pending; `force` waits for them (profile_get / reset, after the streams were synchronised).
void prof_collect(erasor_hip_handle *h, bool force = false) {
    if (!force && h->pending.size() < 256) return;  // (not on every step: an elapsed-time query costs microseconds)
    std::vector<PendingEvt> keep;
    for (auto &p : h->pending) {
        if (!force && hipEventQuery(p.b) == hipErrorNotReady) {
            keep.push_back(p);
            continue;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->prof_tab[p.name_id].ms += ms;
            h->prof_tab[p.name_id].launches += 1;
        }
        h->evt_pool.push_back(p.a);
        h->evt_pool.push_back(p.b);
    }
    h->pending.swap(keep);
}

// kernel launch with optional event bracketing on the handle's stream
#define LAUNCH(h, name, kern, grid, block, ...)                                   \
    do {                                                                          \
        PendingEvt pe_;                                                           \
        const bool prof_ = (h)->prof == 1 || ((h)->prof == 2 && strncmp(name, "voi_split", 9) == 0); \
        if (prof_) {                                                              \
            pe_.name_id = prof_id((h), name);                                     \
            pe_.a = get_evt(h);                                                   \
            pe_.b = get_evt(h);                                                   \
            (void)hipEventRecord(pe_.a, CUR(h));                                  \
        }                                                                         \
        if (g_debug_sync) fprintf(stderr, "[erasor_hip] launch %s grid=%u\n", name, (unsigned)dim3(grid).x); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, CUR(h), __VA_ARGS__);   \
        if (g_debug_sync) {                                                       \
            hipError_t e2_ = hipStreamSynchronize(CUR(h));                        \
            if (e2_ != hipSuccess) fprintf(stderr, "[erasor_hip]   -> %s\n", hipGetErrorString(e2_)); \
        }                                                                         \
        if (prof_) {                                                              \
            (void)hipEventRecord(pe_.b, CUR(h));                                  \
            (h)->pending.push_back(pe_);                                          \
        }                                                                         \
    } while (0)

inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
inline uint32_t rup(uint64_t a, uint64_t b) { return (uint32_t)(((a + b - 1) / b) * b); }

Xf to_xf(const float T[16]) {
    Xf x;
    for (int k = 0; k < 12; ++k) x.m[k] = T[k];
    return x;
}

void fill_dp(erasor_hip_handle *h) {
    const erasor_params &p = h->P;
    DP &d = h->dp;
    d.max_r = p.max_range;
    d.R = p.num_rings;
    d.S = p.num_sectors;
    d.B = p.num_rings * p.num_sectors;
    d.ring_size = p.max_range / p.num_rings;    // erasor.h:63
    d.sector_size = 2 * PI_REF / p.num_sectors;  // erasor.h:64
    d.max_h = p.max_h;
    d.min_h = p.min_h;
    d.th_bin_max_h = p.th_bin_max_h;
    d.srt_thr = p.scan_ratio_threshold;
    d.gf_dist = p.gf_dist_thr;
    d.gf_seeds_h = p.gf_th_seeds_height;
    const double R = p.voi_max_range > 0 ? p.voi_max_range : p.max_range;
    d.voi_r2 = R * R;  // pow(max_range_ + margin, 2), margin = 0 (OMU.cpp:385,391)
    d.num_lowest = p.num_lowest_pts;
    d.min_pts = p.minimum_num_pts;
    d.gf_iter = p.gf_iter;
    d.gf_lpr = p.gf_num_lpr;
    d.version = p.version;
    d.leaf_map = (float)p.map_voxel_size;     // setLeafSize(float...) narrowing
    d.leaf_query = (float)p.query_voxel_size;
}

// two-level exclusive scan helper (n on host or device)
// k_bbox ends with six same-line atomics per workgroup (~6 ns each, serialised): 475 workgroups for a 121 k-point scan
// were 17 us of atomics around 2 MB of loads.  Eight points per thread, at most 512 workgroups.
static inline uint32_t bbox_grid(uint32_t n) { return std::max(1u, std::min<uint32_t>(cdiv(n, 2048), 512)); }

int scan_u32(erasor_hip_handle *h, const uint32_t *in, uint32_t *out_local, uint32_t *tops, uint32_t n_host_max, uint32_t n_host,
             const uint32_t *n_dev, uint32_t *total_out, const char *tag) {
    const uint32_t nb = std::max(1u, cdiv(n_host_max, 1024));
    LAUNCH(h, tag, k_scan_local, nb, 256, in, out_local, tops, n_host, n_dev);
    LAUNCH(h, tag, k_scan_top, 1, 1024, tops, n_host, n_dev, total_out);
    return 0;
}

// stable LSD radix sort of keys[0..n): n = *n_dev when given (n_ub is then only an upper bound that sizes grids and
// scratch), else n_ub itself.  Returns pointers to the sorted keys / permutation.
int radix_sort(erasor_hip_handle *h, const uint32_t *keys_in, uint32_t n_ub, const uint32_t *n_dev, int bits, uint32_t *ka, uint32_t *kb,
               uint32_t *va, uint32_t *vb, const uint32_t **skeys, const uint32_t **sperm, const char *tag) {
    const uint32_t nblk = std::max(1u, cdiv(n_ub, RTILE));
    const uint32_t nhist = 256u * nblk;
    DBuf<uint32_t> &H = h->bank ? h->hist2 : h->hist, &HL = h->bank ? h->hist2_l : h->hist_l, &HT = h->bank ? h->hist2_t : h->hist_t;
    if (ensure(h, H, nhist) || ensure(h, HL, nhist) || ensure(h, HT, cdiv(nhist, 1024) + 2)) return ERASOR_E_NO_DEVICE;
    const uint32_t *kin = keys_in;
    const uint32_t *vin = nullptr;
    uint32_t *kout = ka, *vout = va;
    uint32_t *nhist_dev = h->dn.p + 8 + h->bank;
    for (int shift = 0; shift < bits; shift += 8) {
        LAUNCH(h, tag, k_radix_hist, nblk, 256, kin, n_ub, n_dev, shift, H.p, nhist_dev);
        scan_u32(h, H.p, HL.p, HT.p, nhist, nhist, n_dev ? (const uint32_t *)nhist_dev : nullptr, nullptr, tag);
        LAUNCH(h, tag, k_radix_scatter, nblk, 256, kin, vin, n_ub, n_dev, shift, (const uint32_t *)HL.p, (const uint32_t *)HT.p,
               kout, vout);
        kin = kout;
        vin = vout;
        kout = (kout == ka) ? kb : ka;
        vout = (vout == va) ? vb : va;
    }
    *skeys = kin;
    *sperm = vin;
    return 0;
}

int key_bits(uint32_t nbuckets) {
    int b = 1;
    while ((1u << b) < nbuckets) ++b;
    return b;
}

int alloc_bins(erasor_hip_handle *h) {
    const size_t B = h->B;
    int rc = 0;
    const int keep = h->qi;
    for (h->qi = 0; h->qi < NSIDE; ++h->qi) {
        rc |= ensure(h, Q(h).qoff, B + 2) | ensure(h, Q(h).ccnt, B) | ensure(h, Q(h).cmin, B) | ensure(h, Q(h).cmax, B);
        rc |= ensure(h, Q(h).bb, 8) | ensure(h, Q(h).qgrid, 1) | ensure(h, Q(h).esqs, 1) | ensure(h, Q(h).qb_tot, B + 2);
        rc |= ensure(h, Q(h).d_nvox, 4) | ensure(h, Q(h).d_qctr, 1);
    }
    h->qi = keep;
    rc |= ensure(h, h->moff, B + 4) | ensure(h, h->alt.moff, B + 4) | ensure(h, h->mcnt, B) | ensure(h, h->alt.mcnt, B);
    rc |= ensure(h, h->alt.mmin, B) | ensure(h, h->alt.mmax, B) | ensure(h, h->alt.st1b, B + 8);
    rc |= ensure(h, h->mmin, B) | ensure(h, h->mmax, B);
    rc |= ensure(h, h->st1, B) | ensure(h, h->status, B) | ensure(h, h->action, B) | ensure(h, h->rev_idx, B) | ensure(h, h->rev_list, B);
    rc |= ensure(h, h->out_off0, B + 2) | ensure(h, h->rev_before, B + 2) | ensure(h, h->st1b, B + 8);
    rc |= ensure(h, h->out_offR, B + 2) | ensure(h, h->gres_off, B + 2);
    for (int k = 0; k < 2; ++k) rc |= ensure(h, h->late[k], 2 * B + 4) | ensure(h, h->late_holes[k], 2 * B + 4);
    rc |= ensure(h, h->vox_off, B) | ensure(h, h->nvox, B) | ensure(h, h->ng, B) | ensure(h, h->out_off, B) | ensure(h, h->ground_off, B);
    rc |= ensure(h, h->rej_off, B) | ensure(h, h->crej_off, B);
    rc |= ensure(h, h->plane_n, B * (size_t)std::max(h->P.gf_iter, 1) * 3) | ensure(h, h->plane_d, B * (size_t)std::max(h->P.gf_iter, 1));
    rc |= ensure(h, h->d_st, 1) | ensure(h, h->d_ctr, 1) | ensure(h, h->dn, 16) | ensure(h, h->lab_slots, 128) | ensure(h, h->mb_tot, B + 4);
    rc |= ensure(h, h->alt.d_st, 1) | ensure(h, h->alt.d_ctr, 1) | ensure(h, h->alt.lab_slots, 128);
    return rc ? ERASOR_E_NO_DEVICE : 0;
}

// capacity-sized buffers that depend on the map size
int alloc_map(erasor_hip_handle *h, uint32_t n) {
    // ERASOR_HIP_MAP_SLACK (test hook): head-room in points, so that a small test map outgrows it and exercises grow_map_scratch
    static const long slack_env = getenv("ERASOR_HIP_MAP_SLACK") ? atol(getenv("ERASOR_HIP_MAP_SLACK")) : -1;
    const uint64_t slack = slack_env >= 0 ? (uint64_t)slack_env : std::max<uint64_t>(n / 4, 1u << 20);
    const uint64_t capMap = (uint64_t)n + slack;
    if (2 * capMap + (1u << 22) >= 0xFFFFFFF0ull) {
        h->err = "map too large for 32-bit indexing";
        return ERASOR_E_INVALID;
    }
    h->capMap = (uint32_t)capMap;
    h->capV = h->capMap;
    h->capO = rup(capMap + std::max<uint64_t>(capMap / 2, 1u << 20), CHUNK);
    int rc = 0;
    rc |= ensure(h, h->Oxy, h->capO) | ensure(h, h->Ozi, h->capO);
    const uint32_t V = h->capV;
    rc |= ensure(h, h->voi_ego, V) | ensure(h, h->spts, V) | ensure(h, h->voi_key, V) | ensure(h, h->voi_src, V) | ensure(h, h->ssrc, V);
    rc |= ensure(h, h->alt.voi_ego, V) | ensure(h, h->alt.voi_key, V) | ensure(h, h->alt.voi_src, V);
    rc |= ensure(h, h->alt.spts, V) | ensure(h, h->alt.ssrc, V) | ensure(h, h->alt.rk_a, V);
    rc |= ensure(h, h->rejected, V) | ensure(h, h->rejected_src, V) | ensure(h, h->grank, V) | ensure(h, h->glist, V) | ensure(h, h->gflag, V);
    rc |= ensure(h, h->rk_a, V) | ensure(h, h->rk_b, V) | ensure(h, h->rv_a, V) | ensure(h, h->rv_b, V);
    return rc ? ERASOR_E_NO_DEVICE : 0;
}

// the map outgrew the capacity chosen at set_map: re-allocate every map-sized scratch array (none of them carries state from
// one step to the next; the outskirts region grows on its own in rebuild_outskirts, the F buffers in alloc_step)
int grow_map_scratch(erasor_hip_handle *h, uint64_t need) {
    static const long slack_env = getenv("ERASOR_HIP_MAP_SLACK") ? atol(getenv("ERASOR_HIP_MAP_SLACK")) : -1;
    const uint64_t capMap = need + (slack_env >= 0 ? (uint64_t)slack_env : std::max<uint64_t>(need / 2, 1u << 20));
    if (2 * capMap + (1u << 22) >= 0xFFFFFFF0ull) {
        h->err = "map too large for 32-bit indexing";
        return ERASOR_E_CAPACITY;
    }
    ++h->n_grow;
    h->capMap = (uint32_t)capMap;
    h->capV = h->capMap;
    const uint32_t V = h->capV;
    int rc = 0;
    rc |= ensure(h, h->voi_ego, V) | ensure(h, h->spts, V) | ensure(h, h->voi_key, V) | ensure(h, h->voi_src, V) | ensure(h, h->ssrc, V);
    rc |= ensure(h, h->alt.voi_ego, V) | ensure(h, h->alt.voi_key, V) | ensure(h, h->alt.voi_src, V);
    rc |= ensure(h, h->alt.spts, V) | ensure(h, h->alt.ssrc, V) | ensure(h, h->alt.rk_a, V);
    rc |= ensure(h, h->rejected, V) | ensure(h, h->rejected_src, V) | ensure(h, h->grank, V) | ensure(h, h->glist, V) | ensure(h, h->gflag, V);
    rc |= ensure(h, h->rk_a, V) | ensure(h, h->rk_b, V) | ensure(h, h->rv_a, V) | ensure(h, h->rv_b, V);
    h->have_step = false;  // the previous step's read-back clouds lived in these arrays
    return rc ? ERASOR_E_NO_DEVICE : 0;
}

static int alloc_scan_side(erasor_hip_handle *h, uint32_t ns) {
    if (ns <= Q(h).capS && Q(h).capS) return 0;
    const uint32_t S = ns + ns / 4 + 1024;
    Q(h).capS = S;
    int rc = 0;
    rc |= ensure(h, Q(h).scan, S) | ensure(h, Q(h).cent, S) | ensure(h, Q(h).query, S) | ensure(h, Q(h).sq, S) | ensure(h, h->curr_rejected, S, /*keep: the last step's CURR_REJECTED*/ true);
    rc |= ensure(h, Q(h).qk_a, S) | ensure(h, Q(h).qk_b, S) | ensure(h, Q(h).qv_a, S) | ensure(h, Q(h).qv_b, S) | ensure(h, Q(h).qposL, S) | ensure(h, Q(h).qposR, S);
    rc |= ensure(h, Q(h).qflag, S) | ensure(h, Q(h).qpl, S) | ensure(h, Q(h).qtops, S / 1024 + 4) | ensure(h, Q(h).run_begin, S + 1) | ensure(h, Q(h).ukeys, S);
    rc |= ensure(h, Q(h).qkey, S) | ensure(h, Q(h).qhead, S / 32 + 8);
    Q(h).hbits = 10;  // voxel hash table: >= 2 slots per possible voxel
    while ((1ull << Q(h).hbits) < 2ull * S) ++Q(h).hbits;
    rc |= ensure(h, Q(h).hkey, (size_t)1 << Q(h).hbits) | ensure(h, Q(h).hval, (size_t)1 << Q(h).hbits);
    if (getenv("ERASOR_HIP_SORT_STAMPS")) rc |= ensure(h, h->dbg_stamps, 96);
    rc |= ensure(h, Q(h).wseg0, WSEG_MAX) | ensure(h, Q(h).wseg1, WSEG_MAX) | ensure(h, Q(h).wstate, 1) | ensure(h, Q(h).wtileL, WTILES_MAX) | ensure(h, Q(h).wtileR, WTILES_MAX);
    rc |= ensure(h, Q(h).esq0, 65536) | ensure(h, Q(h).esq1, 65536) | ensure(h, Q(h).esq2, 65536) | ensure(h, Q(h).essmall, 65536);
    return rc ? ERASOR_E_NO_DEVICE : 0;
}

// The query side in use gets room for a scan of ns points -- and, round 6, every IDLE side with it: eight sides take turns, and a side's
// first scan used to pay for ~40 device allocations (and a pinned staging buffer) in the middle of a sequence, 5-7 ms once per side: a
// visible share of a 20-node pass (erasor_offline_demo --bench: 0.35 or 0.70 ms per callback depending on where the first uses fell).
int alloc_scan(erasor_hip_handle *h, uint32_t ns) {
    int rc = alloc_scan_side(h, ns);
    if (rc || g_tl.h == h) return rc;  // (the worker thread never allocates)
    const int me = h->qi;
    for (int k = 0; k < NSIDE && !rc; ++k) {
        QSide &q = h->q[k];
        // (NOT the side of the step in flight or last collected: the Scan Ratio Test, the per-bin launch, the write-back and the getters read its
        // bucketed scan -- a side whose chain has run out is not an idle side; found by the 240-walk soak, not by the suite's four walks)
        if (k == me || k == h->step_side || (q.capS && ns <= q.capS) || q.held || (h->ann.valid && h->ann.side == k)) continue;
        bool busy = h->worker && h->worker->busy[k].load(std::memory_order_acquire) > 0;
        for (int j = 0; j < h->npend; ++j) busy = busy || h->pend[j] == k;
        if (busy || (q.used && hipEventQuery(q.ev_done) != hipSuccess)) continue;  // (a dropped chain may still be running on it)
        h->qi = k;
        rc = alloc_scan_side(h, ns);
    }
    h->qi = me;
    return rc;
}

// per-step scratch whose size depends on (VoI size + query size)
int alloc_step(erasor_hip_handle *h, uint32_t n_voi, uint32_t nq) {
    const size_t G = (size_t)n_voi + nq + 8;
    int rc = 0;
    // (twice: in the fused launch k_revert_bins the per-bin voxelisation's global-memory path works in the upper half, see there)
    rc |= ensure(h, h->gsK, 2 * G) | ensure(h, h->gsV, 2 * G) | ensure(h, h->gsL, 2 * G) | ensure(h, h->gsR, 2 * G) | ensure(h, h->gsK2, 2 * G) |
          ensure(h, h->gsV2, 2 * G);
    rc |= ensure(h, h->gsH, 2 * (G / 32 + 2 * (size_t)h->B + 16)) | ensure(h, h->gsC, G) | ensure(h, h->vox_out, G);
    // (round 5: the reserved layout keeps mc + cc slots for a reverted bin and mc for its ground, twice that where its points may leave
    // the next VoI: at most 4 n_voi + 2 nq in all)
    const size_t needF = (h->P.version == 3 ? 4 : 2) * (size_t)n_voi + 2 * (size_t)nq + 64;
    rc |= ensure(h, h->F[h->curF ^ 1], needF);
    return rc ? ERASOR_E_NO_DEVICE : 0;
}

// Everything that rewrites the outskirts other than a step voids the chunk records (OMeta: bounding box + valid count per chunk); the
// next VoI pass reads every chunk and rebuilds them.  ERASOR_HIP_NO_OMETA=1: no records, every pass reads everything (A/B).
static int ometa_reset(erasor_hip_handle *h) {
    static const bool off = getenv("ERASOR_HIP_NO_OMETA") != nullptr;
    h->use_ometa = !off;
    if (off) return ERASOR_OK;
    const size_t nrec = (size_t)h->capO / CHUNK + 8;
    if (ensure(h, h->ometa, nrec)) return ERASOR_E_NO_DEVICE;
    HIPC(h, hipMemsetAsync(h->ometa.p, 0, nrec * sizeof(OMeta), h->stream));
    return ERASOR_OK;
}

// the POINTS of the VoI-resident region, in order, into dst[0 .. nFv) on the device (main stream; returns once they are there): the
// region as it lies when it is dense, a stable compaction when the reserved layout has left holes in it (round 5)
int copy_F_dense(erasor_hip_handle *h, float4 *dst) {
    if (!h->nF) return ERASOR_OK;
    if (h->nFv == h->nF) {
        HIPC(h, hipMemcpyAsync(dst, h->F[h->curF].p, (size_t)h->nF * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        return ERASOR_OK;
    }
    DBuf<uint32_t> flag, pl, tops;
    if (ensure(h, flag, (size_t)h->nF + 1) || ensure(h, pl, (size_t)h->nF + 1) || ensure(h, tops, h->nF / 1024 + 4)) return ERASOR_E_NO_DEVICE;
    hipStream_t keep = h->cur;
    h->cur = h->stream;
    LAUNCH(h, "f_dense", k_f_valid, cdiv(h->nF, 256), 256, (const float4 *)h->F[h->curF].p, h->nF, flag.p);
    scan_u32(h, flag.p, pl.p, tops.p, h->nF, h->nF, nullptr, nullptr, "f_dense");
    LAUNCH(h, "f_dense", k_f_compact, cdiv(h->nF, 256), 256, (const float4 *)h->F[h->curF].p, h->nF, (const uint32_t *)flag.p, (const uint32_t *)pl.p,
           (const uint32_t *)tops.p, dst);
    h->cur = keep;
    const hipError_t e = hipStreamSynchronize(h->stream);
    release(flag);
    release(pl);
    release(tops);
    if (e != hipSuccess) {
        h->err = std::string("copy_F_dense: ") + hipGetErrorString(e);
        return ERASOR_E_NO_DEVICE;
    }
    return ERASOR_OK;
}

// rebuild the outskirts region without tombstones at the end of the buffer (stable)
int rebuild_outskirts(erasor_hip_handle *h, uint32_t min_front_room) {
    ++h->store_epoch;
    if (getenv("ERASOR_HIP_CHAIN_STAMPS"))
        fprintf(stderr, "[rebuild_outskirts] o_begin %u, capO %u, o_valid %llu, nF %u (points %u), front room wanted %u\n", h->o_begin, h->capO,
                (unsigned long long)h->o_valid, h->nF, h->nFv, min_front_room);
    if (h->bstream) HIPC(h, hipStreamSynchronize(h->bstream));  // (passes launched ahead of the next step write into the region's front)
    HIPC(h, hipStreamSynchronize(h->stream));
    const uint32_t span = h->capO - h->o_begin;
    DBuf<uint32_t> flag, pl, tops;
    DBuf<float4> tmp;
    if (ensure(h, flag, span + 1) || ensure(h, pl, span + 1) || ensure(h, tops, span / 1024 + 4)) return ERASOR_E_NO_DEVICE;
    uint32_t nvalid = 0;
    if (span) {
        LAUNCH(h, "o_rebuild", k_o_valid, cdiv(span, 256), 256, (const float2 *)h->Oxy.p, h->o_begin, h->capO, flag.p);
        scan_u32(h, flag.p, pl.p, tops.p, span, span, nullptr, h->dn.p, "o_rebuild");
        HIPC(h, hipMemcpyAsync(&nvalid, h->dn.p, 4, hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
    }
    if (ensure(h, tmp, (size_t)nvalid + 1)) return ERASOR_E_NO_DEVICE;
    if (span) LAUNCH(h, "o_rebuild", k_o_compact, cdiv(span, 256), 256, (const float2 *)h->Oxy.p, (const float2 *)h->Ozi.p, h->o_begin,
                     h->capO, (const uint32_t *)flag.p, (const uint32_t *)pl.p, (const uint32_t *)tops.p, tmp.p);
    uint32_t need = rup((uint64_t)nvalid + min_front_room + CHUNK, CHUNK);
    if (need > h->capO) {
        const uint32_t ncap = rup((uint64_t)need + need / 2, CHUNK);
        HIPC(h, hipStreamSynchronize(h->stream));
        release(h->Oxy);
        release(h->Ozi);
        if (ensure(h, h->Oxy, ncap) || ensure(h, h->Ozi, ncap)) return ERASOR_E_NO_DEVICE;
        h->capO = ncap;
    }
    h->o_begin = h->capO - nvalid;
    if (nvalid) LAUNCH(h, "o_rebuild", k_store_outskirts, cdiv(nvalid, 256), 256, (const float4 *)tmp.p, nvalid, h->Oxy.p, h->Ozi.p, h->o_begin);
    {
        const int rc_m = ometa_reset(h);
        if (rc_m) return rc_m;
    }
    HIPC(h, hipStreamSynchronize(h->stream));
    h->o_valid = nvalid;
    release(flag);
    release(pl);
    release(tops);
    release(tmp);
    return 0;
}

int push_state(erasor_hip_handle *h);

// set_submap (OMU.cpp:360-379) over [F | outskirts | complement] in that (logical) order: stable partition by the box test
// into the new submap (stored as outskirts, F empty) and the new complement.  Rare (every ~submap_size/2 of travel).
int split_submap(erasor_hip_handle *h, double x, double y) {
    ++h->store_epoch;
    if (h->bstream) HIPC(h, hipStreamSynchronize(h->bstream));
    HIPC(h, hipStreamSynchronize(h->stream));
    const uint64_t total64 = (uint64_t)h->nFv + h->o_valid + h->nC;
    if (total64 > 0x7FFFFFF0ull) {
        h->err = "map too large for 32-bit indexing";
        return ERASOR_E_INVALID;
    }
    const uint32_t total = (uint32_t)total64;
    DBuf<float4> G, newC;
    DBuf<uint32_t> flag, pl, tops;
    if (ensure(h, G, (size_t)total + 1) || ensure(h, flag, (size_t)total + 1) || ensure(h, pl, (size_t)total + 1) ||
        ensure(h, tops, total / 1024 + 4))
        return ERASOR_E_NO_DEVICE;
    // logical global map = *map_arranged_ + *map_arranged_complement_ (OMU.cpp:349)
    {
        const int rc_f = copy_F_dense(h, G.p);
        if (rc_f) return rc_f;
    }
    const uint32_t span = h->capO - h->o_begin;
    if (span && h->o_valid) {
        DBuf<uint32_t> f2, p2, t2;
        if (ensure(h, f2, span + 1) || ensure(h, p2, span + 1) || ensure(h, t2, span / 1024 + 4)) return ERASOR_E_NO_DEVICE;
        LAUNCH(h, "submap", k_o_valid, cdiv(span, 256), 256, (const float2 *)h->Oxy.p, h->o_begin, h->capO, f2.p);
        scan_u32(h, f2.p, p2.p, t2.p, span, span, nullptr, nullptr, "submap");
        LAUNCH(h, "submap", k_o_compact, cdiv(span, 256), 256, (const float2 *)h->Oxy.p, (const float2 *)h->Ozi.p, h->o_begin, h->capO,
               (const uint32_t *)f2.p, (const uint32_t *)p2.p, (const uint32_t *)t2.p, G.p + h->nFv);
        HIPC(h, hipStreamSynchronize(h->stream));
        release(f2);
        release(p2);
        release(t2);
    }
    if (h->nC) HIPC(h, hipMemcpyAsync(G.p + h->nFv + h->o_valid, h->Cbuf.p, (size_t)h->nC * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    uint32_t n_sub = 0;
    if (total) {
        LAUNCH(h, "submap", k_box_flag, cdiv(total, 256), 256, (const float4 *)G.p, total, x, y, h->P.submap_size, flag.p);
        scan_u32(h, flag.p, pl.p, tops.p, total, total, nullptr, h->dn.p, "submap");
        HIPC(h, hipMemcpyAsync(&n_sub, h->dn.p, 4, hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
    }
    const uint32_t n_cmp = total - n_sub;
    const uint32_t needO = rup((uint64_t)n_sub + std::max<uint64_t>(n_sub / 2, 1u << 20), CHUNK);
    if (needO > h->capO) {  // nothing to preserve: the region is rewritten from G
        HIPC(h, hipStreamSynchronize(h->stream));
        release(h->Oxy);
        release(h->Ozi);
        if (ensure(h, h->Oxy, needO) || ensure(h, h->Ozi, needO)) return ERASOR_E_NO_DEVICE;
        h->capO = needO;
    }
    if (ensure(h, newC, (size_t)n_cmp + 1)) return ERASOR_E_NO_DEVICE;
    const uint32_t dst0 = h->capO - n_sub;
    if (total)
        LAUNCH(h, "submap", k_box_partition, cdiv(total, 256), 256, (const float4 *)G.p, total, (const uint32_t *)flag.p, (const uint32_t *)pl.p,
               (const uint32_t *)tops.p, h->Oxy.p, h->Ozi.p, dst0, newC.p);
    {
        const int rc_m = ometa_reset(h);
        if (rc_m) return rc_m;
    }
    h->nF = 0;
    h->nFv = 0;
    h->n_late_F = 0;
    h->o_begin = dst0;
    h->o_valid = n_sub;
    memset(&h->st, 0, sizeof(h->st));
    int rc = push_state(h);
    if (rc) return rc;
    if (n_sub) {
        DevState *ds = h->d_st.p;
        LAUNCH(h, "submap", k_count_labels_zi, std::min<uint32_t>(cdiv(n_sub, 256), 1024), 256, (const float2 *)(h->Ozi.p + dst0), n_sub,
               &ds->O_static, &ds->O_dynamic);
    }
    HIPC(h, hipMemcpyAsync(&h->st, h->d_st.p, sizeof(DevState), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    release(h->Cbuf);
    h->Cbuf = newC;
    h->nC = n_cmp;
    release(G);
    release(flag);
    release(pl);
    release(tops);
    ++h->n_reassign;
    return 0;
}

// reassign_submap (OMU.cpp:332-358)
int reassign_submap(erasor_hip_handle *h, double pose_x, double pose_y) {
    if (h->submap_not_initialized) {
        int rc = split_submap(h, pose_x, pose_y);
        if (rc) return rc;
        h->submap_cx = pose_x;
        h->submap_cy = pose_y;
        h->submap_not_initialized = false;
    } else {
        const double diff_x = fabs(h->submap_cx - pose_x), diff_y = fabs(h->submap_cy - pose_y);
        const double half_size = h->P.submap_size / 2.0;
        if ((diff_x > half_size) || (diff_y > half_size)) {
            int rc = split_submap(h, pose_x, pose_y);
            if (rc) return rc;
            h->submap_cx = pose_x;
            h->submap_cy = pose_y;
        }
    }
    return 0;
}

// large-scale mode: would the node at (x, y) move the submap (reassign_submap above)?  Then nothing of its step can be launched ahead.
bool submap_would_move(const erasor_hip_handle *h, double pose_x, double pose_y) {
    if (!h->P.is_large_scale) return false;
    if (h->submap_not_initialized) return true;
    const double half_size = h->P.submap_size / 2.0;
    return fabs(h->submap_cx - pose_x) > half_size || fabs(h->submap_cy - pose_y) > half_size;
}

int push_state(erasor_hip_handle *h) {
    h->st.nF = h->nF;
    h->st.nF_valid = h->nFv;
    h->st.o_begin = h->o_begin;
    HIPC(h, hipMemcpyAsync(h->d_st.p, &h->st, sizeof(DevState), hipMemcpyHostToDevice, h->stream));
    return 0;
}

}  // namespace

// =================================================================================================
extern "C" {

const char *erasor_hip_version(void) { return "erasor_hip 0.1 (gfx950, hand-written HIP; no CPU fallback)"; }

int erasor_hip_params_default(erasor_params *p) {
    if (!p) return ERASOR_E_INVALID;
    memset(p, 0, sizeof(*p));
    p->max_range = 10.0;   // erasor.h:47
    p->num_rings = 20;     // :48
    p->num_sectors = 60;   // :49
    p->max_h = 3.0;        // :50
    p->min_h = 0.0;        // :51
    p->th_bin_max_h = 0.39;
    p->scan_ratio_threshold = 0.22;
    p->num_lowest_pts = 5;
    p->minimum_num_pts = 4;
    p->rejection_ratio = 0.33;
    p->gf_dist_thr = 0.05;
    p->gf_iter = 3;
    p->gf_num_lpr = 10;
    p->gf_th_seeds_height = 0.5;
    p->map_voxel_size = 0.2;   // erasor.h:61
    p->version = 3;            // OMU.cpp:81
    p->query_voxel_size = 0.05;  // OMU.cpp:66
    p->removal_interval = 2;     // OMU.cpp:69
    p->voi_max_range = 0.0;
    p->is_large_scale = 0;   // OMU.cpp:75
    p->submap_size = 200.0;  // OMU.cpp:76
    return ERASOR_OK;
}

static void worker_start(erasor_hip_handle *h);
static void worker_stop(erasor_hip_handle *h);
static bool create_sides(erasor_hip_handle *h, int prio) {
    // round 4: three query streams where the process has the hardware queues for them (main + 3 + copy = 5 streams; HIP's default is 4
    // queues): 233 k-point scans 0.291 -> 0.230 ms, 127 k-point scans the same either way; two otherwise
    if (const char *q = getenv("GPU_MAX_HW_QUEUES")) {
        if (atoi(q) >= 8) h->nqs = 3;
    }
    // (round 5: FOUR busy compute queues are all the hardware runs side by side -- with a fifth every kernel of every queue slows down
    // 3-8 x: main + 4 query streams 0.19 -> 0.26 ms per scan, main + early + 3: 0.34 --, so the steps that use the early stream keep their
    // chains to two of these, see enqueue_query_chain)
    if (const char *e = getenv("ERASOR_HIP_QSTREAMS")) h->nqs = std::max(1, std::min((int)erasor_hip_handle::NQS_MAX, atoi(e)));
    for (int k = 0; k < h->nqs; ++k)
        if (hipStreamCreateWithPriority(&h->qstream[k], hipStreamNonBlocking, prio) != hipSuccess) return false;
    for (int k = 0; k < NSIDE; ++k)
        if (hipEventCreateWithFlags(&h->q[k].ev_keys, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->q[k].ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->q[k].ev_h2d, hipEventDisableTiming) != hipSuccess)
            return false;
    // (round 5: every stream that has run something keeps a hardware queue, and with more than four of them in this process' reach -- main,
    // query streams, early stream; the framework's own on top -- every kernel of every queue slows down 3-8 x, busy or not.  So the early
    // stream IS the third query stream where there is one: a step uses it either for a chain or for its early passes, see ov_mode)
    // Round 6: a stream of its own for the early passes (fewer than three query streams) and the copy stream are created WHEN FIRST NEEDED
    // (early_stream / copy_stream).  The runtime deals a process's streams onto the four compute pipes in the order they are created
    // (tools/queue_probe.hip), and two busy queues of one pipe are time-sliced: a handle that creates four streams and keeps two of them busy
    // -- ERASOR_HIP_QSTREAMS=1: main + one query stream -- left the second handle of the process the SAME two pipes; with two streams per
    // handle, two sequences interleaved on one GPU have a pipe for each of their four queues: 7950-8010 scans/s against 5310 one after the
    // other (1.5 x; MEASUREMENTS R6), where rounds 3-5 measured 0.8-1.07 x.
    if (h->nqs >= 3) h->bstream = h->qstream[2];
    h->bstream_own = false;
    // (events between two streams of ONE device: no system-scope fence -- the cache write-back and invalidation it brings cost the kernels
    // around it tens of microseconds)
    const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
    for (hipEvent_t *e : {&h->ev_stats, &h->ev_srt4, &h->ev_asm, &h->ev_early, &h->ev_scan})
        if (hipEventCreateWithFlags(e, evf) != hipSuccess) return false;
    h->cstream = nullptr;
    return true;
}
// the early stream of overlapped steps (one of the query streams when there are three; else created on first use)
static bool early_stream(erasor_hip_handle *h) {
    if (h->bstream) return true;
    if (hipStreamCreateWithFlags(&h->bstream, hipStreamNonBlocking) != hipSuccess) {
        h->bstream = nullptr;
        return false;
    }
    h->bstream_own = true;
    return true;
}
// the copy stream (host scans announced ahead, read-backs of a collected step): created on first use
static hipStream_t copy_stream(erasor_hip_handle *h) {
    if (!h->cstream && hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking) != hipSuccess) h->cstream = nullptr;
    return h->cstream ? h->cstream : h->stream;
}

// A handle runs its chains on four streams, and HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4):
// with several handles in one process -- one per sequence, or the shim's updaters -- the two query streams of a handle can land on ONE
// queue, its two chains in flight then run one after the other and the step waits for them (measured, round 3: five sequences in one
// process 2777 scans/s with the default, 4180 with 8 or 16 queues; the C++ bench's third handle 0.22 or 0.65 ms per step depending on
// where its streams landed).  The variable is read when the HIP runtime starts, so it is the PROCESS's business: rounds 2-3 set it from a
// load-time constructor of this library, which changed the environment of whoever loaded it.  Now: a documented requirement
// (include/erasor_hip.h; bench.py and the offline driver export it themselves) and ONE warning when a process creates
// its second handle without it.
static std::atomic<int> g_handles_created{0};
static void warn_hw_queues_once() {
    static std::atomic<bool> warned{false};
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    if ((e && atoi(e) >= 8) || warned.exchange(true)) return;
    fprintf(stderr, "[erasor_hip] several handles in one process and GPU_MAX_HW_QUEUES is %s: their streams share %s hardware queues and "
                    "look-ahead chains of one handle may run one after the other (export GPU_MAX_HW_QUEUES=16 before the process starts HIP)\n",
            e ? e : "unset", e ? e : "the default 4");
}

int erasor_hip_create(const erasor_params *p, int device, erasor_hip_handle **out) {
    if (!p || !out) return ERASOR_E_INVALID;
    *out = nullptr;
    if (p->num_rings <= 0 || p->num_sectors <= 0 || (int64_t)p->num_rings * p->num_sectors > 65000) return ERASOR_E_INVALID;
    if (p->gf_iter < 1 || p->gf_iter > 64) return ERASOR_E_INVALID;
    if (p->version != 2 && p->version != 3) return ERASOR_E_UNSUPPORTED;  // OMU.cpp:273-275 "Other version is not implemented!"
    if (!(p->max_range > 0) || !(p->map_voxel_size > 0) || !(p->query_voxel_size > 0)) return ERASOR_E_INVALID;
    if (p->is_large_scale && !(p->submap_size > 0)) return ERASOR_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return ERASOR_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return ERASOR_E_NO_DEVICE;
    erasor_hip_handle *h = new erasor_hip_handle();
    h->P = *p;
    h->device = device;
    h->B = (uint32_t)(p->num_rings * p->num_sectors);
    fill_dp(h);
    memset(&h->st, 0, sizeof(h->st));
    memset(&h->ctr, 0, sizeof(h->ctr));
    memset(&h->last_res, 0, sizeof(h->last_res));
    // Every stream at the DEFAULT priority: priorities are strict between hardware queues that share a pipe, and with several handles in
    // one process a handle's low-priority query queues could land beside its always-busy main queue and starve (EXPERIMENTS, round 3).
    const int prio_lo = 0, prio_hi = 0;
    if (hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        !create_sides(h, prio_lo) ||
        hipHostMalloc((void **)&h->pin, sizeof(HostOut), hipHostMallocDefault) != hipSuccess) {
        delete h;
        return ERASOR_E_NO_DEVICE;
    }
    memset(h->pin, 0, sizeof(HostOut));
    h->cur = h->stream;
    if (alloc_bins(h)) {
        erasor_hip_destroy(h);
        return ERASOR_E_NO_DEVICE;
    }
    if (g_handles_created.fetch_add(1) >= 1) warn_hw_queues_once();
    worker_start(h);
    *out = h;
    return ERASOR_OK;
}

void erasor_hip_destroy(erasor_hip_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    worker_stop(h);
    for (int k = 0; k < erasor_hip_handle::NQS_MAX; ++k)
        if (h->qstream[k]) (void)hipStreamSynchronize(h->qstream[k]);
    if (h->cstream) (void)hipStreamSynchronize(h->cstream);
    if (h->bstream) (void)hipStreamSynchronize(h->bstream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    prof_collect(h, true);
    for (auto e : h->evt_pool) (void)hipEventDestroy(e);
    for (h->qi = 0; h->qi < NSIDE; ++h->qi) {  // every query side
        release(Q(h).d_nvox); release(Q(h).d_qctr); release(Q(h).qb_hist); release(Q(h).qb_tot); release(Q(h).hkey); release(Q(h).hval); release(Q(h).qoff); release(Q(h).ccnt); release(Q(h).cmin); release(Q(h).cmax); release(Q(h).scan); release(Q(h).cent); release(Q(h).query); release(Q(h).sq); release(Q(h).bb); release(Q(h).qk_a); release(Q(h).qk_b); release(Q(h).qv_a); release(Q(h).qv_b); release(Q(h).qposL); release(Q(h).qposR); release(Q(h).qflag); release(Q(h).qpl); release(Q(h).qtops); release(Q(h).run_begin); release(Q(h).ukeys); release(Q(h).qkey); release(Q(h).qhead); release(Q(h).wseg0); release(Q(h).wseg1); release(Q(h).wstate); release(Q(h).wtileL); release(Q(h).wtileR); release(Q(h).esq0); release(Q(h).esq1); release(Q(h).esq2); release(Q(h).essmall); release(Q(h).esqs); release(Q(h).qgrid);
    }
    h->qi = 0;
    release(h->Cbuf); release(h->F[0]); release(h->F[1]); release(h->Oxy); release(h->Ozi);
    release(h->vmask); release(h->hmask); release(h->cinfo); release(h->pvl); release(h->phl); release(h->topv); release(h->toph); release(h->topr); release(h->ometa);
    release(h->voi_ego); release(h->spts); release(h->rejected); release(h->voi_key); release(h->voi_src); release(h->ssrc);
    release(h->rejected_src); release(h->grank); release(h->glist); release(h->gflag);
    release(h->rk_a); release(h->rk_b); release(h->rv_a); release(h->rv_b); release(h->hist); release(h->hist_l); release(h->hist_t); release(h->hist2); release(h->hist2_l); release(h->hist2_t); release(h->dn); release(h->lab_slots); release(h->mb_hist); release(h->mb_tot); release(h->mg_curr); release(h->mg_map); release(h->mg_done); release(h->mg_tmp);
    release(h->moff); release(h->mcnt); release(h->rev_idx); release(h->rev_list); release(h->vox_off);
    release(h->out_off0); release(h->rev_before); release(h->st1b);
    release(h->out_offR); release(h->gres_off);
    for (int k = 0; k < 2; ++k) { release(h->late[k]); release(h->late_holes[k]); }
    release(h->lmask);
    release(h->alt.voi_ego); release(h->alt.voi_key); release(h->alt.voi_src); release(h->alt.moff); release(h->alt.d_st); release(h->alt.d_ctr);
    release(h->alt.spts); release(h->alt.ssrc); release(h->alt.rk_a); release(h->alt.mcnt); release(h->alt.mmin); release(h->alt.mmax); release(h->alt.st1b);
    release(h->alt.lab_slots);
    for (hipEvent_t e : {h->ev_stats, h->ev_srt4, h->ev_asm, h->ev_early, h->ev_scan})
        if (e) (void)hipEventDestroy(e);
    if (h->bstream && h->bstream_own) (void)hipStreamDestroy(h->bstream);
    release(h->nvox); release(h->ng); release(h->out_off); release(h->ground_off); release(h->rej_off); release(h->crej_off);
    release(h->mmin); release(h->mmax); release(h->plane_n); release(h->plane_d);
    release(h->st1); release(h->status); release(h->action);
    release(h->curr_rejected);
    release(h->gsK); release(h->gsV); release(h->gsL); release(h->gsR); release(h->gsK2); release(h->gsV2); release(h->gsH); release(h->gsC);
    release(h->vox_out); release(h->d_st); release(h->d_st_get); release(h->d_ctr);
    for (int k = 0; k < NSIDE; ++k) {
        if (h->q[k].ev_keys) (void)hipEventDestroy(h->q[k].ev_keys);
        if (h->q[k].ev_done) (void)hipEventDestroy(h->q[k].ev_done);
        if (h->q[k].ev_h2d) (void)hipEventDestroy(h->q[k].ev_h2d);
        if (h->q[k].stage) (void)hipHostFree(h->q[k].stage);
    }
    for (int k = 0; k < erasor_hip_handle::NQS_MAX; ++k)
        if (h->qstream[k]) (void)hipStreamDestroy(h->qstream[k]);
    if (h->cstream) (void)hipStreamDestroy(h->cstream);
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *erasor_hip_last_error(const erasor_hip_handle *h) { return h ? h->err.c_str() : "null handle"; }

static int set_map_common(erasor_hip_handle *h, const void *src, size_t n, bool src_is_device) {
    NOFLY(h);
    if (h) ++h->store_epoch;
    if (!h) return ERASOR_E_INVALID;
    if (!src && n) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    if (h->bstream) HIPC(h, hipStreamSynchronize(h->bstream));  // (passes launched ahead of a step that will not come)
    h->ov.valid = false;
    if (n > 0x7FFFFFF0ull) return ERASOR_E_INVALID;
    int rc = alloc_map(h, (uint32_t)n);
    if (rc) return rc;
    // stage the AoS cloud in voi_ego (capV >= n), then split into the outskirts layout at the END of the O buffer
    if (n) {
        HIPC(h, hipMemcpyAsync(h->voi_ego.p, src, n * sizeof(float4), src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    }
    h->nF = 0;
    h->nFv = 0;
    h->n_late_F = 0;
    h->curF = 0;
    h->o_begin = h->capO - (uint32_t)n;
    h->o_valid = n;
    h->nC = 0;
    h->submap_not_initialized = true;
    if (n) LAUNCH(h, "set_map", k_store_outskirts, cdiv(n, 256), 256, (const float4 *)h->voi_ego.p, (uint32_t)n, h->Oxy.p, h->Ozi.p, h->o_begin);
    rc = ometa_reset(h);
    if (rc) return rc;
    memset(&h->st, 0, sizeof(h->st));
    rc = push_state(h);
    if (rc) return rc;
    // initial label counters of the outskirts (parse_dynamic_obj, utils.cpp:57-78)
    if (n) {
        DevState *ds = h->d_st.p;
        LAUNCH(h, "set_map", k_count_labels4, std::min<uint32_t>(cdiv(n, 256), 2048), 256, (const float4 *)h->voi_ego.p, (uint32_t)n,
               (const uint32_t *)nullptr, &ds->O_static, &ds->O_dynamic);
    }
    HIPC(h, hipMemcpyAsync(&h->st, h->d_st.p, sizeof(DevState), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    h->have_map = true;
    h->poisoned = false;
    h->last_n_voi = 0;  // (also the grid hint of the map bucketing: unknown for a new map)
    h->have_step = false;
    return ERASOR_OK;
}
int erasor_hip_set_map(erasor_hip_handle *h, const float *xyzi, size_t n) { return set_map_common(h, xyzi, n, false); }
int erasor_hip_set_map_device(erasor_hip_handle *h, const void *d_xyzi, size_t n) { return set_map_common(h, d_xyzi, n, true); }

// exact std::sort of (qk_a, qv_a)[0..n) -> (qk_b, qv_b): global levels (one workgroup per big segment, one partition per
// launch), then every remaining segment is completed inside LDS by one workgroup.
// (levels beyond lg(n / WIDE_MIN): 4 measured best -- fewer leave k_esort_mid / k_esort_final more than the levels save, EXPERIMENTS r05-5)
#define ESORT_WIDE_SLACK 4
static void run_exact_sort(erasor_hip_handle *h, uint32_t n, bool queues_open = false) {
    Counters *dc = Q(h).d_qctr.p;
    if (!queues_open)  // (the scan's voxelisation opens them in its key kernel: k_voxel_keys_es)
        LAUNCH(h, "q_esort", k_esort_init, 1, 1, Q(h).qk_a.p, Q(h).qv_a.p, Q(h).esq0.p, Q(h).essmall.p, Q(h).esqs.p, Q(h).wseg0.p, Q(h).wstate.p, n);
    int nlev = 0;
    if (n >= WIDE_MIN && (n - 1 + WTILE - 1) / WTILE <= WTILES_MAX) {
        // wide levels: segments >= WIDE_MIN keys, many workgroups each.  A segment halves (roughly) per level, so
        // lg(n / WIDE_MIN) + slack levels empty the wide list; whatever is left is routed to the level queue.
        // ERASOR_HIP_SORT_LEVEL_CAP (test hook): cut the level budget short so that the final kernel meets long segments
        static const int level_cap = getenv("ERASOR_HIP_SORT_LEVEL_CAP") ? atoi(getenv("ERASOR_HIP_SORT_LEVEL_CAP")) : 1 << 20;
        const int wl = std::min(std::min(esort::lg2_floor(n / WIDE_MIN) + ESORT_WIDE_SLACK, 16), level_cap);
        for (int l = 0; l < wl; ++l) {
            const int cur = l & 1;
            // two launches per level (round 2: three -- the children's routing and median move had a kernel of their own)
            LAUNCH(h, "q_esort_wide", k_esort_wide_mark, 256, 256, (const uint32_t *)Q(h).qk_a.p, (const uint32_t *)Q(h).qv_a.p, Q(h).qposL.p, Q(h).qposR.p,
                   cur ? Q(h).wseg1.p : Q(h).wseg0.p, Q(h).wstate.p, cur, Q(h).wtileL.p, Q(h).wtileR.p);
            LAUNCH(h, "q_esort_wide", k_esort_wide_swap, 256, 256, Q(h).qk_a.p, Q(h).qv_a.p, (const uint32_t *)Q(h).qposL.p, (const uint32_t *)Q(h).qposR.p,
                   cur ? Q(h).wseg1.p : Q(h).wseg0.p, cur ? Q(h).wseg0.p : Q(h).wseg1.p, Q(h).wstate.p, cur, (const uint32_t *)Q(h).wtileL.p,
                   (const uint32_t *)Q(h).wtileR.p, Q(h).esq0.p, Q(h).essmall.p, Q(h).esqs.p, 65536u, l == wl - 1 ? 1 : 0, dc);
        }
    }
    bool mid_done = false;
    if (n >= WIDE_MIN && (n - 1 + WTILE - 1) / WTILE <= WTILES_MAX && !getenv("ERASOR_HIP_SORT_LEVEL_CAP")) {
        // whatever is still longer than the finisher's LDS capacity after the wide levels (a few unbalanced subtrees):
        // one workgroup per segment cuts it down, all of its partitions in this one launch
        LAUNCH(h, "q_esort", k_esort_mid, 256, 1024, Q(h).qk_a.p, Q(h).qv_a.p, Q(h).qposL.p, Q(h).qposR.p, (const esort::Seg *)Q(h).esq0.p, Q(h).essmall.p,
               Q(h).esqs.p, 0, 65536u, dc);
        mid_done = true;
    }
    if (n > ES_LMAX && !mid_done) {
        // level queue: one workgroup per segment, one partition per launch.  Segments that are still big after `nlev`
        // levels are finished (slowly, in global memory) by the final kernel, so any nlev is correct.
        static const int level_cap2 = getenv("ERASOR_HIP_SORT_LEVEL_CAP") ? atoi(getenv("ERASOR_HIP_SORT_LEVEL_CAP")) : 1 << 20;
        nlev = std::min(std::min(2 * esort::lg2_floor(n), n >= WIDE_MIN ? (n > (1u << 21) ? 12 : 2) : 12), level_cap2);
        for (int l = 0; l < nlev; ++l)
            LAUNCH(h, "q_esort", k_esort_level, 48, 1024, Q(h).qk_a.p, Q(h).qv_a.p, Q(h).qposL.p, Q(h).qposR.p, Q(h).esq0.p, Q(h).esq1.p, Q(h).esq2.p,
                   Q(h).essmall.p, Q(h).esqs.p, l, 65536u, dc);
    }
    const int bigcur = mid_done ? 1 : nlev % 3;  // after k_esort_mid queue 0 is spent: hand the (empty) queue 1 to the final kernel
    esort::Seg *qs3[3] = {Q(h).esq0.p, Q(h).esq1.p, Q(h).esq2.p};
    // a scan leaves ~120 segments for the finisher: 128 workgroups of 1024 threads take them in one round and leave the compute units
    // to the chains running beside this one (2048 measured 1.5 % slower per scan, gpurun_out/r03o); a whole map keeps the wide grid
    const uint32_t final_grid = n <= (1u << 20) ? 128u : 2048u;
    LAUNCH(h, "q_esort_final", k_esort_final, final_grid, 1024, Q(h).qk_a.p, Q(h).qv_a.p, Q(h).qposL.p, Q(h).qposR.p, Q(h).qhead.p, Q(h).qk_b.p, Q(h).qv_b.p,
           (const esort::Seg *)Q(h).essmall.p, (const esort::Seg *)qs3[bigcur], Q(h).esqs.p, bigcur, dc, h->dbg_stamps.p);
}

// ---- exact voxelisation of the cloud in Q(h).scan[0..n): fills run_begin / cent / ukeys, d_st->q_nvox -------------------
static int voxelize_query_part1(erasor_hip_handle *h, uint32_t n, float leaf, const std::function<void()> &after_keys) {
    Counters *dc = Q(h).d_qctr.p;
    // (the bounding box was reset by k_query_begin)
    if (n) LAUNCH(h, "q_bbox", k_bbox, bbox_grid(n), 256, Q(h).scan_in, n, Q(h).bb.p);
    {
        EsInit es;
        es.q0 = Q(h).esq0.p;
        es.smallq = Q(h).essmall.p;
        es.qs = Q(h).esqs.p;
        es.w0 = Q(h).wseg0.p;
        es.ws = Q(h).wstate.p;
        LAUNCH(h, "q_keys", k_voxel_keys_es, std::max(1u, cdiv(n, 256)), 256, Q(h).scan_in, n, (const uint32_t *)Q(h).bb.p, leaf, Q(h).qk_a.p,
               Q(h).qv_a.p, Q(h).qgrid.p, dc, Q(h).hkey.p, 1u << Q(h).hbits, es);
    }
    after_keys();  // ctr->err (VoxelGrid overflow) is final from here on: the caller may fork work that depends on it
    // exact std::sort: a few global levels (one workgroup per big segment), then per-segment completion in LDS
    run_exact_sort(h, n, true);
    // runs
    const uint32_t ntile = std::max(1u, cdiv(n, 1024));
    if (ntile <= 1024) {  // (a scan: ~120 tiles) two launches: tile totals, then heads -> run_begin with the tile's offset summed in place
        LAUNCH(h, "q_runs", k_run_count, ntile, 256, (const uint32_t *)Q(h).qk_b.p, n, Q(h).qtops.p);
        LAUNCH(h, "q_runs", k_run_emit, ntile, 256, (const uint32_t *)Q(h).qk_b.p, n, (const uint32_t *)Q(h).qtops.p, ntile, Q(h).run_begin.p,
               Q(h).d_nvox.p);
        return 0;
    }
    if (n) LAUNCH(h, "q_runs", k_run_heads, cdiv(n, 256), 256, (const uint32_t *)Q(h).qk_b.p, n, Q(h).qflag.p);
    scan_u32(h, Q(h).qflag.p, Q(h).qpl.p, Q(h).qtops.p, n, n, nullptr, nullptr, "q_runs");
    LAUNCH(h, "q_runs", k_run_begin, cdiv((uint64_t)n + 1, 256), 256, (const uint32_t *)Q(h).qflag.p, (const uint32_t *)Q(h).qpl.p,
           (const uint32_t *)Q(h).qtops.p, n, Q(h).run_begin.p, Q(h).d_nvox.p);
    return 0;
}

enum { STEP_QUERY_PREVOXELIZED = 1, STEP_VOI_EVERYTHING = 2, STEP_RETRIED = 4 };

// Hash of EVERY record of a host scan (the 16 bytes a record contributes: x, y, z, intensity).  Round 3 sampled ~258 records: a
// buffer refilled in place that differed only in unsampled points was taken for the announced scan and the step ran on the stale copy
// (ADVICE r02 / VERDICT r03).  Four independent multiply-xor lanes, one per field; ~0.1 ms for a 2 MB scan on one host core -- which is
// why the drop-in path hands over a TICKET instead (erasor_hip_step_ticket: nothing is guessed, nothing is read twice).
struct RowHash {
    uint64_t a = 0x9E3779B97F4A7C15ull, b = 0xC2B2AE3D27D4EB4Full, c = 0x165667B19E3779F9ull, d = 0x27D4EB2F165667C5ull;
    inline void add(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
        a = (a ^ x) * 0x100000001B3ull;
        b = (b ^ y) * 0x100000001B3ull;
        c = (c ^ z) * 0x100000001B3ull;
        d = (d ^ w) * 0x100000001B3ull;
    }
    inline uint64_t done(size_t n) const {
        uint64_t h = a ^ (b << 1 | b >> 63) ^ (c << 2 | c >> 62) ^ (d << 3 | d >> 61) ^ (uint64_t)n * 0x9FB21C651E98DF25ull;
        h ^= h >> 32;
        h *= 0xD6E8FEB86659FD93ull;
        h ^= h >> 32;
        return h;
    }
};
static uint64_t scan_fingerprint(const void *host_rows, size_t n, RowFmt fmt) {
    RowHash rh;
    const unsigned char *p = static_cast<const unsigned char *>(host_rows);
    for (size_t i = 0; i < n; ++i, p += fmt.stride) {
        uint32_t w[4];
        memcpy(w, p, 12);
        memcpy(w + 3, p + fmt.ioff, 4);
        rh.add(w[0], w[1], w[2], w[3]);
    }
    return rh.done(n);
}

// the hash of the copy a side has staged (float4 rows in its pinned buffer), taken when a step has to recognise its scan by content
static uint64_t staged_fingerprint(QSide &q, size_t n) {
    if (!q.fp_valid) {
        q.fp = scan_fingerprint(q.stage, n, RowFmt());
        q.fp_valid = true;
    }
    return q.fp;
}

// A HOST scan on its way into query side Q(h): REPACKED into the side's pinned staging buffer NOW -- one pass that also hashes every
// record (the caller's buffer is free again when this returns) --, from there asynchronously into q.scan on `stream`.  The side's
// previous chain must be through with q.scan and the staging buffer (the caller has waited for q.ev_done).
static int stage_host_scan(erasor_hip_handle *h, const void *scan_src, uint32_t ns, hipStream_t stream, RowFmt fmt, uint64_t *fp_out) {
    QSide &q = Q(h);
    if (q.h2d_pending) HIPC(h, hipEventSynchronize(q.ev_h2d));  // (an announcement that was dropped: its copy may still read the staging buffer)
    if (scan_src != (const void *)q.stage && q.stage_cap < ns) {
        if (q.stage) (void)hipHostFree(q.stage);
        q.stage = nullptr;
        q.stage_cap = (size_t)ns + ns / 4 + 1024;
        HIPC(h, hipHostMalloc((void **)&q.stage, q.stage_cap * sizeof(float4), hipHostMallocDefault));
        for (int k = 0; k < NSIDE; ++k) {  // (round 6: the sides that have never staged anything get their pinned buffers now, see alloc_scan)
            QSide &o = h->q[k];
            if (o.stage || &o == &q) continue;
            if (hipHostMalloc((void **)&o.stage, q.stage_cap * sizeof(float4), hipHostMallocDefault) == hipSuccess) o.stage_cap = q.stage_cap;
            else o.stage = nullptr;
        }
    }
    const unsigned char *p = static_cast<const unsigned char *>(scan_src);
    uint32_t *dst = reinterpret_cast<uint32_t *>(q.stage);
    if (!fp_out) {
        // Round 6 (second half): the repack alone.  The hash is a chain of dependent multiplies, ~0.1 ms per scan on the caller's thread, and
        // it is only ever compared when a step comes WITHOUT a ticket and has to recognise its announcement by content: it is taken then,
        // from this staged copy (staged_fingerprint), not here -- the drop-in callback path steps by ticket and never pays for it.
        if (scan_src != (const void *)q.stage) {
            if (fmt.stride == 16 && fmt.ioff == 12) {
                memcpy(dst, p, (size_t)ns * 16);
            } else {
                for (uint32_t i = 0; i < ns; ++i, p += fmt.stride) {
                    uint32_t w[4];
                    memcpy(w, p, 12);
                    memcpy(w + 3, p + fmt.ioff, 4);
                    memcpy(dst + 4 * (size_t)i, w, 16);
                }
            }
        }
    } else {
        RowHash rh;
        if (scan_src == (const void *)q.stage) {  // (a step run again from its own staged copy: already in place)
            for (uint32_t i = 0; i < ns; ++i) rh.add(dst[4 * i], dst[4 * i + 1], dst[4 * i + 2], dst[4 * i + 3]);
        } else {
            for (uint32_t i = 0; i < ns; ++i, p += fmt.stride) {
                uint32_t w[4];
                memcpy(w, p, 12);
                memcpy(w + 3, p + fmt.ioff, 4);
                memcpy(dst + 4 * (size_t)i, w, 16);
                rh.add(w[0], w[1], w[2], w[3]);
            }
        }
        *fp_out = rh.done(ns);
    }
    HIPC(h, hipMemcpyAsync(q.scan.p, q.stage, (size_t)ns * sizeof(float4), hipMemcpyHostToDevice, stream));
    return ERASOR_OK;
}

// ---- the handle's worker thread (see erasor_hip_handle::Worker) ----
static void worker_main(erasor_hip_handle *h) {
    (void)hipSetDevice(h->device);
    auto *w = h->worker;
    for (;;) {
        std::function<int()> job;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->stop || !w->jobs.empty(); });
            if (w->jobs.empty()) return;  // (stop, nothing left)
            job = std::move(w->jobs.front());
            w->jobs.pop_front();
        }
        (void)job();  // (the job stores its error before it releases its sides, see dispatch_chain)
    }
}
static void worker_start(erasor_hip_handle *h) {
#ifndef ERASOR_NO_WORKER_THREAD  // (a build option: every launch from the caller's thread)
    h->worker = new erasor_hip_handle::Worker();
    for (int k = 0; k < NSIDE; ++k) h->worker->busy[k].store(0);
    h->worker->th = std::thread(worker_main, h);
#else
    (void)h;
#endif
}
static void worker_stop(erasor_hip_handle *h) {
    if (!h->worker) return;
    {
        std::lock_guard<std::mutex> lk(h->worker->mu);
        h->worker->stop = true;
    }
    h->worker->cv.notify_all();
    if (h->worker->th.joinable()) h->worker->th.join();
    delete h->worker;
    h->worker = nullptr;
}
// ERASOR_HIP_OVERLAP: 1 = consecutive steps overlap wherever they can, 0 = never, unset = the handle decides: by measurement (OvAuto), and
// only for a caller that announces DEEP -- at least `batch_lead` nodes beyond the step's own.  Overlapped steps leave the query chains two
// streams, which is enough only when the chains share their launches, and chains are held back for that only with `batch_lead` of them
// in their queues in front (flush_announced): a callback that announces one or two nodes ahead gets plain steps and three query
// streams (0.19 against 0.20-0.21 ms per scan with two ahead; the drop-in path's callback_next_announced_ms 0.42 against 0.50).
static int overlap_env() {
    static const int v = getenv("ERASOR_HIP_OVERLAP") && getenv("ERASOR_HIP_OVERLAP")[0] ? atoi(getenv("ERASOR_HIP_OVERLAP")) : -1;
    return v;
}
static bool overlap_wanted(const erasor_hip_handle *h, int announced_beyond) {
    // (a handle that was given ONE query stream is meant to keep to two busy queues -- several handles on one GPU: it does not try the
    // overlap, whose early passes are a third queue)
    return overlap_env() >= 0 ? overlap_env() != 0 : (h->ova.mode != 0 && announced_beyond >= h->batch_lead && h->nqs >= 2);
}
static int flush_held(erasor_hip_handle *h);
// the launches (and event records) of side `side`'s chain have all been made; side < 0: of every side
static int chain_wait(erasor_hip_handle *h, int side) {
    bool held = false;
    for (int k = 0; k < NSIDE; ++k) held = held || ((side < 0 || k == side) && h->q[k].held);
    if (held) {  // (a chain that was held back for a shared set of launches: whoever needs its events sends the set off)
        const int rc = flush_held(h);
        if (rc) return rc;
    }
    if (!h->worker) return ERASOR_OK;
    for (int k = 0; k < NSIDE; ++k) {
        if (side >= 0 && k != side) continue;
        while (h->worker->busy[k].load(std::memory_order_acquire) > 0) std::this_thread::yield();
    }
    const int e = h->worker->err.exchange(0);
    if (e) {
        h->err = "a query chain could not be enqueued (worker thread)";
        return e;
    }
    return ERASOR_OK;
}

// a query side no announced / in-flight chain owns; the side of the last finished step only if nothing else is free
// (its query-derived outputs are then gone)
static int pick_side(const erasor_hip_handle *h) {
    int fallback = -1;
    for (int k = 0; k < NSIDE; ++k) {
        bool busy = h->ann.valid && h->ann.side == k;
        for (int j = 0; j < h->npend; ++j) busy = busy || h->pend[j] == k;
        if (busy) continue;
        if (k != h->qi) return k;
        fallback = k;
    }
    return fallback;
}

// restores the handle's active query side / target stream when a helper that switched them returns (also on errors)
struct SideGuard {
    erasor_hip_handle *h;
    int qi;
    hipStream_t cur;
    explicit SideGuard(erasor_hip_handle *hh) : h(hh), qi(hh->qi), cur(hh->cur) {}
    ~SideGuard() {
        h->qi = qi;
        h->cur = cur;
    }
};

// The QUERY CHAIN of one scan, enqueued on the stream of query side `side`: voxelize_preserving_labels (OMU.cpp:238),
// lidar->body + R-POD key (OMU.cpp:240; erasor.cpp:100-115), bucketing and per-bin statistics of the query.  It depends
// on the scan and the lidar->body transform only -- not on the map -- which is what lets erasor_hip_prefetch_scan run
// it for scan k+1 while step k is still in its map-side stages.
// A host scan is copied to the device when it is announced (erasor_hip_prefetch_scan).  The step that claims the prefetched
// chain recognises "the same scan" by (pointer, size, T_lidar2body) -- and by this fingerprint of the contents, so that a
// buffer that was refilled (or freed and re-used at the same address) in between is not mistaken for the announced one.
// FNV-1a over at most 256 evenly spaced points plus the first and the last: microseconds, not a checksum of 2 MB.

// VoxelGrid pass-through (see k_dup_label_passthrough): out[0..ns) = T * (point with the label of its first exact duplicate),
// on the current stream.  Scratch: the current query side's sort buffers.
static int enqueue_passthrough(erasor_hip_handle *h, const float4 *d_src, uint32_t ns, const float T[16], float4 *out, uint32_t *qkey) {
    if (!ns) return ERASOR_OK;
    int bits = 8;
    while ((1u << bits) < 2u * ns && bits < 24) ++bits;
    QSide &q = Q(h);
    LAUNCH(h, "q_passthrough", k_xyz_hash_keys, cdiv(ns, 256), 256, d_src, ns, bits, q.ukeys.p);
    const uint32_t *sk = nullptr, *sp = nullptr;
    const int rc = radix_sort(h, q.ukeys.p, ns, nullptr, bits, q.qk_a.p, q.qposL.p, q.qv_a.p, q.qposR.p, &sk, &sp, "q_passthrough");
    if (rc) return rc;
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    LAUNCH(h, "q_passthrough", k_dup_label_passthrough, cdiv(ns, 256), 256, d_src, ns, sk, sp, bits, to_xf(T ? T : I), T ? 1 : 0, h->dp, q.d_qctr.p, out,
           qkey);
    return ERASOR_OK;
}

// ---- the launches of ONE scan's chain on `qstream` (the caller's thread or the handle's worker: TlCtx) ----
struct ChainJob {
    erasor_hip_handle *h;
    int side;
    hipStream_t qstream;
    uint32_t ns;
    bool prevox, passthrough, to_worker;
    float Tl[16];
};
static int chain_launches(const ChainJob &j) {
    erasor_hip_handle *h = j.h;
    QSide &q = h->q[j.side];
    const DP P = h->dp;
    const uint32_t B = h->B, ns = j.ns, nq = j.ns;  // nq: upper bound; the actual count lives in d_nvox
    const int bits = key_bits(B + 1);
    const bool prevox = j.prevox, passthrough = j.passthrough;
    hipStream_t qstream = j.qstream;
    Counters *qc = q.d_qctr.p;
    const uint32_t *nq_dev = q.d_nvox.p;
    const float *T_l2b = j.Tl;
    int rc = 0;
    // (LAUNCH() and Q() look at the thread's context first: the chain's own stream and side, on the worker thread and on the caller's
    // alike; on the caller's thread the handle's own fields say the same -- the tests' CPU stand-in evaluates a launch's arguments on
    // its wavefront threads, which have no context of their own)
    TlCtx keep_tl = g_tl;
    g_tl.h = h;
    g_tl.cur = qstream;
    g_tl.qi = j.side;
    struct HandleCtx {
        erasor_hip_handle *h;
        int qi;
        hipStream_t cur;
        ~HandleCtx() {
            if (h) {
                h->qi = qi;
                h->cur = cur;
            }
        }
    } handle_ctx{j.to_worker ? nullptr : h, h->qi, h->cur};
    if (!j.to_worker) {
        h->qi = j.side;
        h->cur = qstream;
    }
    struct TlRestore {
        TlCtx k;
        ~TlRestore() { g_tl = k; }
    } tl_restore{keep_tl};
    LAUNCH(h, "q_begin", k_query_begin, 1, 256, qc, q.bb.p, B + 1 <= QB_NB_MAX ? q.qb_tot.p : (uint32_t *)nullptr, B + 1 <= QB_NB_MAX ? B + 1 : 0u,
           q.d_nvox.p, (prevox || passthrough) ? ns : 0u);
    // ---- part 1: bounding box, voxel keys, exact std::sort, runs ----
    if (passthrough && !prevox) {
        // PCL's VoxelGrid refuses this cloud (index overflow) and hands it back unchanged (utils.cpp:88-91): the chain verifies
        // that on the device (err 4 if not) and produces the un-voxelised query with the label search's answers
        if (ns) LAUNCH(h, "q_bbox", k_bbox, bbox_grid(ns), 256, q.scan_in, ns, q.bb.p);
        LAUNCH(h, "q_passthrough", k_passthrough_check, 1, 64, (const uint32_t *)q.bb.p, ns, P.leaf_query, q.qgrid.p, qc);
        (void)hipEventRecord(q.ev_keys, qstream);
        rc = enqueue_passthrough(h, q.scan_in, ns, T_l2b, q.query.p, q.qkey.p);
        if (rc) return rc;
    } else if (!prevox) {
        voxelize_query_part1(h, ns, P.leaf_query, [&] { (void)hipEventRecord(q.ev_keys, qstream); });
    } else
        (void)hipEventRecord(q.ev_keys, qstream);
    // ---- part 2: centroids, label NN, lidar->body, R-POD key ----
    const uint32_t *sq_keys = nullptr, *sq_perm = nullptr;
    if (prevox) {
        if (nq) LAUNCH(h, "q_direct", k_query_direct, cdiv(nq, 256), 256, q.scan_in, nq, P, qc, q.query.p, q.qkey.p);
    } else if (passthrough) {
        // (query and keys are already there)
    } else if (nq) {
        LAUNCH(h, "q_centroids", k_centroids, cdiv((uint64_t)nq * 8, 256), 256, q.scan_in, (const uint32_t *)q.qk_b.p, (const uint32_t *)q.qv_b.p,
               (const uint32_t *)q.run_begin.p, nq_dev, q.cent.p, q.ukeys.p, q.hkey.p, q.hval.p, q.hbits);
        LAUNCH(h, "q_nn", k_query_nn, cdiv((uint64_t)nq * NN_SUB, 256), 256, q.scan_in, (const uint32_t *)q.qv_b.p, (const uint32_t *)q.run_begin.p,
               (const uint32_t *)q.ukeys.p, (const float4 *)q.cent.p, nq_dev, (const VoxGrid *)q.qgrid.p, to_xf(T_l2b), P, qc, q.query.p, q.qkey.p,
               (const uint32_t *)q.hkey.p, (const uint32_t *)q.hval.p, q.hbits);
    }
    if (B + 1 <= QB_NB_MAX) {  // one-digit stable counting sort with the gather and the bucket offsets folded in
        const uint32_t ntile_ub = std::max(1u, cdiv(nq, QB_TILE));
        LAUNCH(h, "q_bucket", k_qb_hist, ntile_ub, 1024, (const uint32_t *)q.qkey.p, nq, nq_dev, B + 1, q.qb_hist.p, q.qb_tot.p);
        LAUNCH(h, "q_bucket", k_qb_scan, 1, 1024, (const uint32_t *)q.qb_tot.p, B + 1, q.qoff.p);
        LAUNCH(h, "q_bucket", k_qb_scatter, ntile_ub, 1024, (const uint32_t *)q.qkey.p, (const float4 *)q.query.p, nq, nq_dev, B + 1, bits,
               (const uint32_t *)q.qb_hist.p, (const uint32_t *)q.qoff.p, q.sq.p);
    } else {  // very fine R-POD grids: LSD radix passes
        radix_sort(h, q.qkey.p, nq, nq_dev, bits, q.qk_a.p, q.qposL.p, q.qv_a.p, q.qposR.p, &sq_keys, &sq_perm, "q_bucket");
        if (nq) LAUNCH(h, "q_bucket", k_gather, cdiv(nq, 256), 256, (const float4 *)q.query.p, (const uint32_t *)nullptr, sq_perm, nq, nq_dev, q.sq.p,
                       (uint32_t *)nullptr);
        LAUNCH(h, "q_bucket", k_bin_offsets, cdiv((uint64_t)std::max(nq, B + 2) + 1, 256), 256, sq_keys, nq, nq_dev, B + 1, q.qoff.p);
    }
    LAUNCH(h, "bin_stats", k_bin_stats, cdiv((uint64_t)B * 64, 256), 256, (const float4 *)q.sq.p, (const uint32_t *)q.qoff.p, B, q.ccnt.p, q.cmin.p,
           q.cmax.p);
    (void)hipEventRecord(q.ev_done, qstream);
    return rc;
}

// ---- round 6: the chains of `nb` announced scans as ONE set of launches (see QBatch in kernels.hip.h): blockIdx.y is the scan ----
static QSideDev side_dev(const QSide &q) {
    QSideDev d;
    memset(&d, 0, sizeof(d));
    d.scan_in = q.scan_in;
    d.cent = q.cent.p; d.query = q.query.p; d.sq = q.sq.p;
    d.bb = q.bb.p; d.qk_a = q.qk_a.p; d.qk_b = q.qk_b.p; d.qv_a = q.qv_a.p; d.qv_b = q.qv_b.p; d.qposL = q.qposL.p; d.qposR = q.qposR.p;
    d.qtops = q.qtops.p; d.run_begin = q.run_begin.p; d.ukeys = q.ukeys.p; d.qkey = q.qkey.p; d.qhead = q.qhead.p; d.wtileL = q.wtileL.p;
    d.wtileR = q.wtileR.p; d.hkey = q.hkey.p; d.hval = q.hval.p;
    d.qb_tot = q.qb_tot.p; d.qb_hist = q.qb_hist.p; d.qoff = q.qoff.p; d.ccnt = q.ccnt.p; d.d_nvox = q.d_nvox.p;
    d.cmin = q.cmin.p; d.cmax = q.cmax.p;
    d.esq0 = q.esq0.p; d.esq1 = q.esq1.p; d.essmall = q.essmall.p; d.esqs = q.esqs.p;
    d.wseg0 = q.wseg0.p; d.wseg1 = q.wseg1.p; d.wstate = q.wstate.p;
    d.qgrid = q.qgrid.p; d.d_qctr = q.d_qctr.p;
    d.Tl = to_xf(q.Tl);
    d.n = q.ns;
    d.hbits = q.hbits;
    return d;
}
struct BatchJob {
    erasor_hip_handle *h;
    int sides[QBATCH_MAX];
    int nb;
    hipStream_t qstream;
    QBatch qb;
    bool to_worker;
};
static int batch_launches(const BatchJob &j) {
    erasor_hip_handle *h = j.h;
    const DP P = h->dp;
    const uint32_t B = h->B;
    const int bits = key_bits(B + 1);
    const uint32_t nb = (uint32_t)j.nb;
    hipStream_t qstream = j.qstream;
    uint32_t nmax = 0;
    for (int k = 0; k < j.nb; ++k) nmax = std::max(nmax, j.qb.s[k].n);
    TlCtx keep_tl = g_tl;
    g_tl.h = h;
    g_tl.cur = qstream;
    g_tl.qi = j.sides[0];
    struct TlRestore {
        TlCtx k;
        ~TlRestore() { g_tl = k; }
    } tl_restore{keep_tl};
    const QBatch &qb = j.qb;
#define LAUNCH_B(name, kern, gx, block, ...) LAUNCH(h, name, kern, dim3((gx), nb), block, qb, ##__VA_ARGS__)
    LAUNCH_B("q_begin", k_query_begin_b, 1, 256, B + 1);
    LAUNCH_B("q_bbox", k_bbox_b, bbox_grid(nmax), 256);
    LAUNCH_B("q_keys", k_voxel_keys_es_b, std::max(1u, cdiv(nmax, 256)), 256, P.leaf_query);
    for (int k = 0; k < j.nb; ++k) (void)hipEventRecord(h->q[j.sides[k]].ev_keys, qstream);  // (the VoxelGrid overflow flags are final)
    const int wl = std::min(esort::lg2_floor(nmax / WIDE_MIN) + ESORT_WIDE_SLACK, 16);
    for (int l = 0; l < wl; ++l) {
        LAUNCH_B("q_esort_wide", k_esort_wide_mark_b, 256, 256, l & 1);
        LAUNCH_B("q_esort_wide", k_esort_wide_swap_b, 256, 256, l & 1, l == wl - 1 ? 1 : 0);
    }
    LAUNCH_B("q_esort", k_esort_mid_b, 256, 1024);
    LAUNCH_B("q_esort_final", k_esort_final_b, 128, 1024);
    const uint32_t ntile = std::max(1u, cdiv(nmax, 1024));
    LAUNCH_B("q_runs", k_run_count_b, ntile, 256);
    LAUNCH_B("q_runs", k_run_emit_b, ntile, 256);
    LAUNCH_B("q_centroids", k_centroids_b, cdiv((uint64_t)nmax * 8, 256), 256);
    LAUNCH_B("q_nn", k_query_nn_b, cdiv((uint64_t)nmax * NN_SUB, 256), 256, P);
    const uint32_t ntile_ub = std::max(1u, cdiv(nmax, QB_TILE));
    LAUNCH_B("q_bucket", k_qb_hist_b, ntile_ub, 1024, B + 1);
    LAUNCH_B("q_bucket", k_qb_scan_b, 1, 1024, B + 1);
    LAUNCH_B("q_bucket", k_qb_scatter_b, ntile_ub, 1024, B + 1, bits);
    LAUNCH_B("bin_stats", k_bin_stats_b, cdiv((uint64_t)B * 64, 256), 256, B);
#undef LAUNCH_B
    for (int k = 0; k < j.nb; ++k) (void)hipEventRecord(h->q[j.sides[k]].ev_done, qstream);
    return ERASOR_OK;
}

// a job for the handle's worker thread (round 5: with the steps overlapped the caller's thread had become the bound), or run right here
static int dispatch_chain(erasor_hip_handle *h, bool to_worker, const int *sides, int nsides, std::function<int()> job) {
    if (!to_worker) return job();
    auto *w = h->worker;
    int sd[QBATCH_MAX];
    for (int k = 0; k < nsides; ++k) {
        sd[k] = sides[k];
        w->busy[sd[k]].fetch_add(1, std::memory_order_acq_rel);
    }
    {
        std::lock_guard<std::mutex> lk(w->mu);
        std::array<int, QBATCH_MAX> sda;
        for (int k = 0; k < QBATCH_MAX; ++k) sda[k] = k < nsides ? sd[k] : 0;
        w->jobs.emplace_back([job, w, sda, nsides]() -> int {
            const int rc_j = job();
            if (rc_j) w->err.store(rc_j);  // (before the sides are released: whoever waits for them sees the error)
            for (int k = 0; k < nsides; ++k) w->busy[sda[k]].fetch_sub(1, std::memory_order_acq_rel);
            return rc_j;
        });
    }
    w->cv.notify_one();
    return ERASOR_OK;
}
// the stream a chain (or a set of chains) goes to.  (The radix fallback for very fine R-POD grids and the pass-through chain's radix
// sort share their scratch bank between the sides: one stream for all of them.  While the early stream is in use the chains keep to
// two query streams: four busy compute queues are all that run side by side.)
static hipStream_t next_chain_stream(erasor_hip_handle *h, bool shared_scratch) {
    const unsigned nqs_now = (unsigned)(h->ov_mode ? std::min(h->nqs, 2) : h->nqs);
    return shared_scratch ? h->qstream[0] : h->qstream[h->n_chain++ % nqs_now];
}
// what a chain's stream has to wait for before the side's buffers are written: the staged copy of a host scan, the side's previous chain
static void chain_stream_waits(erasor_hip_handle *h, QSide &q, hipStream_t qstream) {
    if (q.h2d_pending) {  // announced earlier (erasor_hip_prefetch_*): the copy runs on the copy stream
        (void)hipStreamWaitEvent(qstream, q.ev_h2d, 0);
        q.h2d_pending = false;
    }
    if (q.used) (void)hipStreamWaitEvent(qstream, q.ev_done, 0);  // (possibly a dropped chain, possibly on the other query stream)
}

// The chains that were set up and HELD BACK (QSide::held) go into a queue now: two or more as one set of launches, a single one as
// the plain chain.  Called when the set is full, when a held chain comes within `batch_lead` steps of its own step, and by whoever
// needs a held side's events (chain_wait).
static int flush_held(erasor_hip_handle *h) {
    int sides[NSIDE], n = 0;
    for (int j = 0; j < h->npend; ++j)
        if (h->q[h->pend[j]].held) sides[n++] = h->pend[j];
    for (int k = 0; k < NSIDE; ++k) h->q[k].held = false;  // (a held side that is no longer pending was dropped: nothing to launch)
    for (int i0 = 0; i0 < n;) {
        const int nb = std::min(n - i0, std::max(1, std::min(h->batch_n, (int)QBATCH_MAX)));
        hipStream_t qstream = next_chain_stream(h, false);
        const bool to_worker = h->worker != nullptr;
        if (nb == 1) {
            QSide &q = h->q[sides[i0]];
            chain_stream_waits(h, q, qstream);
            ChainJob cj{h, sides[i0], qstream, q.ns, false, false, to_worker, {0}};
            memcpy(cj.Tl, q.Tl, sizeof(cj.Tl));
            const int rc = dispatch_chain(h, to_worker, &sides[i0], 1, [cj]() { return chain_launches(cj); });
            if (rc) return rc;
        } else {
            BatchJob bj;
            bj.h = h;
            bj.nb = nb;
            bj.qstream = qstream;
            bj.to_worker = to_worker;
            memset(&bj.qb, 0, sizeof(bj.qb));
            for (int k = 0; k < nb; ++k) {
                QSide &q = h->q[sides[i0 + k]];
                chain_stream_waits(h, q, qstream);
                bj.sides[k] = sides[i0 + k];
                bj.qb.s[k] = side_dev(q);
            }
            const int rc = dispatch_chain(h, to_worker, bj.sides, nb, [bj]() { return batch_launches(bj); });
            if (rc) return rc;
            ++h->n_batches;
            h->n_batched_chains += (unsigned)nb;
        }
        i0 += nb;
    }
    return ERASOR_OK;
}

// staged: 0 = the scan comes from the caller now; 1 = announced (h->ann): a host scan lies in the side's staging copy already;
// 2 = the side's OWN announcement once more (its chain ran in the wrong VoxelGrid mode, step_collect): scan, ticket, hash and pose stay
// may_hold: the chain may be held back for a set of launches shared with the next announcements (flush_announced decides)
static int enqueue_query_chain(erasor_hip_handle *h, int side, const void *scan_src, uint32_t ns, bool src_is_device, const float T_l2b[16],
                               bool prevox, int staged = 0, bool passthrough = false, RowFmt fmt = RowFmt(), bool may_hold = false) {
    {
        const int rc_w = chain_wait(h, side);  // (the side's previous chain: its events are waited for below)
        if (rc_w) return rc_w;
    }
    SideGuard guard(h);
    h->qi = side;
    int rc = alloc_scan(h, std::max(ns, 1u));
    if (rc) return rc;
    const uint32_t B = h->B, nq = ns;
    if (B + 1 <= QB_NB_MAX) {
        if (ensure(h, Q(h).qb_hist, (size_t)(B + 1) * std::max(1u, cdiv(nq, QB_TILE)) + 8)) return ERASOR_E_NO_DEVICE;
    } else {
        const uint32_t nb_q = 256u * std::max(1u, cdiv(nq, RTILE));
        if (ensure(h, h->hist, nb_q) || ensure(h, h->hist_l, nb_q) || ensure(h, h->hist_t, cdiv(nb_q, 1024) + 2)) return ERASOR_E_NO_DEVICE;
    }
    QSide &q = Q(h);
    // the common chain (voxelising, counting-sort bucketing, nothing profiled launch by launch) may go to the worker thread and may share
    // its launches with other scans' chains
    const bool common = !passthrough && !prevox && B + 1 <= QB_NB_MAX && h->prof != 1 && !g_debug_sync;
    const bool hold = may_hold && common && staged == 1 && h->batch_n >= 2 && ns >= WIDE_MIN && ns <= (1u << 20);
    hipStream_t qstream = hold ? nullptr : next_chain_stream(h, !(B + 1 <= QB_NB_MAX && !passthrough));
    // a device-resident scan is read in place (it must stay valid until the step that consumes it has returned);
    // a host scan is copied now (blocking copy: this side has no work in flight)
    if (ns && !src_is_device && !staged) {
        // the side may still be executing a chain that was dropped (a prefetch that no step claimed): its kernels read
        // q.scan, and a scan that changes between the bounding-box pass and the voxel keys sends them astray
        if (q.used) HIPC(h, hipEventSynchronize(q.ev_done));
        rc = stage_host_scan(h, scan_src, ns, qstream, fmt, nullptr);  // (on the chain's own stream: ordered before its first kernel)
        if (rc) return rc;
        q.h2d_pending = false;
        q.fp = 0ull;
        q.fp_valid = false;  // (staged_fingerprint, if anybody ever asks)
    }
    q.scan_in = src_is_device && ns ? (const float4 *)scan_src : (const float4 *)q.scan.p;
    q.ns = ns;
    if (q.Tl != T_l2b) memcpy(q.Tl, T_l2b, sizeof(q.Tl));
    if (hold) {
        q.held = true;  // (flush_held: stream, waits and launches)
    } else {
        chain_stream_waits(h, q, qstream);
        // (round 6: a chain the step needs AT ONCE -- its own scan, not announced ahead -- is launched right here: handing it to the worker
        // only to wait for the worker's event records puts a thread wake-up in front of the step)
        const bool to_worker = h->worker && common && staged != 0;
        ChainJob cj{h, side, qstream, ns, prevox, passthrough, to_worker, {0}};
        memcpy(cj.Tl, T_l2b, sizeof(cj.Tl));
        if (to_worker) rc = dispatch_chain(h, true, &side, 1, [cj]() { return chain_launches(cj); });
        else rc = chain_launches(cj);
        if (rc) return rc;
    }
    q.used = true;
    q.src = scan_src;
    q.src_n = ns;
    q.src_dev = src_is_device;
    if (staged != 2) {
        q.pose_valid = false;  // (flush_announced adds the pose of an announced node)
        q.to_valid = false;
        if (!src_is_device && staged) {  // (the announcement's hash, if it has been taken)
            q.fp = h->ann.fp;
            q.fp_valid = h->ann.fp_valid;
        }
        q.ticket = staged ? h->ann.ticket : 0ull;
    }
    q.fmt = fmt;
    return ERASOR_OK;
}

// the chain of the announced scan goes into its queue (after whatever the caller has just enqueued)
static int flush_announced(erasor_hip_handle *h) {
    if (!h->ann.valid) return ERASOR_OK;
    h->ann.valid = false;
    int in_front = 0, n_held = 0;
    for (int j = 0; j < h->npend; ++j) (h->q[h->pend[j]].held ? n_held : in_front) += 1;
    // (only while the steps overlap: the early passes then take the third query stream and two chains in flight bound the step -- shared
    // launches halve a chain's queue time; plain steps have three query streams, where holding a chain back only delays it: 9.8 M-point
    // map 0.190 ms per scan alone, 0.196 in sets of two, MEASUREMENTS R6)
    // (... or the handle has ONE query stream -- ERASOR_HIP_QSTREAMS=1: several handles that share a GPU must keep to four busy compute queues
    // between them, two each: main + one query stream, and one query stream is enough only for chains that share launches)
    const bool may_hold = h->batch_n >= 2 && in_front >= h->batch_lead && (overlap_wanted(h, in_front) || h->nqs == 1);
    if (n_held && !may_hold) {  // (keeps the queues in announcement order)
        const int rc_h = flush_held(h);
        if (rc_h) return rc_h;
        n_held = 0;
    }
    const int rc = enqueue_query_chain(h, h->ann.side, h->ann.src, (uint32_t)h->ann.n, h->ann.is_device, h->ann.Tl, false, /*staged=*/1,
                                       h->q_passthrough, h->ann.fmt, may_hold);
    if (rc) return rc;
    h->q[h->ann.side].pose_valid = h->ann.pose_valid;
    h->q[h->ann.side].pose_x = h->ann.pose_x;
    h->q[h->ann.side].pose_y = h->ann.pose_y;
    h->q[h->ann.side].to_valid = h->ann.to_valid;
    memcpy(h->q[h->ann.side].To, h->ann.To, sizeof(h->ann.To));
    h->pend[h->npend++] = h->ann.side;
    if (h->q[h->ann.side].held ? n_held + 1 >= h->batch_n : n_held > 0) return flush_held(h);
    return ERASOR_OK;
}
// a held chain that has come within `batch_lead` steps of its own step goes off now, alone if it has to
static int flush_held_if_due(erasor_hip_handle *h) {
    for (int j = 0; j < h->npend && j < h->batch_lead; ++j)
        if (h->q[h->pend[j]].held) return flush_held(h);
    return ERASOR_OK;
}

// standalone entry points (voxelize, mapgen, sort hooks, ERASOR-class runs) use the active query side as scratch:
// forget prefetched chains and let the query streams run dry first
static void q_drain(erasor_hip_handle *h) {
    h->npend = 0;
    h->ann.valid = false;
    for (int k = 0; k < NSIDE; ++k) h->q[k].held = false;  // (held back and never launched: nothing to wait for)
    (void)chain_wait(h, -1);
    if (h->ov.valid) {  // (the passes launched ahead for the node that is dropped here: they run out, nothing takes them)
        h->ov.valid = false;
        if (h->bstream) (void)hipStreamSynchronize(h->bstream);
        (void)hipStreamSynchronize(h->stream);
    }
    for (int k = 0; k < h->nqs; ++k) (void)hipStreamSynchronize(h->qstream[k]);
    // a dropped announcement's host scan may still be on its way from pinned staging into its side (copy stream): whoever uses the
    // sides as scratch next must not race with that copy
    if (h->cstream) (void)hipStreamSynchronize(h->cstream);
    for (int k = 0; k < NSIDE; ++k) h->q[k].h2d_pending = false;
}


// The VoI split on the main stream.  dev == nullptr: a step's own pass (extents from the host mirror).  dev != nullptr: the
// pass of the NEXT step, launched ahead; the kernel takes the extents the step in flight commits on the device.
static void launch_voi_split(erasor_hip_handle *h, const float4 *F, uint32_t nF, uint32_t nFchunks, uint32_t o_begin, uint32_t o_chunk0,
                             uint32_t nOchunks, uint32_t nchunks_grid, double xc, double yc, double voi_r2, const DevState *dev, uint32_t cap_chunks,
                             const StepEnd *step_end = nullptr, const OvSplit *ovs = nullptr) {
    // one chunk per wave while that needs <= 16 workgroups per CU: the hardware back-fills workgroups as they finish,
    // which balances better than a grid-stride tail (measured, tools/bw_probe.hip)
    // (step_end: the launch's LAST workgroup ends the step in flight -- round 4, see StepEnd in kernels.hip.h)
    StepEnd se;
    memset(&se, 0, sizeof(se));
    if (step_end) se = *step_end;
    OvSplit ov;
    if (ovs) ov = *ovs;
    const uint32_t grid = std::max(1u, std::min<uint32_t>(cdiv(nchunks_grid, 4), 256 * 16)) + (step_end ? 1u : 0u);
    const uint32_t capO_chunks = h->capO / CHUNK;
    // (prof 3: the same on every FOURTH launch -- the bracket costs the step it observes ~9 us (gpurun_out/r03k: 0.276 vs 0.267 ms
    // per scan with / without), a sample of the timed region's launches costs a quarter of that)
    if (ovs && (h->prof == 2 || (h->prof == 3 && (h->n_split_launch++ & 3u) == 0))) {
        // (the roofline measurement of an overlapped step's pass: start / stop events on the early stream, see below)
        PendingEvt ke;
        ke.name_id = prof_id(h, "voi_split");
        ke.a = get_evt(h);
        ke.b = get_evt(h);
        hipExtLaunchKernelGGL(k_voi_split, dim3(grid), dim3(256), 0, h->bstream, ke.a, ke.b, 0, F, nF, nFchunks, (const float2 *)h->Oxy.p, o_begin, o_chunk0,
                              nOchunks, xc, yc, voi_r2, h->vmask.p, h->hmask.p, h->cinfo.p, dev, capO_chunks, cap_chunks,
                              h->use_ometa ? h->ometa.p : (OMeta *)nullptr, se, ov);
        h->pending.push_back(ke);
    } else if (ovs) {  // (an overlapped step's pass: on the early stream, beside the per-bin launch of the step in flight)
        hipStream_t keep = h->cur;
        h->cur = h->bstream;
        LAUNCH(h, "voi_split", k_voi_split, grid, 256, F, nF, nFchunks, (const float2 *)h->Oxy.p, o_begin, o_chunk0, nOchunks, xc, yc, voi_r2, h->vmask.p,
               h->hmask.p, h->cinfo.p, dev, capO_chunks, cap_chunks, h->use_ometa ? h->ometa.p : (OMeta *)nullptr, se, ov);
        h->cur = keep;
    } else if (h->prof == 2 || (h->prof == 3 && (h->n_split_launch++ & 3u) == 0)) {
        // roofline measurement: the launch carries its own start / stop events (hipExtLaunchKernelGGL: they stamp the
        // kernel's execution window itself, the figure rocprofv3 reports too).  A record / record bracket around the
        // launch costs two extra barrier packets on a 17 us kernel and slows the step it is supposed to observe.
        PendingEvt ke;
        ke.name_id = prof_id(h, "voi_split");
        ke.a = get_evt(h);
        ke.b = get_evt(h);
        hipExtLaunchKernelGGL(k_voi_split, dim3(grid), dim3(256), 0, h->stream, ke.a, ke.b, 0, F, nF, nFchunks, (const float2 *)h->Oxy.p, o_begin,
                              o_chunk0, nOchunks, xc, yc, voi_r2, h->vmask.p, h->hmask.p, h->cinfo.p, dev, capO_chunks, cap_chunks,
                              h->use_ometa ? h->ometa.p : (OMeta *)nullptr, se, ov);
        h->pending.push_back(ke);
    } else {
        hipStream_t keep = h->cur;
        h->cur = h->stream;
        LAUNCH(h, "voi_split", k_voi_split, grid, 256, F, nF, nFchunks, (const float2 *)h->Oxy.p, o_begin, o_chunk0, nOchunks, xc, yc, voi_r2,
               h->vmask.p, h->hmask.p, h->cinfo.p, dev, capO_chunks, cap_chunks, h->use_ometa ? h->ometa.p : (OMeta *)nullptr, se, ov);
        h->cur = keep;
    }
}

// The NEXT step's VoI split behind the step in flight (its pose is known: erasor_hip_prefetch_node): reads the store that step writes
// and the extents its end commits; `step_end`: the launch also ends the step in flight (see StepEnd).
static void launch_split_ahead(erasor_hip_handle *h, double nx, double ny, uint32_t nchunks_hint, const StepEnd *step_end) {
    const size_t cap_chunks = std::min(std::min(h->vmask.cap, h->hmask.cap) / CHUNK_TILES, h->cinfo.cap) - 8;
    launch_voi_split(h, (const float4 *)h->F[h->curF ^ 1].p, 0u, 0u, 0u, 0u, 0u, nchunks_hint + 64, nx, ny, h->dp.voi_r2, (const DevState *)h->d_st.p,
                     (uint32_t)cap_chunks, step_end);
    h->spec.valid = true;
    h->spec.x = nx;
    h->spec.y = ny;
    h->spec.seq = h->step_seq;
    h->spec.epoch = h->store_epoch;
    h->spec.curF = h->curF ^ 1;
    h->spec.cap_chunks = (uint32_t)cap_chunks;
    h->spec.vmask = h->vmask.p;
    h->spec.hmask = h->hmask.p;
    h->spec.cinfo = h->cinfo.p;
    h->spec.scan = false;
    h->fly.spec_launched = true;
    ++h->n_spec_launched;
    // round 4: ... and the next step's chunk scan behind it: the stream goes on while the host is still collecting this step's results
    // and comes back with the next one (its turnaround, ~10 us, used to be idle time between the split and the scan)
    const bool mb_count = h->B + 1 <= QB_NB_MAX;
    const uint32_t scan_cap = (uint32_t)std::min<size_t>(std::min(std::min(h->pvl.cap, h->phl.cap), h->cinfo.cap) - 8, 16384);  // (k_chunk_scan_one reads cinfo up to here)
    if (nchunks_hint + 64u <= scan_cap && h->topv.cap >= 24 && h->toph.cap >= 24 && h->prof != 1) {
        hipStream_t keep = h->cur;
        h->cur = h->stream;
        // (round 5: the next step's state, counters and tallies live in the OTHER set, see erasor_hip_handle::alt; it starts from what the
        // step in flight has committed)
        LAUNCH(h, "chunk_scan", k_chunk_scan_one, 1, 1024, (const uint32_t *)h->cinfo.p, 0u, h->pvl.p, h->phl.p, h->topv.p, h->toph.p, 16u, 0u, h->alt.d_st.p,
               h->alt.d_ctr.p, h->st, h->alt.lab_slots.p, mb_count ? h->mb_tot.p : (uint32_t *)nullptr, mb_count ? h->B + 2 : 0u,
               h->use_ometa ? h->ometa.p : (OMeta *)nullptr, h->capO / CHUNK, scan_cap, (const DevState *)h->d_st.p, 0xFFFFFFFFu);
        h->cur = keep;
        h->spec.scan = true;
        h->spec.scan_cap = scan_cap;
        h->spec.pvl = h->pvl.p;
        h->spec.phl = h->phl.p;
    } else if (h->prof != 1) {
        // maps beyond k_chunk_scan_one's 16384 chunks (config 4: 38 k): the two-level scan ahead, its grid an upper bound
        const size_t room = std::min(h->pvl.cap, h->phl.cap);
        const uint32_t grid = (uint32_t)cdiv(nchunks_hint + 64u, 1024);
        const size_t top_cap = std::min(std::min(h->topv.cap, h->toph.cap), h->topr.cap);
        if (room >= 8 && (size_t)grid * 1024 + 8 <= room && grid + 1 <= top_cap) {
            const uint32_t cap2 = grid * 1024u;
            hipStream_t keep = h->cur;
            h->cur = h->stream;
            LAUNCH(h, "chunk_scan", k_chunk_scan_local, grid, 256, (const uint32_t *)h->cinfo.p, 0u, h->pvl.p, h->phl.p, h->topv.p, h->toph.p, h->topr.p,
                   (const DevState *)h->d_st.p, h->capO / CHUNK, cap2, 1u);
            LAUNCH(h, "chunk_scan", k_chunk_scan_top, 1, 1024, h->topv.p, h->toph.p, grid, (const uint32_t *)h->pvl.p, (const uint32_t *)h->phl.p, 0u, 0u,
                   h->alt.d_st.p, h->alt.d_ctr.p, h->st, h->alt.lab_slots.p, mb_count ? h->mb_tot.p : (uint32_t *)nullptr, mb_count ? h->B + 2 : 0u,
                   (const uint32_t *)h->topr.p, h->use_ometa ? h->ometa.p : (OMeta *)nullptr, h->capO / CHUNK, cap2, (const DevState *)h->d_st.p, 0xFFFFFFFFu);
            h->cur = keep;
            h->spec.scan = true;
            h->spec.scan_cap = cap2;
            h->spec.pvl = h->pvl.p;
            h->spec.phl = h->phl.p;
        }
    }
}

// round 5: the two sets of per-step arrays change places (see erasor_hip_handle::alt): what the passes launched ahead wrote becomes the
// current step's, the finished step's becomes scratch for the passes launched ahead of the next
static void swap_sides(erasor_hip_handle *h) {
    std::swap(h->voi_ego, h->alt.voi_ego);
    std::swap(h->voi_key, h->alt.voi_key);
    std::swap(h->voi_src, h->alt.voi_src);
    std::swap(h->moff, h->alt.moff);
    std::swap(h->d_st, h->alt.d_st);
    std::swap(h->d_ctr, h->alt.d_ctr);
    std::swap(h->lab_slots, h->alt.lab_slots);
    // (the bucketed VoI and the bins' statistics too: an overlapped step's scatter and first Scan Ratio Test pass are launched ahead)
    std::swap(h->spts, h->alt.spts);
    std::swap(h->ssrc, h->alt.ssrc);
    std::swap(h->rk_a, h->alt.rk_a);
    std::swap(h->mcnt, h->alt.mcnt);
    std::swap(h->mmin, h->alt.mmin);
    std::swap(h->mmax, h->alt.mmax);
    std::swap(h->st1b, h->alt.st1b);
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}
// ERASOR_HIP_OVERLAP unset: does overlapping consecutive steps pay on this handle's workload?  See the definition of OvAuto.
static void overlap_auto_sample(erasor_hip_handle *h, double period_us);
static int step_collect(erasor_hip_handle *h, erasor_step_result *res);
// First half of a step: everything is ENQUEUED (this scan's query chain unless it is in flight already, the map chain, Scan Ratio
// Test .. write-back, k_step_end, the next step's VoI split and the next announced scan's query chain); nothing is waited for.
// ticket != 0: the scan is the OLDEST announced one and must carry that ticket (erasor_hip_step_ticket): nothing is compared, the
// caller's buffer is not read again; scan_src / n_scan / T_l2b are ignored.
static int step_enqueue(erasor_hip_handle *h, const void *scan_src, size_t n_scan, bool src_is_device, const float T_l2b[16],
                        const float T_b2o[16], const float T_o2b[16], int flags = 0, RowFmt fmt = RowFmt(), uint64_t ticket = 0) {
    if (!h) return ERASOR_E_INVALID;
    if (h->fly.active) {
        h->err = "erasor_hip_step_async: the previous step has not been collected (erasor_hip_step_wait)";
        return ERASOR_E_STATE;
    }
    if (!h->have_map) {
        h->err = h->poisoned ? "an earlier step failed after it had modified the map store: call erasor_hip_set_map again"
                             : "erasor_hip_step before erasor_hip_set_map";
        return ERASOR_E_STATE;
    }
    if (!ticket && (!T_l2b || (!scan_src && n_scan) || n_scan > 0x3FFFFFFFull)) return ERASOR_E_INVALID;
    if (!T_b2o || !T_o2b || (src_is_device && (fmt.stride != 16 || fmt.ioff != 12)) || fmt.stride < 16 || fmt.ioff < 12 || fmt.ioff + 4 > fmt.stride)
        return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    prof_collect(h);
    float Tl_ticket[16];
    if (ticket) {
        // the announced scan, by ticket: still only announced (the first node of a sequence) -> its chain starts now
        if (h->npend == 0 && h->ann.valid && h->ann.ticket == ticket) {
            const int rc_f = flush_announced(h);
            if (rc_f) return rc_f;
        }
        if (h->npend == 0 || h->q[h->pend[0]].ticket != ticket) {
            h->err = "erasor_hip_step_ticket: not the ticket of the oldest announced scan (announcements are consumed in order)";
            return ERASOR_E_STATE;
        }
        const QSide &c = h->q[h->pend[0]];
        memcpy(Tl_ticket, c.Tl, sizeof(Tl_ticket));
        T_l2b = Tl_ticket;
        n_scan = c.src_n;
        src_is_device = c.src_dev;
        // (a host scan lives on in the side's pinned staging copy: that is what a step that has to run again reads)
        scan_src = c.src_dev ? c.src : (const void *)c.stage;
        fmt = RowFmt();
    }
    const auto t_host0 = std::chrono::steady_clock::now();
    static const bool host_timing = getenv("ERASOR_HIP_HOST_TIMING") != nullptr;
    std::vector<std::pair<const char *, double>> marks;
    auto MARK = [&](const char *what) {
        if (host_timing) marks.emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count());
    };
    const uint32_t ns = (uint32_t)n_scan;
    const bool prevox = (flags & STEP_QUERY_PREVOXELIZED) != 0;
    int rc = 0;
    // ---- this scan's query chain: already in flight (erasor_hip_prefetch_scan), or enqueued now -- first, it is the long one ----
    {
        int side = -1;
        // (recognition WITHOUT a ticket: pointer, size, T_lidar2body -- and, for a host scan, the hash of every record of the caller's
        // buffer against the hash of the copy that was staged: only taken when the cheap tests have passed)
        const bool ann_cand = !ticket && !(flags & STEP_RETRIED) && h->npend == 0 && h->ann.valid && !prevox && h->ann.src == scan_src && h->ann.n == n_scan &&
                              h->ann.is_device == src_is_device && memcmp(h->ann.Tl, T_l2b, sizeof(h->ann.Tl)) == 0 &&
                              h->ann.fmt.stride == fmt.stride && h->ann.fmt.ioff == fmt.ioff;
        const bool pend_cand = !ticket && !(flags & STEP_RETRIED) && h->npend > 0 && !prevox && h->q[h->pend[0]].src == scan_src && h->q[h->pend[0]].src_n == n_scan &&
                               h->q[h->pend[0]].src_dev == src_is_device && memcmp(h->q[h->pend[0]].Tl, T_l2b, sizeof(h->q[0].Tl)) == 0 &&
                               h->q[h->pend[0]].fmt.stride == fmt.stride && h->q[h->pend[0]].fmt.ioff == fmt.ioff;
        const uint64_t fp_now = (!src_is_device && (ann_cand || pend_cand)) ? scan_fingerprint(scan_src, n_scan, fmt) : 0ull;
        if (ann_cand && !src_is_device && !h->ann.fp_valid) {  // (the announced copy is hashed now that somebody asks)
            h->ann.fp = staged_fingerprint(h->q[h->ann.side], h->ann.n);
            h->ann.fp_valid = true;
        }
        if (ann_cand && (src_is_device || h->ann.fp == fp_now)) {
            rc = flush_announced(h);  // announced, not yet started (first scan of a sequence): start it now, it is ours
            if (rc) return rc;
        }
        if (flags & STEP_RETRIED) {
            // the step runs again in the other VoxelGrid mode (step_collect): the nodes announced behind it are NOT this scan and are not
            // dropped -- a ticket has no caller buffer to come back with --; their chains (which ran in the old mode) go again below
        } else if (h->npend > 0) {
            QSide &c = h->q[h->pend[0]];
            if (ticket || (!prevox && c.src == scan_src && c.src_n == n_scan && c.src_dev == src_is_device &&
                           (src_is_device || (pend_cand && staged_fingerprint(c, c.src_n) == fp_now)) &&
                           memcmp(c.Tl, T_l2b, sizeof(c.Tl)) == 0 && c.fmt.stride == fmt.stride && c.fmt.ioff == fmt.ioff)) {
                side = h->pend[0];
                for (int j = 1; j < h->npend; ++j) h->pend[j - 1] = h->pend[j];
                --h->npend;
            } else {
                // not the scan that was announced: the prefetched chains are dropped.  Let them run out before going on, so
                // that a device-resident scan they read in place may be released once this call has returned
                q_drain(h);
            }
        }
        if (side < 0) {
            // (an announcement that is not this scan is the NEXT scan: it keeps the side it was staged into)
            side = pick_side(h);
            rc = enqueue_query_chain(h, side, scan_src, ns, src_is_device, T_l2b, prevox, 0, h->q_passthrough && !prevox, fmt);
            if (rc) return rc;
        }
        if (flags & STEP_RETRIED)
            for (int j = 0; j < h->npend; ++j) {  // (in announcement order, behind this step's own chain; a host scan still lies in q.scan)
                QSide &c = h->q[h->pend[j]];
                rc = enqueue_query_chain(h, h->pend[j], c.src, (uint32_t)c.src_n, c.src_dev, c.Tl, false, 2, h->q_passthrough, c.fmt);
                if (rc) return rc;
            }
        h->qi = side;
        h->step_side = side;
        rc = chain_wait(h, side);  // (the chain's events are waited for below: they must have been recorded)
        if (rc) return rc;
    }
    MARK("query chain");
    h->Tl2b = to_xf(T_l2b);
    h->Tb2o = to_xf(T_b2o);
    h->To2b = to_xf(T_o2b);
    const double xc = (double)T_b2o[3], yc = (double)T_b2o[7];  // OMU.cpp:246-247
    swap_sides(h);  // (whatever was launched ahead of this step wrote into the other set: it is this step's now)
    DevState *ds = h->d_st.p;
    Counters *dc = h->d_ctr.p;
    const DP &P = h->dp;
    const uint32_t B = h->B;

    if (h->P.is_large_scale) {  // OMU.cpp:249-251
        rc = reassign_submap(h, xc, yc);
        if (rc) return rc;
    }
    // room for the points that may leave the VoI-resident region (upper bound nF)
    if ((uint64_t)h->o_begin < (uint64_t)h->nF + CHUNK || (uint64_t)(h->capO - h->o_begin) > 2 * h->o_valid + (1u << 20)) {
        rc = rebuild_outskirts(h, h->nF + CHUNK);
        if (rc) return rc;
    }
    const uint64_t n_map_in = (uint64_t)h->nFv + h->o_valid;
    const uint64_t n_voi_room = (uint64_t)h->nF + h->o_valid;  // entries of the VoI-order arrays: reserved slots keep places too (round 5)
    if (n_voi_room + ns + 64 > h->capV) {
        // the reference's map_arranged_ simply grows (scan points enter every reverted bin, erasor.cpp:512; v2 merges whole
        // query bins, erasor.cpp:296-307): enlarge the map-sized scratch between steps (it carries no state across steps)
        rc = grow_map_scratch(h, n_voi_room + ns + 64);
        if (rc) return rc;
    }
    h->st.nF = h->nF;  // the host's mirror of the device state rides along as a kernel argument of k_chunk_scan_top
    h->st.nF_valid = h->nFv;
    h->st.o_begin = h->o_begin;
    const bool mb_count = B + 1 <= QB_NB_MAX;  // the map's bucketing as a one-digit counting sort (else: LSD radix passes)
    // (state push, counters and tallies are reset by k_chunk_scan_top, the first launch of the step that needs them)

    MARK("prologue");
    // ---- sizes, scratch ----
    const double voi_r2 = (flags & STEP_VOI_EVERYTHING) ? HUGE_VAL : P.voi_r2;
    const uint32_t nFchunks = cdiv(h->nF, CHUNK);
    const uint32_t o_chunk0 = h->o_begin / CHUNK;
    const uint32_t nOchunks = h->capO / CHUNK - o_chunk0;
    const uint32_t nchunks = nFchunks + nOchunks;
    // (the masks and chunk counts also hold the NEXT step's split when it is launched ahead: its F region may be longer by
    // this step's VoI + scan, its outskirts region may begin up to nF entries earlier)
    // (round 5: in the reserved layout the next region may be as long as 4 x this step's VoI + 2 x its scan)
    const size_t chunks_room = (size_t)nchunks + cdiv(4 * n_voi_room + 2 * (uint64_t)ns + 64, CHUNK) + cdiv(h->nF, CHUNK) + 8;
    if (chunks_room * CHUNK_TILES + 8 > h->vmask.cap || chunks_room * CHUNK_TILES + 8 > h->hmask.cap || chunks_room + 8 > h->cinfo.cap)
        h->spec.valid = false;  // (re-allocated below: a pass launched ahead wrote into the old buffers)
    if (ensure(h, h->lmask, chunks_room * CHUNK_TILES + 8)) return ERASOR_E_NO_DEVICE;
    if (ensure(h, h->vmask, chunks_room * CHUNK_TILES + 8) || ensure(h, h->hmask, chunks_room * CHUNK_TILES + 8) ||
        ensure(h, h->cinfo, chunks_room + 8) || ensure(h, h->pvl, chunks_room + 8) || ensure(h, h->phl, chunks_room + 8) ||
        ensure(h, h->topv, chunks_room / 1024 + 8) || ensure(h, h->toph, chunks_room / 1024 + 8) || ensure(h, h->topr, chunks_room / 1024 + 8))
        return ERASOR_E_NO_DEVICE;
    // No mid-step read-back: everything is launched on upper-bound grids and reads the actual counts
    // (st->voi_total, st->q_nvox) from device memory.  n_voi <= logical map size, nq <= n_scan.
    const uint32_t n_voi = (uint32_t)std::min<uint64_t>(n_voi_room, 0xFFFFFFF0ull), nq = ns;  // upper bounds from here on
    const uint32_t *nvoi_dev = &ds->voi_total;
    rc = alloc_step(h, n_voi, nq);
    if (rc) return rc;
    if (B + 1 <= QB_NB_MAX) {  // [tiles][B + 1] table of the map's counting sort (round 5: B + 2 with the dead bucket of an overlapped step)
        // (rows for every tile the VoI-order arrays can hold: the NEXT step's table is filled ahead, and its VoI order keeps places for
        // this step's reserved slots -- more entries than this step's own bound)
        if (ensure(h, h->mb_hist, (size_t)mb_row_stride(B + 2) * std::max(1u, cdiv(std::max(n_voi, h->capV), MB_TILE)) + 8)) return ERASOR_E_NO_DEVICE;
    } else {  // the map chain's scratch bank of the radix bucket sort
        const uint32_t nb_m = 256u * std::max(1u, cdiv(n_voi, RTILE));
        if (ensure(h, h->hist2, nb_m) || ensure(h, h->hist2_l, nb_m) || ensure(h, h->hist2_t, cdiv(nb_m, 1024) + 2)) return ERASOR_E_NO_DEVICE;
    }
    const uint32_t *sm_keys = nullptr, *sm_perm = nullptr;
    // v3: the Scan Ratio Test's first pass rides along with the map's bin statistics (k_bin_stats_srt)
    const bool st1_ahead = P.version == 3 && B <= 1024 * SRT_KPT;
    // up to 4096 bins the output layout has no launch of its own: k_srt4 prepares what does not depend on R-GPF, every workgroup of
    // the write-back finishes it (beyond that, and under per-launch profiling: k_layout4 / k_layout)
    const bool fold = B <= 1024 * SRT_KPT && n_voi > 0;
    // round 4: v3's second pass is the LAST WORKGROUP of the per-bin launch (k_revert_bins_srt): the revert decision is local to a bin, so
    // the per-bin workgroups find their bins themselves and nothing on the chain waits for k_srt4
    const bool srt_in_revert = st1_ahead && h->prof != 1;
    // round 5: the RESERVED layout (srt4_body): the write-back of everything but the reverted bins does not wait for the per-bin launch.
    // ERASOR_HIP_OVERLAP: 1 = always, 0 = never, unset = the handle decides (overlap_pays).
    const int announced_beyond = h->npend + (h->ann.valid ? 1 : 0);  // nodes announced behind this step's own
    const bool overlap_pays = overlap_wanted(h, announced_beyond);
    h->fly.deep = announced_beyond >= h->batch_lead;
    // (decided BEFORE the map chain: a step that does not write the reserved layout must not take passes launched ahead on the assumption
    // that it would -- their VoI-order source indices count reserved slots, which only the reserved write-back converts)
    const bool reserved = srt_in_revert && fold && mb_count && !flags && overlap_pays && early_stream(h);
    bool use_ov = false;   // this step's split .. bucket table were launched ahead of it and are taken (round 5, OVERLAPPED steps)
    bool stats_ahead = false;  // ... its scatter and bin statistics too
    uint32_t nbk = B + 1;  // buckets of the map's counting sort: the bins + the complement (+ the dead bucket of an overlapped step)
    int bits = key_bits(B + 1);

    // ---- map chain (the query chains run on their own streams; the two only meet at the Scan Ratio Test) ----
    auto enqueue_map_chain = [&]() {
        MARK("  mapchain_begin");
        // With the query chains on streams of their own, the map chain simply opens the step on the main stream
        // (no fork / join events, one hardware queue less).
        h->cur = h->stream;
        h->bank = 1;
        {   // round 5: split, chunk scan, gather and the bucket table may all be there already -- launched beside the previous step's per-bin
            // launch (h->ov) -- if this step is exactly the step that was assumed then: node, pose, transform, store, buffers, room
            const bool ov_launched = h->ov.valid;
            use_ov = ov_launched && reserved && !flags && mb_count && h->ov.seq == h->step_seq && h->ov.epoch == h->store_epoch && h->ov.curF == h->curF &&
                     h->ov.qside == h->qi && h->ov.x == xc && h->ov.y == yc && memcmp(h->ov.To, T_o2b, sizeof(h->ov.To)) == 0 &&
                     nchunks <= h->ov.cap_chunks && n_voi_room + 64 <= h->ov.cap_voi && h->ov.vmask == (const void *)h->vmask.p &&
                     h->ov.hmask == (const void *)h->hmask.p && h->ov.lmask == (const void *)h->lmask.p && h->ov.cinfo == (const void *)h->cinfo.p &&
                     h->ov.pvl == (const void *)h->pvl.p && h->ov.phl == (const void *)h->phl.p && h->ov.voi_ego == (const void *)h->voi_ego.p &&
                     h->ov.mb_hist == (const void *)h->mb_hist.p;
            if (ov_launched && !use_ov && getenv("ERASOR_HIP_CHAIN_STAMPS"))
                fprintf(stderr, "[overlap passes not taken] flags %d seq %d epoch %d curF %d qside %d pose %d To %d chunks %u/%u voi %llu/%u ptrs %d%d%d%d%d%d%d%d\n", flags,
                        (int)(h->ov.seq == h->step_seq), (int)(h->ov.epoch == h->store_epoch), (int)(h->ov.curF == h->curF), (int)(h->ov.qside == h->qi),
                        (int)(h->ov.x == xc && h->ov.y == yc), (int)(memcmp(h->ov.To, T_o2b, sizeof(h->ov.To)) == 0), nchunks, h->ov.cap_chunks,
                        (unsigned long long)n_voi_room + 64, h->ov.cap_voi, (int)(h->ov.vmask == (const void *)h->vmask.p), (int)(h->ov.hmask == (const void *)h->hmask.p),
                        (int)(h->ov.lmask == (const void *)h->lmask.p), (int)(h->ov.cinfo == (const void *)h->cinfo.p), (int)(h->ov.pvl == (const void *)h->pvl.p),
                        (int)(h->ov.phl == (const void *)h->phl.p), (int)(h->ov.voi_ego == (const void *)h->voi_ego.p), (int)(h->ov.mb_hist == (const void *)h->mb_hist.p));
            h->ov.valid = false;
            stats_ahead = use_ov && h->ov.stats_done && st1_ahead;
            if (ov_launched && !use_ov) (void)hipStreamWaitEvent(h->stream, h->ev_early, 0);  // (those passes write what this step's own will)
            if (use_ov) {
                ++h->n_ov_used;
                ++h->n_spec_used;
                h->spec.valid = false;
                nbk = h->ov.nbk;
                bits = key_bits(nbk);
            }
        }
        if (!use_ov) {   // VoI split (OMU.cpp:254 fetch_VoI membership)
            // ... unless it is already there: launched ahead by the previous step for exactly this pose and this store
            const bool use_spec = h->spec.valid && !flags && h->spec.seq == h->step_seq && h->spec.epoch == h->store_epoch &&
                                  h->spec.curF == h->curF && h->spec.x == xc && h->spec.y == yc && nchunks <= h->spec.cap_chunks &&
                                  h->spec.vmask == (const void *)h->vmask.p && h->spec.hmask == (const void *)h->hmask.p &&
                                  h->spec.cinfo == (const void *)h->cinfo.p;
            const bool scan_done = use_spec && h->spec.scan && nchunks <= h->spec.scan_cap && h->spec.pvl == (const void *)h->pvl.p &&
                                   h->spec.phl == (const void *)h->phl.p;
            h->spec.valid = false;
            h->spec.scan = false;
            if (use_spec) ++h->n_spec_used;
            else
                launch_voi_split(h, (const float4 *)h->F[h->curF].p, h->nF, nFchunks, h->o_begin, o_chunk0, nOchunks, nchunks, xc, yc, voi_r2,
                                 (const DevState *)nullptr, 0u);
            const uint32_t ntop = std::max(1u, cdiv(nchunks, 1024));
            if (scan_done) {
                // (launched ahead behind the split, see launch_split_ahead)
            } else if (nchunks <= 16384) {
                LAUNCH(h, "chunk_scan", k_chunk_scan_one, 1, 1024, (const uint32_t *)h->cinfo.p, nchunks, h->pvl.p, h->phl.p, h->topv.p, h->toph.p, ntop,
                       nFchunks, ds, dc, h->st, h->lab_slots.p, mb_count ? h->mb_tot.p : (uint32_t *)nullptr, mb_count ? B + 2 : 0u,
                       h->use_ometa ? h->ometa.p : (OMeta *)nullptr, 0u, 0u, (const DevState *)nullptr, 0u);
            } else {
                LAUNCH(h, "chunk_scan", k_chunk_scan_local, ntop, 256, (const uint32_t *)h->cinfo.p, nchunks, h->pvl.p, h->phl.p, h->topv.p, h->toph.p,
                       h->topr.p, (const DevState *)nullptr, 0u, 0u, 0u);
                LAUNCH(h, "chunk_scan", k_chunk_scan_top, 1, 1024, h->topv.p, h->toph.p, ntop, (const uint32_t *)h->pvl.p, (const uint32_t *)h->phl.p,
                       nchunks, nFchunks, ds, dc, h->st, h->lab_slots.p, mb_count ? h->mb_tot.p : (uint32_t *)nullptr, mb_count ? B + 2 : 0u,
                       (const uint32_t *)h->topr.p, h->use_ometa ? h->ometa.p : (OMeta *)nullptr, 0u, 0u, (const DevState *)nullptr, 0u);
            }
        }
        if (!use_ov) (void)hipStreamWaitEvent(h->stream, Q(h).ev_keys, 0);  // k_voi_gather must see the query side's error flag
        if (!use_ov) {   // VoI gather + egocentric transform + R-POD key (OMU.cpp:435-437; erasor.cpp:124-139)
            // one wavefront per work item: GATHER_SUB pieces per chunk of the VoI-resident region, one per outskirts chunk
            const uint32_t grid = std::max(1u, std::min<uint32_t>(cdiv(nFchunks * GATHER_SUB + nOchunks, 4), 256 * 16));
            LAUNCH(h, "voi_gather", k_voi_gather, grid, 256, (const float4 *)h->F[h->curF].p, h->nF, nFchunks, h->Oxy.p, h->Ozi.p, o_chunk0,
                   nOchunks, (const unsigned long long *)h->vmask.p, (const unsigned long long *)h->hmask.p, (const uint32_t *)h->cinfo.p,
                   (const uint32_t *)h->pvl.p, (const uint32_t *)h->phl.p, (const uint32_t *)h->topv.p, (const uint32_t *)h->toph.p, h->To2b, P, ds,
                   dc, (const Counters *)Q(h).d_qctr.p, h->voi_ego.p, h->voi_key.p, h->voi_src.p, h->use_ometa ? h->ometa.p : (OMeta *)nullptr,
                   (const unsigned long long *)nullptr, 0u);
        }
        if (mb_count) {
            // the table is sized for the whole map (n_voi is an upper bound), the GRIDS for 1.5 x the previous step's VoI: the
            // kernels walk the tiles with a grid stride, so an under-estimate only costs a second round
            const uint32_t ntile_tab = std::max(1u, cdiv(n_voi, MB_TILE));
            const uint32_t ntile_ub = h->last_n_voi ? std::min(ntile_tab, std::max(64u, cdiv((uint64_t)h->last_n_voi * 3 / 2, MB_TILE))) : ntile_tab;
            if (!use_ov) {  // (an overlapped step: both launched ahead, behind k_late_gather)
                LAUNCH(h, "voi_bucket", k_mb_hist, ntile_ub, 1024, (const uint32_t *)h->voi_key.p, n_voi, nvoi_dev, nbk, h->mb_hist.p, h->mb_tot.p);
                LAUNCH(h, "voi_bucket", k_mb_colscan, cdiv(nbk, MB_PAD), 256, h->mb_hist.p, n_voi, nvoi_dev, nbk, (const uint32_t *)h->mb_tot.p, h->moff.p);
            }
            if (use_ov && stats_ahead) {
                // (scatter, statistics and first Scan Ratio Test pass were launched ahead as well)
            } else if (nbk <= MBW_NB_SMALL)
                LAUNCH(h, "voi_bucket", k_mb_scatter_w<MBW_NB_SMALL>, ntile_ub, 1024, (const uint32_t *)h->voi_key.p, (const float4 *)h->voi_ego.p,
                       (const uint32_t *)h->voi_src.p, n_voi, nvoi_dev, nbk, bits, (const uint32_t *)h->mb_hist.p, h->spts.p, h->ssrc.p, h->rk_a.p,
                       h->dbg_stamps.p);
            else if (nbk <= MBW_NB_MAX)
                LAUNCH(h, "voi_bucket", k_mb_scatter_w<MBW_NB_MAX>, ntile_ub, 1024, (const uint32_t *)h->voi_key.p, (const float4 *)h->voi_ego.p,
                       (const uint32_t *)h->voi_src.p, n_voi, nvoi_dev, nbk, bits, (const uint32_t *)h->mb_hist.p, h->spts.p, h->ssrc.p, h->rk_a.p,
                       h->dbg_stamps.p);
            else
                LAUNCH(h, "voi_bucket", k_mb_scatter, ntile_ub, 1024, (const uint32_t *)h->voi_key.p, (const float4 *)h->voi_ego.p,
                       (const uint32_t *)h->voi_src.p, n_voi, nvoi_dev, nbk, bits, (const uint32_t *)h->mb_hist.p, h->spts.p, h->ssrc.p, h->rk_a.p);
            sm_keys = h->rk_a.p;
        } else {
            radix_sort(h, h->voi_key.p, n_voi, nvoi_dev, bits, h->rk_a.p, h->rk_b.p, h->rv_a.p, h->rv_b.p, &sm_keys, &sm_perm, "voi_bucket");
            if (n_voi) LAUNCH(h, "voi_bucket", k_gather, cdiv(n_voi, 256), 256, (const float4 *)h->voi_ego.p, (const uint32_t *)h->voi_src.p, sm_perm,
                              n_voi, nvoi_dev, h->spts.p, h->ssrc.p);
            LAUNCH(h, "voi_bucket", k_bin_offsets, cdiv((uint64_t)std::max(n_voi, B + 2) + 1, 256), 256, sm_keys, n_voi, nvoi_dev, B + 1, h->moff.p);
        }
        // v3: the Scan Ratio Test's first pass rides along, bin by bin (the query's statistics are needed: the join with its chain
        // comes before this launch instead of after it; with nodes announced ahead the chain finished long ago)
        if (use_ov && stats_ahead) {
            // (done)
        } else if (st1_ahead) {
            (void)hipStreamWaitEvent(h->stream, Q(h).ev_done, 0);
            LAUNCH(h, "bin_stats", k_bin_stats_srt, cdiv((uint64_t)B * 64, 256), 256, P, (const float4 *)h->spts.p, (const uint32_t *)h->moff.p, B, h->mcnt.p,
                   h->mmin.p, h->mmax.p, (const uint32_t *)Q(h).ccnt.p, (const float *)Q(h).cmin.p, (const float *)Q(h).cmax.p, h->st1b.p);
        } else
        LAUNCH(h, "bin_stats", k_bin_stats, cdiv((uint64_t)B * 64, 256), 256, (const float4 *)h->spts.p, (const uint32_t *)h->moff.p, B, h->mcnt.p,
               h->mmin.p, h->mmax.p);
        if (use_ov) {
            // the entering outskirts entries are erased NOW that the step is certain (k_voi_gather left them: a getter could still ask
            // for the store as the previous step left it); on the early stream, in front of the next pass over the store (enqueued
            // behind the launches the main stream was waiting for)
            h->cur = h->bstream;
            LAUNCH(h, "voi_gather", k_o_commit, std::max(1u, std::min<uint32_t>(cdiv(nOchunks, 4), 2048)), 256, h->Oxy.p,
                   (const unsigned long long *)h->vmask.p, (const uint32_t *)h->cinfo.p, (const DevState *)ds, h->capO / CHUNK,
                   h->use_ometa ? h->ometa.p : (OMeta *)nullptr, (const Counters *)dc, (const Counters *)Q(h).d_qctr.p);
        }
        MARK("  mapchain_end");
        h->cur = h->stream;
        h->bank = 0;
    };

    enqueue_map_chain();
    MARK("map chain");
    HIPC(h, hipStreamWaitEvent(h->stream, Q(h).ev_done, 0));  // join: the query's bins are ready

    // ---- Scan Ratio Test, R-GPF, per-bin voxelisation (erasor.cpp:332-571) ----
    h->ov_mode = reserved;
    bool have_pose = false;  // the NEXT node's pose, if it was announced with it (erasor_hip_prefetch_node)
    double nx = 0, ny = 0;
    if (h->npend > 0) {
        const QSide &nq_ = h->q[h->pend[0]];
        have_pose = nq_.pose_valid;
        nx = nq_.pose_x;
        ny = nq_.pose_y;
    } else if (h->ann.valid) {
        have_pose = h->ann.pose_valid;
        nx = h->ann.pose_x;
        ny = h->ann.pose_y;
    }
    float4 *Fnew = h->F[h->curF ^ 1].p;
    LateEnt *late_out = h->late[h->late_idx ^ 1].p;
    // round 5, OVERLAPPED steps: with the next node announced together with BOTH its transforms (erasor_hip_prefetch_node +
    // erasor_hip_announce_origin2body) its split, chunk scan and gather run beside this step's per-bin launch, on the early stream
    bool ov_next = false;
    int nxt_side = -1;
    if (reserved && have_pose && h->prof != 1 && !submap_would_move(h, nx, ny)) {
        if (h->npend == 0 && h->ann.valid && h->ann.to_valid) {
            // (only just announced: its chain goes into its queue NOW -- the gather ahead waits for the chain's error flag)
            const int keep_side = h->qi;
            const int rc_f = flush_announced(h);
            h->qi = keep_side;
            if (rc_f) return rc_f;
        }
        if (h->npend > 0 && h->q[h->pend[0]].pose_valid && h->q[h->pend[0]].to_valid) {
            ov_next = true;
            nxt_side = h->pend[0];
            const int rc_w = chain_wait(h, nxt_side);  // (its ev_keys is waited for on the early stream)
            if (rc_w) return rc_w;
        }
    }
    std::function<void()> enqueue_early;
    if (getenv("ERASOR_HIP_CHAIN_STAMPS") && !ov_next)
        fprintf(stderr, "[no overlap for the next step] reserved %d (srt_in_revert %d fold %d mb_count %d flags %d) have_pose %d prof %d npend %d ann %d/%d/%d pend0 %d/%d\n",
                (int)reserved, (int)srt_in_revert, (int)fold, (int)mb_count, flags, (int)have_pose, h->prof, h->npend, (int)h->ann.valid, (int)h->ann.pose_valid,
                (int)h->ann.to_valid, h->npend ? (int)h->q[h->pend[0]].pose_valid : -1, h->npend ? (int)h->q[h->pend[0]].to_valid : -1);
    if (reserved) {
        // a reverted bin whose points may lie outside the NEXT VoI circle reserves places in the outskirts' order as well: without the
        // next pose, every bin does
#ifdef ERASOR_HIP_TEST_HOOKS
        static const bool leave_all = getenv("ERASOR_HIP_LEAVE_ALL") != nullptr;  // (test hook: every reverted bin reserves places in the outskirts' order)
#else
        constexpr bool leave_all = false;
#endif
        const double leave_lim = (have_pose && !leave_all) ? sqrt(P.voi_r2) - hypot(nx - xc, ny - yc) - 0.05 : -1.0;
        // (enqueued BEHIND the per-bin launch, which the main stream is waiting for: see enqueue_early() below)
        enqueue_early = [&, leave_lim]() {
        MARK("srt: begin");
        (void)hipStreamWaitEvent(h->bstream, h->ev_stats, 0);
        MARK("  ev_stats record + wait");
        h->cur = h->bstream;  // ---- the early stream: beside the per-bin launch
        // (round 6: ONE launch -- every workgroup of the early write-back derives the reserved offsets it needs from the status bytes itself,
        // the launch's last workgroup is k_srt4 for everything behind it: see k_assemble_early)
        {
            EarlyArgs ea;
            ea.mcnt = h->mcnt.p;
            ea.mmin = h->mmin.p;
            ea.mmax = h->mmax.p;
            ea.ccnt = Q(h).ccnt.p;
            ea.cmin = Q(h).cmin.p;
            ea.cmax = Q(h).cmax.p;
            ea.st1 = h->st1.p;
            ea.status = h->status.p;
            ea.action = h->action.p;
            ea.rev_idx = h->rev_idx.p;
            ea.rev_list = h->rev_list.p;
            ea.vox_off = h->vox_off.p;
            ea.st = ds;
            ea.out_off0 = h->out_off0.p;
            ea.rev_before = h->rev_before.p;
            ea.crej_off = h->crej_off.p;
            ea.st1_in = h->st1b.p;
            ea.moff = h->moff.p;
            ea.qoff = Q(h).qoff.p;
            ea.out_offR = h->out_offR.p;
            ea.gres_off = h->gres_off.p;
            ea.late = late_out;
            ea.leave_lim = leave_lim;
            ea.skeys = sm_keys;
            ea.spts = h->spts.p;
            ea.Fnew = Fnew;
            ea.cnt = h->lab_slots.p;
            // (one round of workgroups: a 1024-thread workgroup of this kernel has a compute unit to itself, and the per-bin launch and the query
            // chains hold some of the 256)
            LAUNCH(h, "srt+assemble", k_assemble_early, std::min<uint32_t>(cdiv(n_voi, 1024), ERASOR_EARLY_GMAP) + 1u, 1024, P, h->Tb2o, ea);
        }
        MARK("  k_srt4 + assemble early");
        // (an overlapped step's late write-back joins at ev_scan, further down this stream and long there when the per-bin launch ends)
        if (!ov_next) (void)hipEventRecord(h->ev_srt4, h->bstream);
        MARK("  ev_srt4 record");
        MARK("  assemble early");
        if (!ov_next) (void)hipEventRecord(h->ev_asm, h->bstream);  // (an overlapped step joins at ev_early: one packet less on the early chain)
        MARK("  ev_asm record");
        if (ov_next) {
            const QSide &nq_ = h->q[nxt_side];
            const size_t cap_chunks = std::min(std::min(std::min(h->vmask.cap, h->hmask.cap), h->lmask.cap) / CHUNK_TILES, h->cinfo.cap) - 8;
            const uint32_t cap_voi = h->capV - 64u;
            OvSplit ovs;
            ovs.late = late_out;
            ovs.prev = ds;
            ovs.lmask = h->lmask.p;
            // (its extents come from *ds: the region this step writes and the outskirts as its gather has left them)
            launch_voi_split(h, (const float4 *)Fnew, 0u, 0u, 0u, 0u, 0u, nchunks + 64 + cdiv((uint64_t)ns + 4096, CHUNK), nx, ny, P.voi_r2,
                             (const DevState *)nullptr, (uint32_t)cap_chunks, (const StepEnd *)nullptr, &ovs);
            const uint32_t scan_cap1 = (uint32_t)std::min<size_t>(std::min(h->pvl.cap, h->phl.cap) - 8, 16384);
            // (the next region's extent is not known here: it keeps this step's reserved places -- up to 4 x the VoI + 2 x the scan, see
            // alloc_step --, and the outskirts grow by what leaves; one workgroup scans up to 16384 chunk counts, beyond that two levels
            // over everything the buffers hold.  What the scan ahead can take is recorded: a step with more chunks runs its own passes)
            const uint32_t hint = nchunks + 64 + cdiv(4 * (uint64_t)std::max(h->last_n_voi, 1u << 16) + 2 * (uint64_t)ns + 4096, CHUNK);
            uint32_t scan_cap_used = 0;
            if (hint <= scan_cap1) {
                scan_cap_used = std::min<uint32_t>(scan_cap1, (uint32_t)cap_chunks);
                LAUNCH(h, "chunk_scan", k_chunk_scan_one, 1, 1024, (const uint32_t *)h->cinfo.p, 0u, h->pvl.p, h->phl.p, h->topv.p, h->toph.p, 16u, 0u,
                       h->alt.d_st.p, h->alt.d_ctr.p, h->st, h->alt.lab_slots.p, h->mb_tot.p, B + 2, h->use_ometa ? h->ometa.p : (OMeta *)nullptr,
                       h->capO / CHUNK, std::min<uint32_t>(scan_cap1, (uint32_t)cap_chunks), (const DevState *)ds, cap_voi);
            } else {
                const size_t top_cap = std::min(std::min(h->topv.cap, h->toph.cap), h->topr.cap);
                const uint32_t grid2 = (uint32_t)std::min<size_t>(cdiv(std::min<size_t>(cap_chunks, std::min(h->pvl.cap, h->phl.cap) - 8), 1024), top_cap - 2);
                const uint32_t cap2 = std::min<uint32_t>(grid2 * 1024u, (uint32_t)cap_chunks);
                scan_cap_used = cap2;
                LAUNCH(h, "chunk_scan", k_chunk_scan_local, grid2, 256, (const uint32_t *)h->cinfo.p, 0u, h->pvl.p, h->phl.p, h->topv.p, h->toph.p, h->topr.p,
                       (const DevState *)ds, h->capO / CHUNK, cap2, 1u);
                LAUNCH(h, "chunk_scan", k_chunk_scan_top, 1, 1024, h->topv.p, h->toph.p, grid2, (const uint32_t *)h->pvl.p, (const uint32_t *)h->phl.p, 0u, 0u,
                       h->alt.d_st.p, h->alt.d_ctr.p, h->st, h->alt.lab_slots.p, h->mb_tot.p, B + 2, (const uint32_t *)h->topr.p,
                       h->use_ometa ? h->ometa.p : (OMeta *)nullptr, h->capO / CHUNK, cap2, (const DevState *)ds, cap_voi);
            }
            MARK("  split + scan ahead");
            // (round 6: the late gather needs the masks and the chunk prefix, not the gather: it joins HERE and runs beside the gather ahead)
            (void)hipEventRecord(h->ev_scan, h->bstream);
            (void)hipStreamWaitEvent(h->bstream, nq_.ev_keys, 0);  // k_voi_gather must see that query side's error flag
            MARK("  ev_keys wait");
            {
                const uint32_t grid = std::max(1u, std::min<uint32_t>(cdiv((nFchunks + 64) * GATHER_SUB + nOchunks + 64, 4), 256 * 16));
                LAUNCH(h, "voi_gather", k_voi_gather, grid, 256, (const float4 *)Fnew, 0u, 0u, h->Oxy.p, h->Ozi.p, 0u, 0u, (const unsigned long long *)h->vmask.p,
                       (const unsigned long long *)h->hmask.p, (const uint32_t *)h->cinfo.p, (const uint32_t *)h->pvl.p, (const uint32_t *)h->phl.p,
                       (const uint32_t *)h->topv.p, (const uint32_t *)h->toph.p, to_xf(nq_.To), P, h->alt.d_st.p, h->alt.d_ctr.p,
                       (const Counters *)nq_.d_qctr.p, h->alt.voi_ego.p, h->alt.voi_key.p, h->alt.voi_src.p, h->use_ometa ? h->ometa.p : (OMeta *)nullptr,
                       (const unsigned long long *)h->lmask.p, h->capO / CHUNK);
            }
            MARK("  gather ahead");
            (void)hipEventRecord(h->ev_early, h->bstream);
            MARK("  ev_early record");
            h->ov.valid = true;
            h->ov.seq = h->step_seq + 1;  // (this step's number: checked by the step that follows before it takes a number of its own)
            h->ov.epoch = h->store_epoch;
            h->ov.curF = h->curF ^ 1;
            h->ov.qside = nxt_side;
            h->ov.x = nx;
            h->ov.y = ny;
            memcpy(h->ov.To, nq_.To, sizeof(h->ov.To));
            h->ov.cap_chunks = std::min<uint32_t>((uint32_t)cap_chunks, scan_cap_used);
            h->ov.cap_voi = cap_voi;
            h->ov.nbk = B + 2;
            h->ov.vmask = h->vmask.p;
            h->ov.hmask = h->hmask.p;
            h->ov.lmask = h->lmask.p;
            h->ov.cinfo = h->cinfo.p;
            h->ov.pvl = h->pvl.p;
            h->ov.phl = h->phl.p;
            h->ov.voi_ego = h->alt.voi_ego.p;
            h->ov.mb_hist = h->mb_hist.p;
            ++h->n_ov_launched;
            ++h->n_spec_launched;  // (it is the split ahead of this node, with company)
            h->fly.spec_launched = true;
        }
        h->cur = h->stream;  // ---- back on the main stream
        };
    } else if (srt_in_revert) {
        // (no launch)
    } else if (B <= 1024 * SRT_KPT)
        LAUNCH(h, "srt", k_srt4, 1, 1024, P, (const uint32_t *)h->mcnt.p, (const float *)h->mmin.p, (const float *)h->mmax.p, (const uint32_t *)Q(h).ccnt.p,
           (const float *)Q(h).cmin.p, (const float *)Q(h).cmax.p, h->st1.p, h->status.p, h->action.p, h->rev_idx.p, h->rev_list.p, h->vox_off.p, ds,
           fold ? h->out_off0.p : (uint32_t *)nullptr, fold ? h->rev_before.p : (uint32_t *)nullptr, fold ? h->crej_off.p : (uint32_t *)nullptr,
           st1_ahead ? (const uint8_t *)h->st1b.p : (const uint8_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
           (uint32_t *)nullptr, (LateEnt *)nullptr, -1.0);
    else
        LAUNCH(h, "srt", k_srt, 1, 1024, P, (const uint32_t *)h->mcnt.p, (const float *)h->mmin.p, (const float *)h->mmax.p, (const uint32_t *)Q(h).ccnt.p,
           (const float *)Q(h).cmin.p, (const float *)Q(h).cmax.p, h->st1.p, h->status.p, h->action.p, h->rev_idx.p, h->rev_list.p, h->vox_off.p, ds);
    // R-GPF and the per-bin voxelisation walk the reverted-bin LIST on a small fixed grid (n_rev is on the device)
    const uint32_t rev_grid = ERASOR_REV_GRID;
    // v3: R-GPF and the per-bin voxelisation of a reverted bin in ONE launch (k_revert_bins); ERASOR_HIP_NO_FUSE=1: two launches (A/B)
    RevArgs ra;
    ra.moff = h->moff.p;
    ra.spts = h->spts.p;
    ra.qoff = Q(h).qoff.p;
    ra.sq = Q(h).sq.p;
    ra.gsK = h->gsK.p;
    ra.gsV = h->gsV.p;
    ra.gsL = h->gsL.p;
    ra.gsR = h->gsR.p;
    ra.gsH = h->gsH.p;
    ra.gsK2 = h->gsK2.p;
    ra.gsV2 = h->gsV2.p;
    ra.gsC = h->gsC.p;
    ra.gflag = h->gflag.p;
    ra.grank = h->grank.p;
    ra.glist = h->glist.p;
    ra.ng_arr = h->ng.p;
    ra.plane_n = h->plane_n.p;
    ra.plane_d = h->plane_d.p;
    ra.vox_out = h->vox_out.p;
    ra.nvox_out = h->nvox.p;
    ra.ctr = dc;
    ra.dbg = h->dbg_stamps.p;
    // (fused launch: the voxelisation's global-memory path works in the upper half of the shared scratch arrays)
    ra.vox_base = (uint32_t)((size_t)n_voi + nq + 8);
    ra.h_base = (uint32_t)(((size_t)n_voi + nq + 8) / 32 + 2 * (size_t)B + 16);
    if (srt_in_revert) {
        SrtArgs sa;
        sa.mcnt = h->mcnt.p;
        sa.mmin = h->mmin.p;
        sa.mmax = h->mmax.p;
        sa.ccnt = Q(h).ccnt.p;
        sa.cmin = Q(h).cmin.p;
        sa.cmax = Q(h).cmax.p;
        sa.st1 = h->st1.p;
        sa.status = h->status.p;
        sa.action = h->action.p;
        sa.rev_idx = h->rev_idx.p;
        sa.rev_list = h->rev_list.p;
        sa.vox_off = h->vox_off.p;
        sa.st = ds;
        sa.out_off0 = fold ? h->out_off0.p : (uint32_t *)nullptr;
        sa.rev_before = fold ? h->rev_before.p : (uint32_t *)nullptr;
        sa.crej_off = fold ? h->crej_off.p : (uint32_t *)nullptr;
        sa.st1_in = h->st1b.p;
        sa.moff = h->moff.p;
        sa.qoff = Q(h).qoff.p;
        if (reserved) sa.status = nullptr;  // (k_srt4 is the first workgroup of k_assemble_early, on the early stream: no extra workgroup here)
        if (reserved) (void)hipEventRecord(h->ev_stats, h->stream);  // (the bin statistics are there: the early stream may start)
        LAUNCH(h, "rgpf+bin_voxelize", k_revert_bins_srt, std::min<uint32_t>(rev_grid, B) + (reserved ? 0u : 1u), 1024, P, sa, ra);
        MARK("per-bin launch");
        if (reserved) enqueue_early();
        // (the reverted list, the reserved offsets, the late table; ev_scan: also the next step's masks and chunk prefix, for the late gather)
        if (reserved) (void)hipStreamWaitEvent(h->stream, ov_next ? h->ev_scan : h->ev_srt4, 0);
        MARK("  ev_srt4 wait");
        if (reserved)
            LAUNCH(h, "assemble", k_assemble_late, std::min<uint32_t>(rev_grid, B) + 1u, 256, P, h->Tb2o, (const uint32_t *)h->rev_list.p, (const float4 *)h->spts.p,
                   (const uint32_t *)h->ssrc.p, (const uint32_t *)h->moff.p, (const uint32_t *)Q(h).qoff.p, (const uint8_t *)h->gflag.p,
                   (const uint32_t *)h->grank.p, (const uint32_t *)h->ng.p, (const uint32_t *)h->nvox.p, (const uint32_t *)h->vox_off.p,
                   (const float4 *)h->vox_out.p, (const uint32_t *)h->out_offR.p, (const uint32_t *)h->gres_off.p, (const uint32_t *)h->out_off0.p,
                   (const uint32_t *)h->rev_before.p, late_out, h->late_holes[h->late_idx ^ 1].p, h->out_off.p, h->ground_off.p, h->rej_off.p, ds, Fnew,
                   h->rejected.p, h->rejected_src.p, h->lab_slots.p,
                   // (an overlapped step read a region with reserved slots as if they were entries: its source indices count them)
                   use_ov ? (const LateEnt *)h->late[h->late_idx].p : (const LateEnt *)nullptr,
                   use_ov ? (const uint32_t *)h->late_holes[h->late_idx].p : (const uint32_t *)nullptr, use_ov ? h->n_late_F : 0u);
    } else if (P.version == 3 && h->prof != 1) {
        LAUNCH(h, "rgpf+bin_voxelize", k_revert_bins, std::min<uint32_t>(rev_grid, B), 1024, P, (const uint32_t *)h->rev_list.p, (const DevState *)ds,
               (const uint32_t *)h->vox_off.p, ra);
    } else {
        // (two launches: the stages' global-memory paths do not overlap in time and share the scratch arrays from their start)
        ra.vox_base = 0u;
        ra.h_base = 0u;
        LAUNCH(h, "rgpf", k_rgpf2, std::min<uint32_t>(rev_grid, B), 1024, P, (const uint32_t *)h->rev_list.p, (const DevState *)ds, ra);
        if (P.version == 3)
            LAUNCH(h, "bin_voxelize", k_binvox2, std::min<uint32_t>(rev_grid, B), 1024, P, (const uint32_t *)h->rev_list.p, (const DevState *)ds,
                   (const uint32_t *)h->vox_off.p, ra);
    }
    if (fold) {
        // (no launch)
    } else if (B <= 1024 * SRT_KPT)
        LAUNCH(h, "layout", k_layout4, 1, 1024, P, (const uint8_t *)h->action.p, (const uint32_t *)h->rev_idx.p, (const uint32_t *)h->mcnt.p,
           (const uint32_t *)Q(h).ccnt.p, (const uint32_t *)h->moff.p, (const uint32_t *)h->nvox.p, (const uint32_t *)h->ng.p, h->out_off.p,
           h->ground_off.p, h->rej_off.p, h->crej_off.p, ds);
    else
        LAUNCH(h, "layout", k_layout, 1, 1024, P, (const uint8_t *)h->action.p, (const uint32_t *)h->rev_idx.p, (const uint32_t *)h->mcnt.p,
           (const uint32_t *)Q(h).ccnt.p, (const uint32_t *)h->moff.p, (const uint32_t *)h->nvox.p, (const uint32_t *)h->ng.p, h->out_off.p,
           h->ground_off.p, h->rej_off.p, h->crej_off.p, ds);

    // ---- map write-back (OMU.cpp:281-290) ----
    if (!reserved) {
        // v3: the voxelised reverted bins are written by `tail` extra workgroups of the same launch (from the reverted list)
        const uint32_t tail = (P.version == 3 && n_voi) ? 32u : 0u;
#define ASM_ARGS(FOLDPTR)                                                                                                                   \
    P, h->Tb2o, (const uint8_t *)h->action.p, (const uint32_t *)h->rev_idx.p, sm_keys, (const float4 *)h->spts.p, (const uint32_t *)h->ssrc.p, \
        (const uint32_t *)h->moff.p, (const uint32_t *)Q(h).ccnt.p, (const uint8_t *)h->gflag.p, (const uint32_t *)h->grank.p, h->out_off.p,   \
        h->ground_off.p, h->rej_off.p, ds, Fnew, h->rejected.p, h->rejected_src.p, h->lab_slots.p, tail, (const uint32_t *)h->rev_list.p,      \
        (const uint32_t *)Q(h).qoff.p, (const uint32_t *)h->nvox.p, (const uint32_t *)h->vox_off.p, (const float4 *)h->vox_out.p,              \
        (const uint32_t *)(FOLDPTR ? h->out_off0.p : nullptr), (const uint32_t *)(FOLDPTR ? h->rev_before.p : nullptr), (const uint32_t *)h->ng.p
        const uint32_t asm_grid = std::min<uint32_t>(cdiv(n_voi, 256), 2048) + tail;
        if (n_voi && fold) LAUNCH(h, "assemble", (k_assemble_map<true, true>), asm_grid, 256, ASM_ARGS(true));
        else if (n_voi) LAUNCH(h, "assemble", (k_assemble_map<true, false>), asm_grid, 256, ASM_ARGS(false));
#undef ASM_ARGS
        if (!tail)
            LAUNCH(h, "assemble", k_assemble_bins<true>, B, 256, P, h->Tb2o, (const uint8_t *)h->action.p, (const uint32_t *)h->rev_idx.p,
                   (const uint32_t *)Q(h).qoff.p, (const float4 *)Q(h).sq.p, (const uint32_t *)h->nvox.p, (const uint32_t *)h->vox_off.p,
                   (const float4 *)h->vox_out.p, (const uint32_t *)h->out_off.p, (const uint32_t *)h->crej_off.p, Fnew, h->curr_rejected.p, h->lab_slots.p);
    }
    // (label counters of the new VoI-resident region are accumulated by the two assemble kernels)
    const unsigned long long step_seq = ++h->step_seq;
    {   // the NEXT step's VoI split goes right behind the step when its pose is known (erasor_hip_prefetch_node): the main
        // stream runs it while the host collects this step's results and the caller comes back with the next scan -- the pass
        // reads the store this step has just written and the extents the step's end commits; a step that finds anything else than
        // what was assumed here (pose, store, buffers) simply runs its own pass.
        // Round 4: that launch also ENDS this step (its last workgroup does k_step_end's work)
        // (large-scale mode, round 4: ahead as well, unless the next node's pose moves the submap)
        const bool spec = have_pose && !flags && !submap_would_move(h, nx, ny);
        StepEnd se;
        se.st = ds;
        se.ctr = dc;
        se.out = h->pin;
        se.lab_slots = (const unsigned long long *)h->lab_slots.p;
        se.qctr = (const Counters *)Q(h).d_qctr.p;
        se.q_nvox = (const uint32_t *)Q(h).d_nvox.p;
        se.seq = step_seq;
        const bool end_in_split = spec && h->prof != 1 && !ov_next;
        h->fly.nchunks = nchunks;
        h->fly.spec_launched = ov_next;
        if (ov_next) {
            // round 5: behind the per-bin launch and the late write-back, the late half of the NEXT step's gather -- its last workgroup ends
            // THIS step --, then that step's bucket histogram and column scan: the stream works through the host's turnaround
            const QSide &nq_ = h->q[nxt_side];
            MARK("assemble late");
            // round 6: the late gather joins the early stream behind the next step's split and chunk scan (ev_scan) -- it fills the late
            // table's places in VoI order, which the gather ahead leaves alone, and reads nothing that gather writes -- and runs BESIDE that
            // gather; what needs both, the bucket histogram, joins at ev_early.  On the 9.8 M-point map the early stream's passes are the
            // longer branch in three steps of four: the late gather (8 us + a boundary) no longer follows them (measured, one box, medians of
            // seven passes twice: 0.1903 / 0.1876 -> 0.1885 / 0.1848 ms per scan; 39 M-point map 0.326 / 0.330 -> 0.3155 / 0.3197)
            // (joined at ev_scan in front of the late write-back already)
            MARK("  ev_scan wait");
            LAUNCH(h, "voi_gather", k_late_gather, std::min<uint32_t>(2 * B, 128u) + 1u, 256, (const float4 *)Fnew, (const LateEnt *)late_out, (const DevState *)ds,
                   h->Oxy.p, h->Ozi.p, (const unsigned long long *)h->vmask.p, (const unsigned long long *)h->hmask.p, (const uint32_t *)h->pvl.p,
                   (const uint32_t *)h->phl.p, (const uint32_t *)h->topv.p, (const uint32_t *)h->toph.p, nx, ny, P.voi_r2, to_xf(nq_.To), P, h->alt.d_st.p,
                   h->alt.d_ctr.p, (const Counters *)nq_.d_qctr.p, h->alt.voi_ego.p, h->alt.voi_key.p, h->alt.voi_src.p, se);
            const uint32_t ntile_tab = std::max(1u, cdiv(n_voi, MB_TILE));
            const uint32_t ntile_ub = h->last_n_voi ? std::min(ntile_tab, std::max(64u, cdiv((uint64_t)h->last_n_voi * 3 / 2, MB_TILE))) : ntile_tab;
            const uint32_t *nvoi_next = &h->alt.d_st.p->voi_total;
            (void)hipStreamWaitEvent(h->stream, h->ev_early, 0);  // (the gather ahead)
            LAUNCH(h, "voi_bucket", k_mb_hist, ntile_ub, 1024, (const uint32_t *)h->alt.voi_key.p, n_voi, nvoi_next, B + 2, h->mb_hist.p, h->mb_tot.p);
            LAUNCH(h, "voi_bucket", k_mb_colscan, cdiv(B + 2, MB_PAD), 256, h->mb_hist.p, n_voi, nvoi_next, B + 2, (const uint32_t *)h->mb_tot.p,
                   h->alt.moff.p);
            // ... and the stable scatter, the bins' statistics and the Scan Ratio Test's first pass (into the OTHER set of arrays: a getter may
            // still ask for this step's): what is left for the next step's call is its per-bin launch -- the host's turnaround is off the chain
            h->ov.stats_done = false;
            if (B + 2 <= MBW_NB_MAX) {
                const int bits2 = key_bits(B + 2);
                if (B + 2 <= MBW_NB_SMALL)
                    LAUNCH(h, "voi_bucket", k_mb_scatter_w<MBW_NB_SMALL>, ntile_ub, 1024, (const uint32_t *)h->alt.voi_key.p, (const float4 *)h->alt.voi_ego.p,
                           (const uint32_t *)h->alt.voi_src.p, n_voi, nvoi_next, B + 2, bits2, (const uint32_t *)h->mb_hist.p, h->alt.spts.p, h->alt.ssrc.p,
                           h->alt.rk_a.p, h->dbg_stamps.p);
                else
                    LAUNCH(h, "voi_bucket", k_mb_scatter_w<MBW_NB_MAX>, ntile_ub, 1024, (const uint32_t *)h->alt.voi_key.p, (const float4 *)h->alt.voi_ego.p,
                           (const uint32_t *)h->alt.voi_src.p, n_voi, nvoi_next, B + 2, bits2, (const uint32_t *)h->mb_hist.p, h->alt.spts.p, h->alt.ssrc.p,
                           h->alt.rk_a.p, h->dbg_stamps.p);
                (void)hipStreamWaitEvent(h->stream, nq_.ev_done, 0);  // (the query's bins: its chain was enqueued steps ago)
                LAUNCH(h, "bin_stats", k_bin_stats_srt, cdiv((uint64_t)B * 64, 256), 256, P, (const float4 *)h->alt.spts.p, (const uint32_t *)h->alt.moff.p, B,
                       h->alt.mcnt.p, h->alt.mmin.p, h->alt.mmax.p, (const uint32_t *)nq_.ccnt.p, (const float *)nq_.cmin.p, (const float *)nq_.cmax.p,
                       h->alt.st1b.p);
                h->ov.stats_done = true;
            }
        } else {
            if (reserved) (void)hipStreamWaitEvent(h->stream, h->ev_asm, 0);  // (the early half of the write-back, its label tallies)
            if (!end_in_split)
                LAUNCH(h, "step_end", k_step_end, 1, 1, se.st, se.ctr, se.out, se.lab_slots, se.qctr, se.q_nvox, se.seq);
            if (spec) launch_split_ahead(h, nx, ny, nchunks, end_in_split ? &se : (const StepEnd *)nullptr);
        }
    }
    MARK("late gather + table ahead / end");
    {   // the NEXT scan's query chain goes into its queue now, behind this step's own launches; it runs while we wait
        const int keep_side = h->qi;
        int rc_next = flush_announced(h);
        if (!rc_next) rc_next = flush_held_if_due(h);  // (a chain held back for a shared set of launches whose own step is near)
        h->qi = keep_side;
        if (rc_next) {
            (void)hipStreamSynchronize(h->stream);
            h->spec.valid = false;
            return rc_next;
        }
    }
    h->fly.active = true;
    h->fly.reserved = reserved;
    h->fly.seq = step_seq;
    h->fly.n_map_in = n_map_in;
    h->fly.ns = ns;
    h->fly.sm_keys = sm_keys;
    h->fly.flags = flags;
    // (what a step that has to run again reads: a host scan's staged copy -- float4 rows in the side's pinned buffer --, never the
    // caller's buffer, which is the caller's again once this call has returned)
    h->fly.scan_src = (src_is_device || !n_scan) ? scan_src : (const void *)Q(h).stage;
    h->fly.n_scan = n_scan;
    h->fly.src_is_device = src_is_device;
    memcpy(h->fly.Tl, T_l2b, sizeof(h->fly.Tl));
    memcpy(h->fly.Tb, T_b2o, sizeof(h->fly.Tb));
    memcpy(h->fly.To, T_o2b, sizeof(h->fly.To));
    if (host_timing) {
        for (auto &m : marks) fprintf(stderr, "   %-20s %.1f us\n", m.first, m.second);
        fprintf(stderr, "[step host] enqueue %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count());
    }
    return ERASOR_OK;
}

// (see erasor_hip_handle::OvAuto)
static constexpr int OVA_W = 4, OVA_SKIP = 3;
static constexpr unsigned long long OVA_AGAIN = 600;
static void overlap_auto_sample(erasor_hip_handle *h, double period_us) {
    auto &a = h->ova;
    static const int block_mode[4] = {0, 1, 1, 0};
    if (a.block >= 4) {  // decided: measure again after a while
        if (++a.since_decision >= OVA_AGAIN) {
            a.block = 0;
            a.n = 0;
            a.sum[0] = a.sum[1] = 0;
            a.mx[0] = a.mx[1] = 0;
            a.cnt[0] = a.cnt[1] = 0;
            if (a.mode != block_mode[0]) {
                a.mode = block_mode[0];
                a.skip = OVA_SKIP;
            }
        }
        return;
    }
    if (a.skip > 0) {
        --a.skip;
        return;
    }
    a.sum[a.mode] += period_us;
    a.mx[a.mode] = std::max(a.mx[a.mode], period_us);
    ++a.cnt[a.mode];
    if (++a.n < OVA_W) return;
    a.n = 0;
    ++a.block;
    if (a.block < 4) {
        if (block_mode[a.block] != a.mode) {
            a.mode = block_mode[a.block];
            a.skip = OVA_SKIP;
        }
        return;
    }
    // (means without each mode's largest sample: a page fault or a late host thread in one of eight steps is not the mode's doing)
    a.est[0] = a.cnt[0] > 1 ? (a.sum[0] - a.mx[0]) / (a.cnt[0] - 1) : a.sum[0] / a.cnt[0];
    a.est[1] = a.cnt[1] > 1 ? (a.sum[1] - a.mx[1]) / (a.cnt[1] - 1) : a.sum[1] / a.cnt[1];
    const int m = (a.est[1] * 1.02 < a.est[0]) ? 1 : 0;
    if (m != a.mode) a.skip = OVA_SKIP;
    a.mode = m;
    a.since_decision = 0;
}

// Second half: wait for the step's results (k_step_end's last store into the pinned block is the step's number), commit the host
// mirror of the map store, report.  A VoxelGrid pass-through flip re-runs the step here, synchronously.
static int step_collect(erasor_hip_handle *h, erasor_step_result *res) {
    if (!h) return ERASOR_E_INVALID;
    if (!h->fly.active) {
        h->err = "erasor_hip_step_wait without a step in flight";
        return ERASOR_E_STATE;
    }
    HIPC(h, hipSetDevice(h->device));
    h->fly.active = false;
    const unsigned long long step_seq = h->fly.seq;
    const uint64_t n_map_in = h->fly.n_map_in;
    const uint32_t ns = h->fly.ns;
    const uint32_t *sm_keys = h->fly.sm_keys;
    const int flags = h->fly.flags;
    static const bool host_timing = getenv("ERASOR_HIP_HOST_TIMING") != nullptr;
    const auto t_host1 = std::chrono::steady_clock::now();
    {   // k_step_end's last store into the pinned block is the step's number: poll it (a blocking stream wait wakes up tens
        // of microseconds late, a visible share of a 0.35 ms step), then fall back to the stream wait
        volatile unsigned long long *seq = &h->pin->seq;
        if (!g_debug_sync) {
            const auto t_spin = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (*seq != step_seq) {
                cpu_relax();
                if ((++spins & 0x3FFu) == 0 && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(20)) break;
            }
        }
        if (*seq != step_seq) HIPC(h, hipStreamSynchronize(h->stream));
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    h->st = h->pin->st;
    h->ctr = h->pin->ctr;
    {   // measurement (erasor_hip_chain_timing): the main chain's span on the device's own clock, and how long the stream sat between the
        // previous step's end and this step's chunk scan
        const unsigned long long t_open = h->st.t_open, t_end = h->pin->t_end;
        if (t_end > t_open) {
            h->tm_span += (double)(t_end - t_open) * 0.01;
            if (h->tm_last_end && h->tm_last_seq + 1 == step_seq) {
                if (t_open > h->tm_last_end) {  // (not for an overlapped step: its chunk scan ran beside the step before it)
                    h->tm_gap += (double)(t_open - h->tm_last_end) * 0.01;
                    ++h->tm_ngap;
                }
                if (t_end > h->tm_last_end) {
                    const double per = (double)(t_end - h->tm_last_end) * 0.01;  // end of a step to the end of the next: the period
                    h->tm_period += per;
                    ++h->tm_nper;
                    if (h->fly.deep) overlap_auto_sample(h, per);
                }
            }
            ++h->tm_n;
        }
        h->tm_last_end = t_end;
        h->tm_last_open = t_open;
        h->tm_last_seq = step_seq;
    }
    if (host_timing)
        fprintf(stderr, "[step host] wait %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host1).count());
    {   // diagnostics: the launches' start times on the device's clock, relative to the per-bin launch of this step (see CHAIN_STAMP)
        static const bool stamps = getenv("ERASOR_HIP_CHAIN_STAMPS") != nullptr;
        if (stamps) {
            static bool on = false;
            if (!on) {
                const int one = 1;
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_on), &one, sizeof(one));
                on = true;
            }
            unsigned long long t[32];
            (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stamps), sizeof(t));
            static unsigned long long prev_rev = 0;
            static const char *nm[16] = {"scatter", "bin_stats", "per-bin", "srt4", "asm_early", "split'", "scan'", "gather'", "asm_late", "late_gather'", "hist'",
                                         "colscan'", "o_commit", "", "", "end"};
            fprintf(stderr, "[stamps, us after this step's per-bin launch; period %.1f]", prev_rev ? (double)(t[2] - prev_rev) * 0.01 : 0.0);
            for (int i = 0; i < 16; ++i)
                if (nm[i][0]) fprintf(stderr, " %s %.1f", nm[i], ((double)t[i] - (double)t[2]) * 0.01);
            fprintf(stderr, "\n");
            prev_rev = t[2];
        }
    }
    if (h->dbg_stamps.p) {
        unsigned long long t[96];
        (void)hipMemcpy(t, h->dbg_stamps.p, sizeof(t), hipMemcpyDeviceToHost);
        fprintf(stderr, "[slowest R-GPF bin, 10 ns ticks] key load %llu, exact sort %llu, seeds %llu, staging %llu, it0: cov %llu svd %llu classify %llu (final ground %llu)\n",
                t[32], t[33], t[34], t[35], t[36], t[37], t[38], t[39]);
        fprintf(stderr, "[slowest R-GPF bin, exact sort, shader cycles] start->levels %llu, levels+small %llu, leaf ranking %llu; level starts:", t[41] - t[40],
                t[42] - t[41], t[43] - t[42]);
        for (int i = 44; i < 56; ++i) fprintf(stderr, " %llu", t[i] > t[40] ? t[i] - t[40] : 0ull);
        fprintf(stderr, "\n[slowest R-GPF bin, level-0 partition, cycles] median+first loads %llu, stop lists %llu, search %llu, swaps %llu (m = %llu), queueing %llu; level 0 began %llu before\n",
                t[57] - t[56], t[58] - t[57], t[59] - t[58], t[60] - t[59], t[61], t[62] - t[60], t[56] - t[44]);
        fprintf(stderr, "[esort slowest segment: len %llu depth %llu of %llu segments] phase1 %llu, queue %llu, finalize %llu cycles; levels:", t[30] >> 32,
                t[30] & 0xFFFFFFFFull, t[29], t[1] - t[0], t[2] - t[1], t[3] - t[2]);
        for (int i = 4; i < 15; ++i) fprintf(stderr, " %llu", t[i + 1] > t[i] ? t[i + 1] - t[i] : 0ull);
        fprintf(stderr, "; kernel span: first start -> last working end %.1f us, -> last end %.1f us; long segments left for the global path: %llu (longest %llu)\n",
                (double)(t[26] - t[28]) / 100.0, (double)(t[27] - t[28]) / 100.0, t[25], t[24]);
        fprintf(stderr, "[slowest reverted bin] R-GPF: %llu points, z-sort %.1f us + rest %.1f us; per-bin voxelisation: %llu points -> %llu voxels, %.1f us\n",
                t[19], (double)t[17] / 100.0, (double)t[18] / 100.0, t[21], t[22], (double)t[20] / 100.0);
        fprintf(stderr, "[last per-bin workgroup] %.1f us: open %.1f, select %.1f, R-GPF %.1f (%llu points), voxelisation %.1f (%llu points)\n", (double)t[64] / 100.0,
                (double)t[65] / 100.0, (double)t[66] / 100.0, (double)t[67] / 100.0, t[69], (double)t[68] / 100.0, t[70]);
        fprintf(stderr, "[map scatter, workgroup 3, 10 ns ticks] table + zero %llu, count %llu, prefix %llu, place %llu\n", t[73] - t[72], t[74] - t[73], t[75] - t[74],
                t[76] - t[75]);
        fprintf(stderr, "[reverted bins] %llu, %llu of them beyond the LDS pool, %llu points in all\n", t[23] & 0xFFFFFFFFull, t[23] >> 32, t[63]);
        memset(t, 0, sizeof(t));
        t[28] = ~0ull;
        (void)hipMemcpy(h->dbg_stamps.p, t, sizeof(t), hipMemcpyHostToDevice);
    }
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        h->spec.valid = false;
        h->ov.valid = false;
        h->err = std::string("kernel launch: ") + hipGetErrorString(le);
        return ERASOR_E_NO_DEVICE;
    }
    if (h->ctr.err || h->ctr.sort_qoverflow) {
        h->spec.valid = false;  // (launched ahead on the assumption that this step succeeds)
        if (h->ov.valid) {
            h->ov.valid = false;
            (void)hipStreamSynchronize(h->bstream);
            (void)hipStreamSynchronize(h->stream);
        }
        // errors raised by the query chain (2, 3, 4) are seen by k_voi_gather before it touches the map store: the step left
        // no trace and the host mirror of the state is simply restored.
        h->st.nF = h->nF;
        h->st.nF_valid = h->nFv;
        h->st.o_begin = h->o_begin;
        if (!h->ctr.sort_qoverflow && (h->ctr.err == 2 || h->ctr.err == 4)) {
            // 2: PCL's VoxelGrid would overflow its int32 voxel indices on this scan and pass it through unchanged
            //    (utils.cpp:88-91) -> run the step again with the pass-through chain, and start the next scans that way;
            // 4: the pass-through chain found that VoxelGrid would NOT overflow -> back to the voxelising chain.
            h->q_passthrough = h->ctr.err == 2;
            if (!(flags & STEP_RETRIED)) {
                // chains announced ahead were enqueued in the other mode: they run out, then go again in the new one behind this step's own
                // (step_enqueue, STEP_RETRIED).  The announcements themselves stay: a node announced by ticket has no caller buffer its
                // step could come back with (erasor_hip_step_ticket used to fail with "not the oldest announced scan" here)
                (void)chain_wait(h, -1);
                for (int k = 0; k < h->nqs; ++k) (void)hipStreamSynchronize(h->qstream[k]);
                if (h->cstream) (void)hipStreamSynchronize(h->cstream);
                for (int k = 0; k < NSIDE; ++k) h->q[k].h2d_pending = false;
                float Tl[16], Tb[16], To[16];  // (step_enqueue stores its arguments into h->fly: no aliasing copies)
                memcpy(Tl, h->fly.Tl, sizeof(Tl));
                memcpy(Tb, h->fly.Tb, sizeof(Tb));
                memcpy(To, h->fly.To, sizeof(To));
                const int rc_again = step_enqueue(h, h->fly.scan_src, h->fly.n_scan, h->fly.src_is_device, Tl, Tb, To, flags | STEP_RETRIED);
                return rc_again ? rc_again : step_collect(h, res);
            }
            h->err = "VoxelGrid pass-through decision did not settle";
            return ERASOR_E_INTERNAL;
        }
        if (!h->ctr.sort_qoverflow && h->ctr.err == 3) {
            h->err = "non-finite coordinate (NaN / Inf) in the scan";
            return ERASOR_E_INVALID;
        }
        // everything else was raised AFTER k_voi_gather had tombstoned / moved map entries (a per-bin VoxelGrid overflow in a
        // bin too large for LDS, an exact-sort queue overflow): the map store is no longer the map.  The handle refuses
        // further work until erasor_hip_set_map replaces it.
        h->poisoned = true;
        h->have_map = false;
        h->have_step = false;
        if (h->ctr.sort_qoverflow) {
            h->err = "exact-sort segment queue overflow (site " + std::to_string(h->ctr.sort_qoverflow) + "); the map store is invalid: call erasor_hip_set_map again";
            return ERASOR_E_INTERNAL;
        }
        h->err = "VoxelGrid index overflow in a reverted bin of more than 4096 points (unsupported on device); the map store is invalid: call erasor_hip_set_map again";
        return ERASOR_E_UNSUPPORTED;
    }
    // commit (from here on n_voi / nq are the actual sizes)
    // (round 5: an overlapped step kept VoI-order places for the previous step's late slots; the ones no point came to are not VoI points)
    const uint32_t n_voi_act = h->st.voi_total - h->st.n_voi_dead, nq_act = h->st.q_nvox;
    h->curF ^= 1;
    h->nF = h->st.nF;
    h->nFv = h->st.nF_valid;
    if (h->fly.reserved) {  // (the region just written lies in the reserved layout: its late table is the one that was filled)
        h->late_idx ^= 1;
        h->n_late_F = h->st.n_late;
    } else
        h->n_late_F = 0;
    h->o_begin = h->st.o_begin;
    // (outskirts entries that entered the VoI: none of the reserved places lies in the outskirts' part of VoI order)
    const uint64_t n_out = (uint64_t)(h->o_valid) - (h->st.voi_total - h->st.voiF) + h->st.n_leaving;
    h->o_valid = n_out;
    h->last_n_voi = n_voi_act;
    h->last_nq = nq_act;
    h->last_n_scan = ns;
    h->last_skeys = sm_keys;
    h->have_step = true;
    erasor_step_result r;
    memset(&r, 0, sizeof(r));
    r.n_map_in = n_map_in;
    r.n_voi = n_voi_act;
    r.n_outskirts = n_out;
    r.n_query = nq_act;
    r.n_static_estimate = h->st.n_static_est;
    r.n_complement = h->st.n_compl;
    r.n_map_rejected = h->st.n_rejected;
    r.n_curr_rejected = h->st.n_curr_rejected;
    r.n_ground = h->st.n_ground;
    r.n_map_out = (uint64_t)h->nFv + n_out;
    r.n_static = h->st.F_static + h->st.O_static;
    r.n_dynamic = h->st.F_dynamic + h->st.O_dynamic;
    r.n_reverted_bins = h->st.n_rev;
    r.n_neg_sector = h->ctr.n_neg_sector;
    r.n_ambiguous = h->ctr.n_ambiguous;
    r.n_degenerate_plane = h->ctr.n_degenerate;
    r.n_voxel_overflow = h->ctr.n_voxel_overflow;
    r.n_sort_fallback = h->ctr.n_sort_fallback;
    h->last_res = r;
    if (res) *res = r;
    return ERASOR_OK;
}

static int step_common(erasor_hip_handle *h, const void *scan_src, size_t n_scan, bool src_is_device, const float T_l2b[16],
                       const float T_b2o[16], const float T_o2b[16], erasor_step_result *res, int flags = 0, RowFmt fmt = RowFmt(),
                       uint64_t ticket = 0) {
    const int rc = step_enqueue(h, scan_src, n_scan, src_is_device, T_l2b, T_b2o, T_o2b, flags, fmt, ticket);
    return rc ? rc : step_collect(h, res);
}

// Announce the NEXT scan: its query chain (voxelisation, label search, bucketing -- everything that does not depend on the
// map) is enqueued now and runs beside the map-side stages of the step in flight.  The step call that follows must pass the
// same (pointer, size, T_lidar2body); anything else simply drops the prefetch.
static int prefetch_common(erasor_hip_handle *h, const void *scan_xyzi, size_t n, int src_is_device, const float T_l2b[16], const float *T_b2o,
                           RowFmt fmt = RowFmt(), uint64_t *ticket = nullptr);
int erasor_hip_prefetch_scan(erasor_hip_handle *h, const void *scan_xyzi, size_t n, int src_is_device, const float T_l2b[16]) {
    return prefetch_common(h, scan_xyzi, n, src_is_device, T_l2b, nullptr);
}
// The same with the node's pose (an erasor::node carries both, msg/node.msg:1-7): the step in flight then launches the next
// step's VoI split (fetch_VoI's membership test around that pose, OMU.cpp:246-254) ahead, behind its own last kernel.
int erasor_hip_prefetch_node(erasor_hip_handle *h, const void *scan_xyzi, size_t n, int src_is_device, const float T_l2b[16],
                             const float T_body2origin[16]) {
    if (!T_body2origin) return ERASOR_E_INVALID;
    return prefetch_common(h, scan_xyzi, n, src_is_device, T_l2b, T_body2origin);
}
static int prefetch_common(erasor_hip_handle *h, const void *scan_xyzi, size_t n, int src_is_device, const float T_l2b[16], const float *T_b2o,
                           RowFmt fmt, uint64_t *ticket) {
    // (round 4: allowed while a step is in flight -- between erasor_hip_step_async / _step_ticket_async and erasor_hip_step_wait: the
    // host then stages the next node's cloud while the GPU is busy with the current one; see the end of this function)
    if (!h || !T_l2b || (!scan_xyzi && n) || n > 0x3FFFFFFFull) return ERASOR_E_INVALID;
    if ((src_is_device && (fmt.stride != 16 || fmt.ioff != 12)) || fmt.stride < 16 || fmt.ioff < 12 || fmt.ioff + 4 > fmt.stride) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    if (h->ann.valid && h->npend >= MAX_AHEAD) {
        h->err = "erasor_hip_prefetch_scan: as many scans as there are query sides to hold them are already announced";
        return ERASOR_E_STATE;
    }
    int rc = flush_announced(h);  // an earlier announcement nobody stepped on yet: its chain starts now
    if (rc) return rc;
    const int side = pick_side(h);
    if (h->fly.active && side == h->qi) {
        // every other side holds an announced scan: the one left is the side of the step in flight, whose Scan Ratio Test, R-GPF and
        // write-back are still reading it (and whose staged copy a step that has to run again reads)
        h->err = "erasor_hip_prefetch_*: all query sides are taken while a step is in flight: call erasor_hip_step_wait first";
        return ERASOR_E_STATE;
    }
    if (n && !src_is_device) {  // a host scan is taken over at once
        SideGuard guard(h);
        h->qi = side;
        rc = alloc_scan(h, (uint32_t)n);
        if (rc) return rc;
        rc = chain_wait(h, side);
        if (rc) return rc;
        if (Q(h).used) HIPC(h, hipEventSynchronize(Q(h).ev_done));  // (a dropped chain may still be reading this side's scan)
        rc = stage_host_scan(h, scan_xyzi, (uint32_t)n, copy_stream(h), fmt, nullptr);
        if (rc) return rc;
        h->ann.fp = 0ull;
        h->ann.fp_valid = false;
        Q(h).fp_valid = false;  // (the side's buffer holds a new copy)
        HIPC(h, hipEventRecord(Q(h).ev_h2d, copy_stream(h)));
        Q(h).h2d_pending = true;
    } else {
        h->ann.fp = 0ull;
        h->ann.fp_valid = false;
    }
    h->ann.valid = true;
    h->ann.is_device = src_is_device != 0;
    h->ann.src = scan_xyzi;
    h->ann.n = n;
    h->ann.fmt = fmt;
    h->ann.ticket = h->next_ticket++;
    if (ticket) *ticket = h->ann.ticket;
    h->ann.side = side;
    memcpy(h->ann.Tl, T_l2b, sizeof(h->ann.Tl));
    h->ann.pose_valid = T_b2o != nullptr;
    h->ann.pose_x = T_b2o ? (double)T_b2o[3] : 0.0;  // OMU.cpp:246-247
    h->ann.pose_y = T_b2o ? (double)T_b2o[7] : 0.0;
    h->ann.to_valid = false;
    h->last_ann_side = side;
    if (h->fly.active) {
        // the step in flight has enqueued everything of its own already: the announced chain starts NOW (nothing to go first), and if
        // this is the node right behind that step and its pose is known, its VoI split goes behind the step as well
        const bool next_in_line = h->npend == 0;
        const bool pose = h->ann.pose_valid;
        const double nx = h->ann.pose_x, ny = h->ann.pose_y;
        rc = flush_announced(h);
        if (!rc) rc = flush_held_if_due(h);
        if (rc) return rc;
        if (next_in_line && pose && !h->fly.flags && !h->fly.spec_launched && !h->ov.valid && !submap_would_move(h, nx, ny)) {
            hipStream_t keep = h->cur;
            h->cur = h->stream;
            launch_split_ahead(h, nx, ny, h->fly.nchunks, nullptr);
            h->cur = keep;
        }
    }
    return ERASOR_OK;
}

// Round 5: the announced node's OTHER transform.  fetch_VoI's egocentric transform uses tf_body2origin_.inverse() (OMU.cpp:436), which the
// caller computes and hands to the step (see erasor_hip_step); announced together with the node (right after erasor_hip_prefetch_node*)
// it lets that step's split, chunk scan AND gather run beside the per-bin launch of the step in front of it (OVERLAPPED steps).  The
// step must then pass the very same 16 floats; anything else and it simply runs its own passes.
int erasor_hip_announce_origin2body(erasor_hip_handle *h, const float T_origin2body[16]) {
    if (!h || !T_origin2body) return ERASOR_E_INVALID;
    if (h->ann.valid && h->ann.pose_valid) {
        memcpy(h->ann.To, T_origin2body, sizeof(h->ann.To));
        h->ann.to_valid = true;
        return ERASOR_OK;
    }
    const int sd = h->last_ann_side;
    bool pending = false;
    for (int j = 0; j < h->npend; ++j) pending = pending || h->pend[j] == sd;
    if (sd >= 0 && pending && h->q[sd].pose_valid) {  // (announced while a step was in flight: its chain is in its queue already)
        memcpy(h->q[sd].To, T_origin2body, sizeof(h->q[sd].To));
        h->q[sd].to_valid = true;
        return ERASOR_OK;
    }
    h->err = "erasor_hip_announce_origin2body: no node announced with its pose to attach the transform to";
    return ERASOR_E_STATE;
}

int erasor_hip_step(erasor_hip_handle *h, const float *scan_xyzi, size_t n_scan, const float T_lidar2body[16], const float T_body2origin[16],
                    const float T_origin2body[16], erasor_step_result *res) {
    return step_common(h, scan_xyzi, n_scan, false, T_lidar2body, T_body2origin, T_origin2body, res);
}
int erasor_hip_step_device(erasor_hip_handle *h, const void *d_scan_xyzi, size_t n_scan, const float T_lidar2body[16],
                           const float T_body2origin[16], const float T_origin2body[16], erasor_step_result *res) {
    return step_common(h, d_scan_xyzi, n_scan, true, T_lidar2body, T_body2origin, T_origin2body, res);
}
// Host records in the caller's own layout (pcl::PointXYZI: stride 32, intensity at byte 16 -- what pcl::fromROSMsg leaves, OMU.cpp:237):
// the one pass that stages the scan in pinned memory repacks it, so a caller no longer rewrites 4 MB per node into XYZI rows first.
static RowFmt row_fmt(size_t stride, size_t ioff) {
    RowFmt f;
    f.stride = stride > 0xFFFFFFFFull ? 0u : (uint32_t)stride;  // (0: rejected as invalid further down)
    f.ioff = ioff > 0xFFFFFFFFull ? 0u : (uint32_t)ioff;
    return f;
}
int erasor_hip_step_rows(erasor_hip_handle *h, const void *rows, size_t n, size_t stride_bytes, size_t intensity_offset_bytes,
                         const float T_lidar2body[16], const float T_body2origin[16], const float T_origin2body[16], erasor_step_result *res) {
    return step_common(h, rows, n, false, T_lidar2body, T_body2origin, T_origin2body, res, 0, row_fmt(stride_bytes, intensity_offset_bytes));
}
int erasor_hip_prefetch_node_rows(erasor_hip_handle *h, const void *rows, size_t n, size_t stride_bytes, size_t intensity_offset_bytes,
                                  const float T_lidar2body[16], const float *T_body2origin, uint64_t *ticket) {
    return prefetch_common(h, rows, n, 0, T_lidar2body, T_body2origin, row_fmt(stride_bytes, intensity_offset_bytes), ticket);
}
// The step of an ANNOUNCED node by the ticket its announcement returned: nothing is recognised by pointer or content, the caller's
// buffer is not touched again (it was the caller's again when the announcement returned).  Tickets are consumed in announcement order.
int erasor_hip_step_ticket(erasor_hip_handle *h, uint64_t ticket, const float T_body2origin[16], const float T_origin2body[16],
                           erasor_step_result *res) {
    if (!ticket) return ERASOR_E_INVALID;
    return step_common(h, nullptr, 0, false, nullptr, T_body2origin, T_origin2body, res, 0, RowFmt(), ticket);
}
int erasor_hip_step_ticket_async(erasor_hip_handle *h, uint64_t ticket, const float T_body2origin[16], const float T_origin2body[16]) {
    if (!ticket) return ERASOR_E_INVALID;
    return step_enqueue(h, nullptr, 0, false, nullptr, T_body2origin, T_origin2body, 0, RowFmt(), ticket);
}
// SURVEY 8(b), threading row: the step in two halves, so that ONE host thread can keep several handles (independent sequences,
// OMU.cpp:203 is one callback per node per updater) busy: async enqueues everything and returns, wait collects.
int erasor_hip_step_async(erasor_hip_handle *h, const void *scan_xyzi, size_t n_scan, int src_is_device, const float T_lidar2body[16],
                          const float T_body2origin[16], const float T_origin2body[16]) {
    return step_enqueue(h, scan_xyzi, n_scan, src_is_device != 0, T_lidar2body, T_body2origin, T_origin2body);
}
int erasor_hip_step_wait(erasor_hip_handle *h, erasor_step_result *res) { return step_collect(h, res); }
// The node loop of an offline driver in ONE call (main_in_your_env.cpp:92-123 is that loop in the reference): nodes [first, first + count)
// of a sequence of n_total, each announced `lookahead` nodes ahead (erasor_hip_prefetch_node), stepped one after the other.
// *announced (in / out): nodes [0, *announced) have been announced already -- lets a caller split a sequence over several calls
// (warm-up, timed part) without breaking the pipeline.  res: `count` result blocks.  Exactly the calls a host loop would make;
// what it saves is the caller's own time between two steps (a Python loop: ~20 us of a 0.26 ms step).
int erasor_hip_run_nodes(erasor_hip_handle *h, const void *const *scans, const size_t *n_pts, size_t n_total, int src_is_device,
                         const float T_lidar2body[16], const float *T_body2origin /* n_total x 16 */, const float *T_origin2body /* n_total x 16 */,
                         size_t first, size_t count, int lookahead, size_t *announced, erasor_step_result *res) {
    if (!h || !scans || !n_pts || !T_lidar2body || !T_body2origin || !T_origin2body || !announced || first + count > n_total || lookahead < 0 || lookahead > MAX_AHEAD)
        return ERASOR_E_INVALID;
    NOFLY(h);
    for (size_t i = first; i < first + count; ++i) {
        const size_t want = std::min(n_total, i + (size_t)lookahead + 1);
        while (*announced < want) {
            const size_t j = *announced;
            if (j >= i && lookahead > 0) {
                int rc = erasor_hip_prefetch_node(h, scans[j], n_pts[j], src_is_device, T_lidar2body, T_body2origin + 16 * j);
                if (!rc) rc = erasor_hip_announce_origin2body(h, T_origin2body + 16 * j);
                if (rc) return rc;
            }
            ++*announced;
        }
        const int rc = step_common(h, scans[i], n_pts[i], src_is_device != 0, T_lidar2body, T_body2origin + 16 * i, T_origin2body + 16 * i, res ? res + (i - first) : nullptr);
        if (rc) return rc;
    }
    return ERASOR_OK;
}
// ---- one host process, several devices (north_star: "RCCL broadcast of the global map over xGMI") --------------
// The map of handles[root] is assembled into one dense XYZI array on its device and sent to every other handle's device:
//   * RCCL, single process: ncclCommInitAll over the handles' devices + ONE ncclBroadcast per rank inside a group (float32, 4 n values) --
//     the library is loaded lazily (dlopen librccl.so: a process that never replicates never pays for it);
//   * otherwise (RCCL missing, two handles on one device -- which ncclCommInitAll refuses --, ERASOR_HIP_NO_RCCL=1): hipMemcpyPeerAsync.
// Every receiver then takes the copy with erasor_hip_set_map_device's code.  *transport: 1 RCCL, 2 peer copies, 0 nothing to send.
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl *rccl_api() {
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    if (getenv("ERASOR_HIP_NO_RCCL")) return &r;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return &r;
    r.CommInitAll = (int (*)(void **, int, const int *))dlsym(r.lib, "ncclCommInitAll");
    r.CommDestroy = (int (*)(void *))dlsym(r.lib, "ncclCommDestroy");
    r.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(r.lib, "ncclBroadcast");
    r.GroupStart = (int (*)())dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(r.lib, "ncclGroupEnd");
    r.GetErrorString = (const char *(*)(int))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.CommInitAll && r.CommDestroy && r.Broadcast && r.GroupStart && r.GroupEnd;
    return &r;
}
// the handle's whole map as ONE dense device array, in the reference's order: [VoI-resident part | outskirts | submap complement]
int map_to_device(erasor_hip_handle *h, DBuf<float4> &dense, size_t *n_out) {
    const size_t total = (size_t)h->nFv + (size_t)h->o_valid + (size_t)h->nC;
    *n_out = total;
    if (ensure(h, dense, total + 8)) return ERASOR_E_NO_DEVICE;
    struct Restore {  // (also on the error returns below: the stream LAUNCH() targets, the temporaries)
        erasor_hip_handle *h;
        hipStream_t keep;
        DBuf<uint32_t> flag, pl, tops;
        ~Restore() {
            release(flag);
            release(pl);
            release(tops);
            h->cur = keep;
        }
    } rs{h, h->cur, {}, {}, {}};
    DBuf<uint32_t> &flag = rs.flag, &pl = rs.pl, &tops = rs.tops;
    h->cur = h->stream;
    {
        const int rc_f = copy_F_dense(h, dense.p);
        if (rc_f) return rc_f;
    }
    const uint32_t span = h->capO - h->o_begin;
    if (span && h->o_valid) {
        if (ensure(h, flag, span + 1) || ensure(h, pl, span + 1) || ensure(h, tops, span / 1024 + 4)) return ERASOR_E_NO_DEVICE;
        LAUNCH(h, "replicate", k_o_valid, cdiv(span, 256), 256, (const float2 *)h->Oxy.p, h->o_begin, h->capO, flag.p);
        scan_u32(h, flag.p, pl.p, tops.p, span, span, nullptr, nullptr, "replicate");
        LAUNCH(h, "replicate", k_o_compact, cdiv(span, 256), 256, (const float2 *)h->Oxy.p, (const float2 *)h->Ozi.p, h->o_begin, h->capO,
               (const uint32_t *)flag.p, (const uint32_t *)pl.p, (const uint32_t *)tops.p, dense.p + h->nFv);
        HIPC(h, hipStreamSynchronize(h->stream));
    }
    if (h->nC)
        HIPC(h, hipMemcpyAsync(dense.p + (size_t)h->nFv + h->o_valid, h->Cbuf.p, (size_t)h->nC * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return ERASOR_OK;
}
}  // namespace
int erasor_hip_replicate_map(erasor_hip_handle *const *handles, int n, int root, int *transport) {
    if (transport) *transport = 0;
    if (!handles || n < 1 || root < 0 || root >= n) return ERASOR_E_INVALID;
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return ERASOR_E_INVALID;
        NOFLY(handles[i]);
    }
    erasor_hip_handle *hr = handles[root];
    if (!hr->have_map) {
        hr->err = "erasor_hip_replicate_map: the root handle has no map (erasor_hip_set_map first)";
        return ERASOR_E_STATE;
    }
    HIPC(hr, hipSetDevice(hr->device));
    DBuf<float4> dense;
    size_t total = 0;
    int rc = map_to_device(hr, dense, &total);
    if (rc) return rc;
    // receive buffers, one per other handle, on its device
    std::vector<float4 *> recv(n, nullptr);
    bool distinct = true;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) distinct = distinct && handles[i]->device != handles[j]->device;
    auto cleanup = [&]() {
        for (int i = 0; i < n; ++i)
            if (recv[i]) {
                (void)hipSetDevice(handles[i]->device);
                (void)hipFree(recv[i]);
            }
        (void)hipSetDevice(hr->device);
        release(dense);
    };
    for (int i = 0; i < n; ++i) {
        if (i == root) continue;
        if (hipSetDevice(handles[i]->device) != hipSuccess || hipMalloc((void **)&recv[i], (total + 8) * sizeof(float4)) != hipSuccess) {
            hr->err = "erasor_hip_replicate_map: no memory for the copy on device " + std::to_string(handles[i]->device);
            cleanup();
            return ERASOR_E_NO_DEVICE;
        }
    }
    int used = 0;
    Rccl &R = *rccl_api();
    if (R.ok && distinct && total) {
        // single-process RCCL: one communicator per device, the broadcasts of all ranks in ONE group
        std::vector<void *> comms(n, nullptr);
        std::vector<int> devs(n);
        for (int i = 0; i < n; ++i) devs[i] = handles[i]->device;
        int e = R.CommInitAll(comms.data(), n, devs.data());
        if (e == 0) {
            e = R.GroupStart();
            for (int i = 0; e == 0 && i < n; ++i) {
                (void)hipSetDevice(handles[i]->device);
                void *buf = i == root ? (void *)dense.p : (void *)recv[i];
                e = R.Broadcast(buf, buf, total * 4, /*ncclFloat*/ 7, root, comms[i], handles[i]->stream);
            }
            const int e2 = R.GroupEnd();
            if (e == 0) e = e2;
            for (int i = 0; i < n; ++i) {
                (void)hipSetDevice(handles[i]->device);
                (void)hipStreamSynchronize(handles[i]->stream);
            }
            for (int i = 0; i < n; ++i)
                if (comms[i]) (void)R.CommDestroy(comms[i]);
            if (e == 0) used = 1;
        }
        if (e != 0) fprintf(stderr, "[erasor_hip] RCCL broadcast failed (%s): falling back to peer copies\n", R.GetErrorString ? R.GetErrorString(e) : "?");
    }
    if (!used && total) {
        for (int i = 0; i < n; ++i) {
            if (i == root) continue;
            (void)hipSetDevice(hr->device);
            if (handles[i]->device != hr->device) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, hr->device, handles[i]->device);
                if (can) (void)hipDeviceEnablePeerAccess(handles[i]->device, 0);  // (already enabled: an error we ignore)
                (void)hipGetLastError();
            }
            if (hipMemcpyPeerAsync(recv[i], handles[i]->device, dense.p, hr->device, total * sizeof(float4), hr->stream) != hipSuccess) {
                hr->err = "erasor_hip_replicate_map: hipMemcpyPeerAsync failed";
                cleanup();
                return ERASOR_E_NO_DEVICE;
            }
        }
        (void)hipStreamSynchronize(hr->stream);
        used = 2;
    }
    for (int i = 0; i < n && rc == ERASOR_OK; ++i) {
        if (i == root) continue;
        rc = set_map_common(handles[i], recv[i], total, true);
    }
    cleanup();
    if (transport) *transport = total ? used : 0;
    return rc;
}

int erasor_hip_step_done(erasor_hip_handle *h) {
    if (!h || !h->fly.active) return 1;
    return *(volatile unsigned long long *)&h->pin->seq == h->fly.seq ? 1 : 0;
}

int erasor_hip_map_size(erasor_hip_handle *h, size_t *n) {
    NOFLY(h);
    if (!h || !n) return ERASOR_E_INVALID;
    if (!h->have_map) return ERASOR_E_STATE;
    *n = (size_t)h->nFv + (size_t)h->o_valid + (size_t)h->nC;  // large-scale: *map_arranged_ + *map_arranged_complement_ (OMU.cpp:181)
    return ERASOR_OK;
}

// copies a device float4 range to a caller buffer with the (cap, *n) convention
// read-back of a step's products: on the MAIN stream (everything a step produces is ordered there) and waiting for that stream only
// -- a blocking hipMemcpy on the null stream also waited for the query chains of the scans announced ahead (0.45 ms each)
static int d2h(erasor_hip_handle *h, void *dst, const void *src, size_t bytes) {
    if (!bytes) return ERASOR_OK;
    HIPC(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return ERASOR_OK;
}
// A read-back of something a COLLECTED step left behind and nothing enqueued since overwrites (map_rejected / curr_rejected: written by the
// step's write-back, next by the following step's): on the copy stream, without waiting for the main stream -- which, when the step's
// end is seen, already holds the launches made ahead for the NEXT step (bucket histogram .. bin statistics, ~50 us).  The step's end is
// written by the last kernel of the step, behind the kernels that wrote these clouds: what the host has seen end is in device memory.
static int d2h_collected(erasor_hip_handle *h, void *dst, const void *src, size_t bytes) {
    if (!bytes) return ERASOR_OK;
    hipStream_t cs = copy_stream(h);
    HIPC(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cs));
    HIPC(h, hipStreamSynchronize(cs));
    return ERASOR_OK;
}
static int out_cloud_collected(erasor_hip_handle *h, const float4 *d, size_t cnt, float *dst, size_t cap, size_t *n) {
    if (n) *n = cnt;
    if (!dst) return ERASOR_OK;
    if (cnt > cap) return ERASOR_E_CAPACITY;
    return d2h_collected(h, dst, d, cnt * sizeof(float4));
}
static int out_cloud(erasor_hip_handle *h, const float4 *d, size_t cnt, float *dst, size_t cap, size_t *n) {
    if (n) *n = cnt;
    if (!dst) return ERASOR_OK;
    if (cnt > cap) return ERASOR_E_CAPACITY;
    return d2h(h, dst, d, cnt * sizeof(float4));
}

int erasor_hip_get_map(erasor_hip_handle *h, float *dst, size_t cap, size_t *n) {
    NOFLY(h);
    if (!h) return ERASOR_E_INVALID;
    if (!h->have_map) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    const size_t total = (size_t)h->nFv + (size_t)h->o_valid + (size_t)h->nC;
    if (n) *n = total;
    if (!dst) return ERASOR_OK;
    if (total > cap) return ERASOR_E_CAPACITY;
    if (h->nC) HIPC(h, hipMemcpy(dst + ((size_t)h->nFv + h->o_valid) * 4, h->Cbuf.p, (size_t)h->nC * sizeof(float4), hipMemcpyDeviceToHost));
    if (h->nF && h->nFv == h->nF) HIPC(h, hipMemcpy(dst, h->F[h->curF].p, (size_t)h->nF * sizeof(float4), hipMemcpyDeviceToHost));
    else if (h->nF) {  // (round 5: the region holds reserved slots nothing filled: its points, compacted)
        DBuf<float4> tmpF;
        if (ensure(h, tmpF, (size_t)h->nFv + 1)) return ERASOR_E_NO_DEVICE;
        const int rc_f = copy_F_dense(h, tmpF.p);
        if (!rc_f && h->nFv) (void)hipMemcpy(dst, tmpF.p, (size_t)h->nFv * sizeof(float4), hipMemcpyDeviceToHost);
        release(tmpF);
        if (rc_f) return rc_f;
    }
    const uint32_t span = h->capO - h->o_begin;
    if (span && h->o_valid) {
        DBuf<uint32_t> flag, pl, tops;
        DBuf<float4> tmp;
        if (ensure(h, flag, span + 1) || ensure(h, pl, span + 1) || ensure(h, tops, span / 1024 + 4) || ensure(h, tmp, h->o_valid + 1))
            return ERASOR_E_NO_DEVICE;
        LAUNCH(h, "get_map", k_o_valid, cdiv(span, 256), 256, (const float2 *)h->Oxy.p, h->o_begin, h->capO, flag.p);
        scan_u32(h, flag.p, pl.p, tops.p, span, span, nullptr, nullptr, "get_map");
        LAUNCH(h, "get_map", k_o_compact, cdiv(span, 256), 256, (const float2 *)h->Oxy.p, (const float2 *)h->Ozi.p, h->o_begin, h->capO,
               (const uint32_t *)flag.p, (const uint32_t *)pl.p, (const uint32_t *)tops.p, tmp.p);
        HIPC(h, hipStreamSynchronize(h->stream));
        HIPC(h, hipMemcpy(dst + (size_t)h->nFv * 4, tmp.p, (size_t)h->o_valid * sizeof(float4), hipMemcpyDeviceToHost));
        release(flag);
        release(pl);
        release(tops);
        release(tmp);
    }
    return ERASOR_OK;
}

// the last step's [selected bins theta-major | ground_viz | complement] WITHOUT tf_body2origin_ (the egocentric clouds
// ERASOR::get_static_estimate hands out, erasor.cpp:612-626), assembled into the retired F buffer (free until the next step)
static int assemble_egocentric(erasor_hip_handle *h, float4 **out) {
    const DevState &s = h->st;
    if (ensure(h, h->F[h->curF ^ 1], (size_t)s.nF_new + 8)) return ERASOR_E_NO_DEVICE;
    // the kernel reads the LAST step's state (sizes of its VoI and of its output): from a copy of the host mirror, not from the live
    // device state -- the next step's chunk scan may have opened that already (launched ahead, round 4)
    if (ensure(h, h->d_st_get, 1)) return ERASOR_E_NO_DEVICE;
    HIPC(h, hipMemcpyAsync(h->d_st_get.p, &h->st, sizeof(DevState), hipMemcpyHostToDevice, h->stream));
    float4 *tmp = h->F[h->curF ^ 1].p;
    const DP &P = h->dp;
    const uint32_t B = h->B, n_voi = h->last_n_voi;
    if (n_voi)
        LAUNCH(h, "get_cloud", (k_assemble_map<false, false>), std::min<uint32_t>(cdiv(n_voi, 256), 2048), 256, P, h->Tb2o, (const uint8_t *)h->action.p,
               (const uint32_t *)h->rev_idx.p, h->last_skeys, (const float4 *)h->spts.p, (const uint32_t *)h->ssrc.p,
               (const uint32_t *)h->moff.p, (const uint32_t *)Q(h).ccnt.p, (const uint8_t *)h->gflag.p, (const uint32_t *)h->grank.p,
               h->out_off.p, h->ground_off.p, h->rej_off.p,
               h->d_st_get.p, tmp, (float4 *)nullptr, (uint32_t *)nullptr, (unsigned long long *)nullptr, 0u, (const uint32_t *)nullptr,
               (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const float4 *)nullptr, (const uint32_t *)nullptr,
               (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    LAUNCH(h, "get_cloud", k_assemble_bins<false>, B, 256, P, h->Tb2o, (const uint8_t *)h->action.p, (const uint32_t *)h->rev_idx.p,
           (const uint32_t *)Q(h).qoff.p, (const float4 *)Q(h).sq.p, (const uint32_t *)h->nvox.p, (const uint32_t *)h->vox_off.p,
           (const float4 *)h->vox_out.p, (const uint32_t *)h->out_off.p, (const uint32_t *)h->crej_off.p, tmp, (float4 *)nullptr, (unsigned long long *)nullptr);
    HIPC(h, hipStreamSynchronize(h->stream));
    *out = tmp;
    return ERASOR_OK;
}

int erasor_hip_get_cloud(erasor_hip_handle *h, int which, float *dst, size_t cap, size_t *n) {
    NOFLY(h);
    if (!h) return ERASOR_E_INVALID;
    if (which == ERASOR_CLOUD_MAP) return erasor_hip_get_map(h, dst, cap, n);
    if (!h->have_step) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    const DevState &s = h->st;
    // (round 6: the two clouds a callback publishes every node, OMU.cpp:316-320, do not wait for what is launched ahead of the next step)
    if (which == ERASOR_CLOUD_MAP_REJECTED) return out_cloud_collected(h, h->rejected.p, s.n_rejected, dst, cap, n);
    if (which == ERASOR_CLOUD_CURR_REJECTED) return out_cloud_collected(h, h->curr_rejected.p, s.n_curr_rejected, dst, cap, n);
    HIPC(h, hipStreamSynchronize(h->stream));
    switch (which) {
        case ERASOR_CLOUD_QUERY_VOI: return out_cloud(h, Q(h).query.p, h->last_nq, dst, cap, n);
        case ERASOR_CLOUD_MAP_VOI: {
            if (!s.n_voi_dead) return out_cloud(h, h->voi_ego.p, h->last_n_voi, dst, cap, n);
            // (round 5: an overlapped step's VoI-order array keeps places no point came to, key B + 1: the VoI is what is left)
            if (n) *n = h->last_n_voi;
            if (!dst) return ERASOR_OK;
            if (h->last_n_voi > cap) return ERASOR_E_CAPACITY;
            const uint32_t tot = s.voi_total;
            DBuf<uint32_t> flag, pl, tops;
            DBuf<float4> tmp;
            if (ensure(h, flag, (size_t)tot + 1) || ensure(h, pl, (size_t)tot + 1) || ensure(h, tops, tot / 1024 + 4) || ensure(h, tmp, (size_t)h->last_n_voi + 1))
                return ERASOR_E_NO_DEVICE;
            hipStream_t keep = h->cur;
            h->cur = h->stream;
            LAUNCH(h, "get_cloud", k_voi_live, cdiv(tot, 256), 256, (const uint32_t *)h->voi_key.p, tot, h->B, flag.p);
            scan_u32(h, flag.p, pl.p, tops.p, tot, tot, nullptr, nullptr, "get_cloud");
            LAUNCH(h, "get_cloud", k_f_compact, cdiv(tot, 256), 256, (const float4 *)h->voi_ego.p, tot, (const uint32_t *)flag.p, (const uint32_t *)pl.p,
                   (const uint32_t *)tops.p, tmp.p);
            h->cur = keep;
            const int rc_c = d2h(h, dst, tmp.p, (size_t)h->last_n_voi * sizeof(float4));
            release(flag);
            release(pl);
            release(tops);
            release(tmp);
            return rc_c;
        }
        case ERASOR_CLOUD_STATIC_ESTIMATE:
        case ERASOR_CLOUD_COMPLEMENT:
        case ERASOR_CLOUD_GROUND_VIZ: {
            // re-run the assembly without tf_body2origin_ into the retired F buffer (free until the next step)
            size_t cnt = which == ERASOR_CLOUD_STATIC_ESTIMATE ? s.n_static_est : (which == ERASOR_CLOUD_COMPLEMENT ? s.n_compl : s.n_ground);
            if (n) *n = cnt;
            if (!dst) return ERASOR_OK;
            if (cnt > cap) return ERASOR_E_CAPACITY;
            float4 *tmp = nullptr;
            const int rc_a = assemble_egocentric(h, &tmp);
            if (rc_a) return rc_a;
            const size_t off = which == ERASOR_CLOUD_STATIC_ESTIMATE ? 0 : (which == ERASOR_CLOUD_COMPLEMENT ? s.n_static_est : s.total_bins);
            if (cnt && d2h(h, dst, tmp + off, cnt * sizeof(float4))) return ERASOR_E_NO_DEVICE;
            return ERASOR_OK;
        }
    }
    return ERASOR_E_INVALID;
}

int erasor_hip_get_rejected_indices(erasor_hip_handle *h, uint64_t *dst, size_t cap, size_t *n) {
    NOFLY(h);
    if (!h) return ERASOR_E_INVALID;
    if (!h->have_step) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    const size_t cnt = h->st.n_rejected;
    if (n) *n = cnt;
    if (!dst) return ERASOR_OK;
    if (cnt > cap) return ERASOR_E_CAPACITY;
    std::vector<uint32_t> tmp(cnt);
    if (cnt && d2h(h, tmp.data(), h->rejected_src.p, cnt * 4)) return ERASOR_E_NO_DEVICE;
    for (size_t k = 0; k < cnt; ++k) dst[k] = tmp[k];
    return ERASOR_OK;
}

int erasor_hip_get_bins(erasor_hip_handle *h, int which, uint32_t *count, double *min_h, double *max_h) {
    NOFLY(h);
    if (!h || !count || !min_h || !max_h || (which != 0 && which != 1)) return ERASOR_E_INVALID;
    if (!h->have_step) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    const uint32_t B = h->B, R = h->P.num_rings, S = h->P.num_sectors;
    std::vector<uint32_t> c(B);
    std::vector<float> mn(B), mx(B);
    HIPC(h, hipMemcpy(c.data(), which ? Q(h).ccnt.p : h->mcnt.p, B * 4, hipMemcpyDeviceToHost));
    HIPC(h, hipMemcpy(mn.data(), which ? Q(h).cmin.p : h->mmin.p, B * 4, hipMemcpyDeviceToHost));
    HIPC(h, hipMemcpy(mx.data(), which ? Q(h).cmax.p : h->mmax.p, B * 4, hipMemcpyDeviceToHost));
    for (uint32_t key = 0; key < B; ++key) {  // device key = sector*R + ring -> API index = ring*S + sector
        const uint32_t ring = key % R, sector = key / R;
        const uint32_t o = ring * S + sector;
        count[o] = c[key];
        min_h[o] = c[key] ? (double)mn[key] : INF_H;
        max_h[o] = c[key] ? (double)mx[key] : -INF_H;
    }
    return ERASOR_OK;
}

int erasor_hip_get_status(erasor_hip_handle *h, double *status) {
    NOFLY(h);
    if (!h || !status) return ERASOR_E_INVALID;
    if (!h->have_step) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    const uint32_t B = h->B, R = h->P.num_rings, S = h->P.num_sectors;
    std::vector<uint8_t> s(B);
    HIPC(h, hipMemcpy(s.data(), h->status.p, B, hipMemcpyDeviceToHost));
    static const double tab[6] = {ERASOR_ST_LITTLE_NUM, ERASOR_ST_MERGE_BINS, ERASOR_ST_MAP_IS_HIGHER, ERASOR_ST_BLOCKED, ERASOR_ST_CURR_IS_HIGHER,
                                  ERASOR_ST_LITTLE_NUM};
    for (uint32_t key = 0; key < B; ++key) status[(key % R) * S + key / R] = tab[s[key] < 6 ? s[key] : 0];
    return ERASOR_OK;
}

// The point lists of an R-POD of the last step (erasor.h:143-145), egocentric, theta-major (the order r_pod2pc walks,
// erasor.cpp:309-320): which 0 = r_pod_map, 1 = r_pod_curr, 2 = r_pod_selected.  begin / count: per bin, index ring*S+sector.
int erasor_hip_get_rpod(erasor_hip_handle *h, int which, float *xyzi, size_t cap, size_t *n, uint32_t *begin, uint32_t *count) {
    NOFLY(h);
    if (!h || which < 0 || which > 2) return ERASOR_E_INVALID;
    if (!h->have_step) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    HIPC(h, hipStreamSynchronize(h->stream));
    const uint32_t B = h->B, R = h->P.num_rings, S = h->P.num_sectors;
    std::vector<uint32_t> off(B + 2, 0u);
    const float4 *src = nullptr;
    size_t total = 0;
    if (which == 0) {
        HIPC(h, hipMemcpy(off.data(), h->moff.p, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost));
        src = h->spts.p;
        total = off[B];
    } else if (which == 1) {
        HIPC(h, hipMemcpy(off.data(), Q(h).qoff.p, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost));
        src = Q(h).sq.p;
        total = off[B];
    } else {
        HIPC(h, hipMemcpy(off.data(), h->out_off.p, (size_t)B * 4, hipMemcpyDeviceToHost));
        off[B] = h->st.total_bins;
        total = h->st.total_bins;
    }
    if (n) *n = total;
    if (begin && count)
        for (uint32_t key = 0; key < B; ++key) {  // device key = sector*R + ring -> API index = ring*S + sector
            const uint32_t o = (key % R) * S + key / R;
            begin[o] = off[key];
            count[o] = off[key + 1] - off[key];
        }
    if (!xyzi) return ERASOR_OK;
    if (total > cap) return ERASOR_E_CAPACITY;
    if (which == 2) {
        float4 *tmp = nullptr;
        const int rc = assemble_egocentric(h, &tmp);
        if (rc) return rc;
        src = tmp;
    }
    if (total) HIPC(h, hipMemcpy(xyzi, src, total * sizeof(float4), hipMemcpyDeviceToHost));
    return ERASOR_OK;
}

int erasor_hip_get_planes(erasor_hip_handle *h, uint32_t *bin_index, float *normal, double *d, size_t cap_bins, size_t *n_bins) {
    NOFLY(h);
    if (!h) return ERASOR_E_INVALID;
    if (!h->have_step) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    const size_t nb = h->st.n_rev;
    if (n_bins) *n_bins = nb;
    if (!bin_index || !normal || !d) return ERASOR_OK;
    if (nb > cap_bins) return ERASOR_E_CAPACITY;
    const uint32_t R = h->P.num_rings, S = h->P.num_sectors;
    const size_t it = h->P.gf_iter;
    std::vector<uint32_t> keys(nb);
    if (nb) {
        HIPC(h, hipMemcpy(keys.data(), h->rev_list.p, nb * 4, hipMemcpyDeviceToHost));
        HIPC(h, hipMemcpy(normal, h->plane_n.p, nb * it * 3 * sizeof(float), hipMemcpyDeviceToHost));
        HIPC(h, hipMemcpy(d, h->plane_d.p, nb * it * sizeof(double), hipMemcpyDeviceToHost));
    }
    for (size_t k = 0; k < nb; ++k) bin_index[k] = (keys[k] % R) * S + keys[k] / R;
    return ERASOR_OK;
}

// erasor_utils::voxelize_preserving_labels (utils.cpp:80-114) of a DEVICE cloud; the result is Q(h).query[0..*nq_out).
// d_src may be any device buffer except the scan-side scratch itself.
static int voxelize_device(erasor_hip_handle *h, const float4 *d_src, uint32_t ns, double leaf_size, uint32_t *nq_out) {
    q_drain(h);
    // this borrows the query side of the last finished step: its QUERY_VOI / STATIC_ESTIMATE / ... read-backs are gone
    // (erasor_hip_get_cloud answers ERASOR_E_STATE until the next step instead of handing out clobbered buffers)
    h->have_step = false;
    int rc = alloc_scan(h, std::max(ns, 1u));
    if (rc) return rc;
    Q(h).scan_in = d_src;
    LAUNCH(h, "q_begin", k_query_begin, 1, 256, Q(h).d_qctr.p, Q(h).bb.p, (uint32_t *)nullptr, 0u, Q(h).d_nvox.p, 0u);
    voxelize_query_part1(h, ns, (float)leaf_size, [] {});
    uint32_t nvox_host = 0;
    VoxGrid g;
    HIPC(h, hipMemcpyAsync(&nvox_host, Q(h).d_nvox.p, sizeof(nvox_host), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipMemcpyAsync(&g, Q(h).qgrid.p, sizeof(g), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    if (ns && g.overflow) {
        if (g.overflow == 2) {
            h->err = "non-finite coordinate (NaN / Inf) in the cloud";
            return ERASOR_E_INVALID;
        }
        // PCL returns the input unchanged (utils.cpp:88-91); the label search then answers with the point itself or its first duplicate
        rc = enqueue_passthrough(h, d_src, ns, nullptr, Q(h).query.p, nullptr);
        if (rc) return rc;
        HIPC(h, hipStreamSynchronize(h->stream));
        *nq_out = ns;
        return ERASOR_OK;
    }
    const uint32_t nq = nvox_host;
    *nq_out = nq;
    if (nq) {
        // identity lidar->body; the R-POD key output is ignored
        const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        // x*1 + y*0 + z*0 + 0 reproduces x exactly (0*finite = 0, x+0 = x; -0.0 would become +0.0, irrelevant for a centroid)
        LAUNCH(h, "q_centroids", k_centroids, cdiv((uint64_t)nq * 8, 256), 256, Q(h).scan_in, (const uint32_t *)Q(h).qk_b.p, (const uint32_t *)Q(h).qv_b.p,
               (const uint32_t *)Q(h).run_begin.p, (const uint32_t *)Q(h).d_nvox.p, Q(h).cent.p, Q(h).ukeys.p, Q(h).hkey.p, Q(h).hval.p, Q(h).hbits);
        LAUNCH(h, "q_nn", k_query_nn, cdiv((uint64_t)nq * NN_SUB, 256), 256, Q(h).scan_in, (const uint32_t *)Q(h).qv_b.p, (const uint32_t *)Q(h).run_begin.p,
               (const uint32_t *)Q(h).ukeys.p, (const float4 *)Q(h).cent.p, (const uint32_t *)Q(h).d_nvox.p, (const VoxGrid *)Q(h).qgrid.p, to_xf(I),
               h->dp, Q(h).d_qctr.p, Q(h).query.p, Q(h).qkey.p, (const uint32_t *)Q(h).hkey.p, (const uint32_t *)Q(h).hval.p, Q(h).hbits);
        HIPC(h, hipStreamSynchronize(h->stream));
    }
    Counters c;
    HIPC(h, hipMemcpy(&c, Q(h).d_qctr.p, sizeof(c), hipMemcpyDeviceToHost));
    if (c.sort_qoverflow) {
        h->err = "exact-sort segment queue overflow (site " + std::to_string(c.sort_qoverflow) + ")";
        return ERASOR_E_INTERNAL;
    }
    return ERASOR_OK;
}

int erasor_hip_voxelize_preserving_labels(erasor_hip_handle *h, const float *src, size_t n, double leaf_size, float *dst, size_t cap,
                                          size_t *n_out) {
    NOFLY(h);
    if (!h || (!src && n) || !(leaf_size > 0) || n > 0x3FFFFFFFull) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    prof_collect(h);
    const uint32_t ns = (uint32_t)n;
    q_drain(h);
    int rc = alloc_scan(h, std::max(ns, 1u));
    if (rc) return rc;
    if (ns) HIPC(h, hipMemcpyAsync(Q(h).scan.p, src, (size_t)ns * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    uint32_t nq = 0;
    rc = voxelize_device(h, (const float4 *)Q(h).scan.p, ns, leaf_size, &nq);
    if (rc) return rc;
    if (n_out) *n_out = nq;
    if (!dst) return ERASOR_OK;
    if (nq > cap) return ERASOR_E_CAPACITY;
    if (nq) HIPC(h, hipMemcpy(dst, Q(h).query.p, (size_t)nq * sizeof(float4), hipMemcpyDeviceToHost));
    return ERASOR_OK;
}

// ---- mapgen (src/mapgen/mapgen.hpp:187-305) ------------------------------------------------------
static int mg_append(erasor_hip_handle *h, DBuf<float4> &dst, uint64_t &ndst, const float4 *src, uint64_t n) {
    if (ndst + n > 0x3FFFFFF0ull) {
        h->err = "mapgen: map too large for 32-bit indexing";
        return ERASOR_E_CAPACITY;
    }
    if (ndst + n > dst.cap) {
        DBuf<float4> bigger;
        const size_t want = (size_t)std::max<uint64_t>((ndst + n) * 2, 1u << 20);
        if (ensure(h, bigger, want)) return ERASOR_E_NO_DEVICE;
        if (ndst) HIPC(h, hipMemcpyAsync(bigger.p, dst.p, (size_t)ndst * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        release(dst);
        dst = bigger;
        bigger.p = nullptr;
        bigger.cap = 0;
    }
    if (n) HIPC(h, hipMemcpyAsync(dst.p + ndst, src, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    ndst += n;
    return ERASOR_OK;
}

int erasor_hip_mapgen_begin(erasor_hip_handle *h, double leafsize, int is_large_scale) {
    NOFLY(h);
    if (!h || !(leafsize > 0)) return ERASOR_E_INVALID;
    h->mg_leaf = leafsize;
    h->mg_large = is_large_scale != 0;
    h->mg_initial = true;
    h->mg_active = true;
    h->mg_ncurr = h->mg_nmap = h->mg_ndone = 0;
    h->mg_cnt_voxel = h->mg_accum = 0;
    return ERASOR_OK;
}

int erasor_hip_mapgen_accum(erasor_hip_handle *h, const float *scan_xyzi, size_t n, const float T_pose[16], const float T_lidar2origin[16],
                            size_t *n_curr) {
    NOFLY(h);
    if (!h || (!scan_xyzi && n) || !T_pose || n > 0x3FFFFFFFull) return ERASOR_E_INVALID;
    if (!h->mg_active) {
        h->err = "erasor_hip_mapgen_accum before erasor_hip_mapgen_begin";
        return ERASOR_E_STATE;
    }
    HIPC(h, hipSetDevice(h->device));
    prof_collect(h);
    const uint32_t ns = (uint32_t)n;
    q_drain(h);
    int rc = alloc_scan(h, std::max(ns, 1u));
    if (rc) return rc;
    if (ensure(h, h->mg_tmp, (size_t)ns + 8)) return ERASOR_E_NO_DEVICE;
    static const float L2O[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1.73f, 0, 0, 0, 1};  // mapgen.hpp:212-215
    uint32_t n_kept = 0;
    if (ns) {
        HIPC(h, hipMemcpyAsync(Q(h).scan.p, scan_xyzi, (size_t)ns * sizeof(float4), hipMemcpyHostToDevice, h->stream));
        const float max_dist_square = (float)pow(2.7, 2);  // CAR_BODY_SIZE, mapgen.hpp:8,221
        LAUNCH(h, "mapgen", k_mapgen_flag, cdiv(ns, 256), 256, (const float4 *)Q(h).scan.p, ns, max_dist_square, Q(h).qflag.p);
        scan_u32(h, Q(h).qflag.p, Q(h).qpl.p, Q(h).qtops.p, ns, ns, nullptr, h->dn.p, "mapgen");
        LAUNCH(h, "mapgen", k_mapgen_scatter, cdiv(ns, 256), 256, (const float4 *)Q(h).scan.p, ns, (const uint32_t *)Q(h).qflag.p,
               (const uint32_t *)Q(h).qpl.p, (const uint32_t *)Q(h).qtops.p, to_xf(T_lidar2origin ? T_lidar2origin : L2O), to_xf(T_pose), h->mg_tmp.p);
        HIPC(h, hipMemcpyAsync(&n_kept, h->dn.p, 4, hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
    }
    uint32_t nq = 0;
    rc = voxelize_device(h, (const float4 *)h->mg_tmp.p, n_kept, 0.2, &nq);  // cloud_curr (mapgen.hpp:239: fixed 0.2 m leaf)
    if (rc) return rc;
    h->mg_ncurr = 0;
    rc = mg_append(h, h->mg_curr, h->mg_ncurr, (const float4 *)Q(h).query.p, nq);
    if (rc) return rc;
    if (h->mg_initial) {  // :241-243
        h->mg_nmap = 0;
        rc = mg_append(h, h->mg_map, h->mg_nmap, (const float4 *)Q(h).query.p, nq);
        if (rc) return rc;
        h->mg_initial = false;
    } else {  // :244-256
        rc = mg_append(h, h->mg_map, h->mg_nmap, (const float4 *)Q(h).query.p, nq);
        if (rc) return rc;
        if (h->mg_large && (h->mg_cnt_voxel++ % 500 == 0)) {
            HIPC(h, hipStreamSynchronize(h->stream));
            uint32_t nv = 0;
            rc = voxelize_device(h, (const float4 *)h->mg_map.p, (uint32_t)h->mg_nmap, h->mg_leaf, &nv);
            if (rc) return rc;
            rc = mg_append(h, h->mg_done, h->mg_ndone, (const float4 *)Q(h).query.p, nv);  // cloud_maps.push_back
            if (rc) return rc;
            h->mg_nmap = 0;  // cloud_map.clear()
        }
        ++h->mg_accum;
    }
    HIPC(h, hipStreamSynchronize(h->stream));
    if (n_curr) *n_curr = h->mg_ncurr;
    return ERASOR_OK;
}

int erasor_hip_mapgen_get(erasor_hip_handle *h, int which, float *dst, size_t cap, size_t *n_out) {
    NOFLY(h);
    if (!h || which < 0 || which > 2) return ERASOR_E_INVALID;
    if (!h->mg_active) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    const uint64_t n = which == 0 ? h->mg_ncurr : (which == 1 ? h->mg_nmap : (h->mg_large ? h->mg_ndone : 0) + h->mg_nmap);
    if (n_out) *n_out = (size_t)n;
    if (!dst) return ERASOR_OK;
    if (n > cap) return ERASOR_E_CAPACITY;
    HIPC(h, hipStreamSynchronize(h->stream));
    if (which == 0) {
        if (n) HIPC(h, hipMemcpy(dst, h->mg_curr.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    } else if (which == 1) {
        if (n) HIPC(h, hipMemcpy(dst, h->mg_map.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    } else {  // saveNaiveMap's cloud_src: previous submaps, then the remaining map (:267-279)
        const uint64_t nd = h->mg_large ? h->mg_ndone : 0;
        if (nd) HIPC(h, hipMemcpy(dst, h->mg_done.p, (size_t)nd * sizeof(float4), hipMemcpyDeviceToHost));
        if (h->mg_nmap) HIPC(h, hipMemcpy(dst + 4 * nd, h->mg_map.p, (size_t)h->mg_nmap * sizeof(float4), hipMemcpyDeviceToHost));
    }
    return ERASOR_OK;
}

int erasor_hip_mapgen_save(erasor_hip_handle *h, float *dst, size_t cap, size_t *n_out) {
    NOFLY(h);
    if (!h) return ERASOR_E_INVALID;
    if (!h->mg_active) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    prof_collect(h);
    const uint64_t nd = h->mg_large ? h->mg_ndone : 0, total = nd + h->mg_nmap;
    if (total > 0x3FFFFFFFull) return ERASOR_E_CAPACITY;
    DBuf<float4> src;
    if (ensure(h, src, (size_t)total + 8)) return ERASOR_E_NO_DEVICE;
    if (nd) HIPC(h, hipMemcpyAsync(src.p, h->mg_done.p, (size_t)nd * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    if (h->mg_nmap) HIPC(h, hipMemcpyAsync(src.p + nd, h->mg_map.p, (size_t)h->mg_nmap * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    uint32_t nq = 0;
    int rc = voxelize_device(h, (const float4 *)src.p, (uint32_t)total, h->mg_leaf, &nq);  // :291
    if (rc == ERASOR_OK) {
        if (n_out) *n_out = nq;
        if (dst) {
            if (nq > cap) rc = ERASOR_E_CAPACITY;
            else if (nq && hipMemcpy(dst, Q(h).query.p, (size_t)nq * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) rc = ERASOR_E_NO_DEVICE;
        }
    }
    (void)hipStreamSynchronize(h->stream);
    release(src);
    return rc;
}

int erasor_hip_count_static_dynamic(erasor_hip_handle *h, uint64_t *n_static, uint64_t *n_dynamic) {
    NOFLY(h);
    if (!h || !n_static || !n_dynamic) return ERASOR_E_INVALID;
    if (!h->have_map) return ERASOR_E_STATE;
    *n_static = h->st.F_static + h->st.O_static;
    *n_dynamic = h->st.F_dynamic + h->st.O_dynamic;
    return ERASOR_OK;
}

int erasor_hip_profiling(erasor_hip_handle *h, int enable) {
    if (!h) return ERASOR_E_INVALID;
    (void)chain_wait(h, -1);  // (a queued worker job reads h->prof and the event pool: ADVICE r05)
    h->prof = enable;
    return ERASOR_OK;
}
int erasor_hip_chain_timing(erasor_hip_handle *h, double *main_chain_us, double *between_steps_us, double *period_us, uint64_t *steps, int reset) {
    if (!h) return ERASOR_E_INVALID;
    if (main_chain_us) *main_chain_us = h->tm_n ? h->tm_span / (double)h->tm_n : 0.0;
    if (between_steps_us) *between_steps_us = h->tm_ngap ? h->tm_gap / (double)h->tm_ngap : 0.0;
    if (period_us) *period_us = h->tm_nper ? h->tm_period / (double)h->tm_nper : 0.0;
    if (steps) *steps = h->tm_n;
    if (reset) {
        h->tm_span = h->tm_gap = h->tm_period = 0;
        h->tm_n = h->tm_ngap = h->tm_nper = 0;
        h->tm_last_end = 0;
    }
    return ERASOR_OK;
}
int erasor_hip_profile_reset(erasor_hip_handle *h) {
    if (!h) return ERASOR_E_INVALID;
    (void)chain_wait(h, -1);
    (void)hipStreamSynchronize(h->stream);
    for (int k = 0; k < h->nqs; ++k) (void)hipStreamSynchronize(h->qstream[k]);
    prof_collect(h, true);
    for (auto &e : h->prof_tab) e = ProfEntry();
    return ERASOR_OK;
}
int erasor_hip_profile_get(erasor_hip_handle *h, const char **names, double *total_ms, uint64_t *launches, size_t cap, size_t *n) {
    if (!h) return ERASOR_E_INVALID;
    (void)hipStreamSynchronize(h->stream);
    for (int k = 0; k < h->nqs; ++k) (void)hipStreamSynchronize(h->qstream[k]);
    prof_collect(h, true);
    const size_t cnt = h->prof_names.size();
    if (n) *n = cnt;
    for (size_t k = 0; k < cnt && k < cap; ++k) {
        if (names) names[k] = h->prof_names[k].c_str();
        if (total_ms) total_ms[k] = h->prof_tab[k].ms;
        if (launches) launches[k] = h->prof_tab[k].launches;
    }
    return ERASOR_OK;
}
int erasor_hip_voi_split_bytes(erasor_hip_handle *h, uint64_t *algorithmic_bytes, uint64_t *physical_entries) {
    if (!h) return ERASOR_E_INVALID;
    if (!h->have_map) return ERASOR_E_STATE;
    const uint64_t oe = (uint64_t)h->capO - h->o_begin;
    if (physical_entries) *physical_entries = (uint64_t)h->nF + oe;
    // F region streams 16 B/entry (float4), the outskirts region 8 B/entry ({x,y} only); + 16 B of masks per 64 entries
    // (outskirts chunks whose bounding box lies outside the VoI circle are not read: what a launch must stream is the chunks the last
    // step's pass DID read -- counted on the device, DevState::n_o_read -- plus one 32-byte record per chunk)
    if (h->use_ometa && h->have_step) {
        const uint64_t rd = (uint64_t)h->st.n_o_read * CHUNK, nrec = ((uint64_t)h->capO - h->o_begin + CHUNK - 1) / CHUNK;
        if (physical_entries) *physical_entries = (uint64_t)h->nF + rd;
        if (algorithmic_bytes) *algorithmic_bytes = 16ull * h->nF + 8ull * rd + ((uint64_t)h->nF + rd) / 4 + 32ull * nrec;
        return ERASOR_OK;
    }
    if (algorithmic_bytes) *algorithmic_bytes = 16ull * h->nF + 8ull * oe + ((uint64_t)h->nF + oe) / 4;
    return ERASOR_OK;
}
int erasor_hip_overlap_counts(erasor_hip_handle *h, uint64_t *launched, uint64_t *used) {
    if (!h || !launched || !used) return ERASOR_E_INVALID;
    *launched = h->n_ov_launched;
    *used = h->n_ov_used;
    return ERASOR_OK;
}
int erasor_hip_drop_announced(erasor_hip_handle *h) {
    if (!h) return ERASOR_E_INVALID;
    NOFLY(h);
    HIPC(h, hipSetDevice(h->device));
    q_drain(h);
    return ERASOR_OK;
}
int erasor_hip_overlap_auto(erasor_hip_handle *h, int *mode, double *plain_period_us, double *overlapped_period_us) {
    if (!h) return ERASOR_E_INVALID;
    if (mode) *mode = h->ova.mode;
    if (plain_period_us) *plain_period_us = h->ova.est[0];
    if (overlapped_period_us) *overlapped_period_us = h->ova.est[1];
    return ERASOR_OK;
}
int erasor_hip_chain_batch(erasor_hip_handle *h, int n_scans, int lead) {
    if (!h || n_scans < 1 || n_scans > QBATCH_MAX || lead < 1 || lead > MAX_AHEAD - 1) return ERASOR_E_INVALID;
    NOFLY(h);
    const int rc = chain_wait(h, -1);  // (chains held back under the old setting go off first)
    h->batch_n = n_scans;
    h->batch_lead = lead;
    return rc;
}
int erasor_hip_chain_batch_counts(erasor_hip_handle *h, uint64_t *sets, uint64_t *chains) {
    if (!h) return ERASOR_E_INVALID;
    if (sets) *sets = h->n_batches;
    if (chains) *chains = h->n_batched_chains;
    return ERASOR_OK;
}
int erasor_hip_ahead_split_counts(erasor_hip_handle *h, uint64_t *launched, uint64_t *used) {
    if (!h) return ERASOR_E_INVALID;
    if (launched) *launched = h->n_spec_launched;
    if (used) *used = h->n_spec_used;
    return ERASOR_OK;
}
void *erasor_hip_stream(erasor_hip_handle *h) { return h ? (void *)h->stream : nullptr; }

// device buffers for callers that keep their scans (or the map) resident in HBM and have no HIP of their own (the C++ bench, the
// Python tests): plain hipMalloc / hipMemcpy / hipFree on the handle's device
int erasor_hip_device_alloc(erasor_hip_handle *h, size_t bytes, void **d_ptr) {
    if (!h || !d_ptr) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    *d_ptr = nullptr;
    HIPC(h, hipMalloc(d_ptr, bytes ? bytes : 1));
    return ERASOR_OK;
}
int erasor_hip_device_upload(erasor_hip_handle *h, void *d_dst, const void *src, size_t bytes) {
    if (!h || (!d_dst && bytes) || (!src && bytes)) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    if (bytes) HIPC(h, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return ERASOR_OK;
}
int erasor_hip_device_free(erasor_hip_handle *h, void *d_ptr) {
    if (!h) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    if (d_ptr) HIPC(h, hipFree(d_ptr));
    return ERASOR_OK;
}


int erasor_hip_erasor_run(erasor_hip_handle *h, const float *map_voi_xyzi, size_t n_map, const float *query_voi_xyzi, size_t n_query,
                          erasor_step_result *res) {
    NOFLY(h);
    int rc = set_map_common(h, map_voi_xyzi, n_map, false);
    if (rc) return rc;
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    return step_common(h, query_voi_xyzi, n_query, false, I, I, I, res, STEP_QUERY_PREVOXELIZED | STEP_VOI_EVERYTHING);
}
#ifdef ERASOR_HIP_TEST_HOOKS
// ---- test hooks: NOT part of the product library (round 4).  The test suite builds its own copy of this file with
// -DERASOR_HIP_TEST_HOOKS (erasor_amd/csrc/Makefile: tests/_build/liberasor_hip_hooks.so) and declares them itself. ----
// test hook: device libm probe (sqrt / div / atan2 in double)
int erasor_hip_probe_math(erasor_hip_handle *h, const double *x, const double *y, size_t n, double *o_sqrt, double *o_div, double *o_atan2) {
    NOFLY(h);
    if (!h || !x || !y || !o_sqrt || !o_div || !o_atan2) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    double *d = nullptr;
    HIPC(h, hipMalloc((void **)&d, 5 * n * sizeof(double)));
    HIPC(h, hipMemcpy(d, x, n * 8, hipMemcpyHostToDevice));
    HIPC(h, hipMemcpy(d + n, y, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_math, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, (const double *)d, (const double *)(d + n), (uint32_t)n, d + 2 * n,
                       d + 3 * n, d + 4 * n);
    HIPC(h, hipStreamSynchronize(h->stream));
    HIPC(h, hipMemcpy(o_sqrt, d + 2 * n, n * 8, hipMemcpyDeviceToHost));
    HIPC(h, hipMemcpy(o_div, d + 3 * n, n * 8, hipMemcpyDeviceToHost));
    HIPC(h, hipMemcpy(o_atan2, d + 4 * n, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return ERASOR_OK;
}

// test hook: bin keys of n points (x y z i rows) by the product's bin_key and by the float64-only restatement; counters[0..3] =
// n_ambiguous / n_neg_sector of the first, then of the second
int erasor_hip_probe_bin_keys(erasor_hip_handle *h, const float *xyzi, size_t n, uint32_t *k_fast, uint32_t *k_exact, uint32_t *counters) {
    NOFLY(h);
    if (!h || !xyzi || !k_fast || !k_exact || !counters || !n) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    float4 *d = nullptr;
    uint32_t *dk = nullptr;
    Counters *dc = nullptr;
    HIPC(h, hipMalloc((void **)&d, n * sizeof(float4)));
    HIPC(h, hipMalloc((void **)&dk, 2 * n * sizeof(uint32_t)));
    HIPC(h, hipMalloc((void **)&dc, 2 * sizeof(Counters)));
    HIPC(h, hipMemcpy(d, xyzi, n * sizeof(float4), hipMemcpyHostToDevice));
    HIPC(h, hipMemset(dc, 0, 2 * sizeof(Counters)));
    hipLaunchKernelGGL(k_probe_bin_keys, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, h->dp, (const float4 *)d, (uint32_t)n, dk, dk + n, dc, dc + 1);
    HIPC(h, hipStreamSynchronize(h->stream));
    HIPC(h, hipMemcpy(k_fast, dk, n * 4, hipMemcpyDeviceToHost));
    HIPC(h, hipMemcpy(k_exact, dk + n, n * 4, hipMemcpyDeviceToHost));
    Counters hc[2];
    HIPC(h, hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
    counters[0] = hc[0].n_ambiguous;
    counters[1] = hc[0].n_neg_sector;
    counters[2] = hc[1].n_ambiguous;
    counters[3] = hc[1].n_neg_sector;
    (void)hipFree(d);
    (void)hipFree(dk);
    (void)hipFree(dc);
    return ERASOR_OK;
}

// test hook: exact std::sort emulation of (key,payload) pairs on the device
int erasor_hip_exact_sort_u32(erasor_hip_handle *h, uint32_t *keys, uint32_t *vals, size_t n, uint32_t *n_fallback) {
    NOFLY(h);
    if (!h || (!keys && n) || (!vals && n) || n > 0x3FFFFFFFull) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    h->have_step = false;  // borrows the last step's query side
    const uint32_t ns = (uint32_t)n;
    q_drain(h);
    int rc = alloc_scan(h, std::max(ns, 1u));
    if (rc) return rc;
    LAUNCH(h, "q_begin", k_query_begin, 1, 256, Q(h).d_qctr.p, Q(h).bb.p, (uint32_t *)nullptr, 0u, Q(h).d_nvox.p, 0u);
    if (ns) {
        HIPC(h, hipMemcpyAsync(Q(h).qk_a.p, keys, (size_t)ns * 4, hipMemcpyHostToDevice, h->stream));
        HIPC(h, hipMemcpyAsync(Q(h).qv_a.p, vals, (size_t)ns * 4, hipMemcpyHostToDevice, h->stream));
    }
    Counters *dc = Q(h).d_qctr.p;
    run_exact_sort(h, ns);
    HIPC(h, hipStreamSynchronize(h->stream));
    Counters c;
    HIPC(h, hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost));
    if (c.sort_qoverflow) {
        h->err = "exact-sort segment queue overflow";
        return ERASOR_E_INTERNAL;
    }
    if (n_fallback) *n_fallback = c.n_sort_fallback;
    if (h->dbg_stamps.p) {
        unsigned long long t[32];
        (void)hipMemcpy(t, h->dbg_stamps.p, sizeof(t), hipMemcpyDeviceToHost);
        fprintf(stderr, "[esort stamps n=%u] phase1 %llu, queue %llu, finalize %llu cycles; levels:", ns, t[1] - t[0], t[2] - t[1], t[3] - t[2]);
        for (int i = 4; i < 15; ++i) fprintf(stderr, " %llu", t[i + 1] > t[i] ? t[i + 1] - t[i] : 0ull);
        fprintf(stderr, "; kernel span: first start -> last working end %.1f us, -> last end %.1f us\n", (double)(t[26] - t[28]) / 100.0,
                (double)(t[27] - t[28]) / 100.0);
        memset(t, 0, sizeof(t));
        t[28] = ~0ull;
        (void)hipMemcpy(h->dbg_stamps.p, t, sizeof(t), hipMemcpyHostToDevice);
    }
    if (ns) {
        HIPC(h, hipMemcpy(keys, Q(h).qk_b.p, (size_t)ns * 4, hipMemcpyDeviceToHost));
        HIPC(h, hipMemcpy(vals, Q(h).qv_b.p, (size_t)ns * 4, hipMemcpyDeviceToHost));
    }
    return ERASOR_OK;
}

// test hook: force the tombstone-free rebuild of the outskirts region (normally triggered by hole / room heuristics)
int erasor_hip_debug_rebuild_outskirts(erasor_hip_handle *h) {
    NOFLY(h);
    if (!h) return ERASOR_E_INVALID;
    if (!h->have_map) return ERASOR_E_STATE;
    HIPC(h, hipSetDevice(h->device));
    return rebuild_outskirts(h, h->nF + CHUNK);
}

// test hook: the stable LSD radix sort used for R-POD bucketing
int erasor_hip_radix_sort_u32(erasor_hip_handle *h, const uint32_t *keys, size_t n, int bits, uint32_t *keys_out, uint32_t *perm_out) {
    NOFLY(h);
    if (!h || (!keys && n) || n > 0x3FFFFFFFull) return ERASOR_E_INVALID;
    HIPC(h, hipSetDevice(h->device));
    h->have_step = false;  // borrows the last step's query side
    const uint32_t ns = (uint32_t)n;
    q_drain(h);
    int rc = alloc_scan(h, std::max(ns, 1u));
    if (rc) return rc;
    if (ns) HIPC(h, hipMemcpyAsync(Q(h).qkey.p, keys, (size_t)ns * 4, hipMemcpyHostToDevice, h->stream));
    const uint32_t *sk = nullptr, *sp = nullptr;
    rc = radix_sort(h, Q(h).qkey.p, ns, nullptr, bits, Q(h).qk_a.p, Q(h).qposL.p, Q(h).qv_a.p, Q(h).qposR.p, &sk, &sp, "radix_test");
    if (rc) return rc;
    HIPC(h, hipStreamSynchronize(h->stream));
    if (ns) {
        HIPC(h, hipMemcpy(keys_out, sk, (size_t)ns * 4, hipMemcpyDeviceToHost));
        HIPC(h, hipMemcpy(perm_out, sp, (size_t)ns * 4, hipMemcpyDeviceToHost));
    }
    return ERASOR_OK;
}
#endif  // ERASOR_HIP_TEST_HOOKS

}  // extern "C"
