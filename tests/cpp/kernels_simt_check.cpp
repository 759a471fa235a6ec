// Runs kernels of erasor_amd/csrc/kernels.hip.h UNMODIFIED on the CPU (tests/cpp/simt_emu: one thread per lane) and checks
// them against plain references: the counting sort of the map bucketing against std::stable_sort, the run detection and the
// chunk scan against loops, and the two per-bin kernels (R-GPF, per-bin voxelisation) against the CPU oracle's per-bin
// functions (oracle/erasor_oracle.cpp -- the checker, linked here as such).  Built and run by tests/test_kernels_on_cpu.py.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../erasor_amd/csrc/kernels.hip.h"
#include "../../include/erasor_hip.h"

extern "C" {
int orc_extract_ground(const erasor_params *p, const float *xyzi, size_t n, uint8_t *ground_mask, float *normals, double *ds, uint32_t *n_degenerate);
int orc_voxelize_preserving_labels(const float *xyzi, size_t n, double leaf, float *out, size_t cap, size_t *n_out);
}

using namespace ek;
static int g_fail = 0;
#define CHECK(cond, ...)                 \
    do {                                 \
        if (!(cond)) {                   \
            printf("  FAILED: " __VA_ARGS__); \
            printf("\n");               \
            ++g_fail;                    \
        }                                \
    } while (0)

static erasor_params seq05_params() {
    erasor_params p;
    memset(&p, 0, sizeof(p));
    p.max_range = 80.0;
    p.num_rings = 20;
    p.num_sectors = 108;
    p.max_h = 3.2;
    p.min_h = -1.3;
    p.th_bin_max_h = 0.05;
    p.scan_ratio_threshold = 0.3;
    p.num_lowest_pts = 5;
    p.minimum_num_pts = 10;
    p.rejection_ratio = 0.33;
    p.gf_dist_thr = 0.15;
    p.gf_iter = 3;
    p.gf_num_lpr = 10;
    p.gf_th_seeds_height = 0.5;
    p.map_voxel_size = 0.2;
    p.version = 3;
    p.query_voxel_size = 0.2;
    p.removal_interval = 1;
    p.voi_max_range = 80.0;
    p.submap_size = 200.0;
    return p;
}
static DP make_dp(const erasor_params &p) {  // erasor_hip.hip: fill_dp
    DP d;
    memset(&d, 0, sizeof(d));
    d.max_r = p.max_range;
    d.R = p.num_rings;
    d.S = p.num_sectors;
    d.B = p.num_rings * p.num_sectors;
    d.ring_size = p.max_range / p.num_rings;
    d.sector_size = 2 * PI_REF / p.num_sectors;
    d.max_h = p.max_h;
    d.min_h = p.min_h;
    d.th_bin_max_h = p.th_bin_max_h;
    d.srt_thr = p.scan_ratio_threshold;
    d.gf_dist = p.gf_dist_thr;
    d.gf_seeds_h = p.gf_th_seeds_height;
    d.voi_r2 = p.voi_max_range * p.voi_max_range;
    d.num_lowest = p.num_lowest_pts;
    d.min_pts = p.minimum_num_pts;
    d.gf_iter = p.gf_iter;
    d.gf_lpr = p.gf_num_lpr;
    d.version = p.version;
    d.leaf_map = (float)p.map_voxel_size;
    d.leaf_query = (float)p.query_voxel_size;
    return d;
}

// ---- map bucketing: k_mb_hist / k_mb_colscan / k_mb_scatter_w == a stable sort by key ------------------------------------
static void test_map_bucketing(std::mt19937 &rng) {
    const uint32_t cases[4][2] = {{1u, 40u}, {4095u, 300u}, {4097u, 300u}, {30000u, 2161u}};
    for (const auto &cs : cases) {
        const uint32_t n = cs[0], nb = cs[1];
        std::vector<uint32_t> keys(n), src(n);
        std::vector<float4> pts(n);
        for (uint32_t i = 0; i < n; ++i) {
            keys[i] = (rng() % 7 == 0) ? nb - 1 : rng() % (nb / 3);  // a skewed distribution with long runs of one key
            src[i] = 1000000u + i;
            pts[i] = make_float4((float)i, 0.5f, -1.f, 40.f);
        }
        const uint32_t ntile = (n + MB_TILE - 1) / MB_TILE;
        std::vector<uint32_t> hist((size_t)ntile * mb_row_stride(nb) + 8, 0xDEADu), tot(nb + 2, 0), off(nb + 2, 0), dsrc(n), dkeys(n), nd(1, n);
        std::vector<float4> dpts(n);
        int bits = 1;
        while ((1u << bits) < nb) ++bits;
        simt::run_grid(std::max(1u, ntile / 2), 1024, [&] { k_mb_hist(keys.data(), n + 100, nd.data(), nb, hist.data(), tot.data()); });
        simt::run_grid((nb + MB_PAD - 1) / MB_PAD, 256, [&] { k_mb_colscan(hist.data(), n + 100, nd.data(), nb, tot.data(), off.data()); });
        simt::run_grid(std::max(1u, ntile / 2), 1024, [&] {
            k_mb_scatter_w<MBW_NB_SMALL>(keys.data(), pts.data(), src.data(), n + 100, nd.data(), nb, bits, hist.data(), dpts.data(), dsrc.data(), dkeys.data(), nullptr);
        });
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
        bool ok = true;
        for (uint32_t i = 0; ok && i < n; ++i) ok = dkeys[i] == keys[order[i]] && dsrc[i] == src[order[i]] && dpts[i].x == pts[order[i]].x;
        for (uint32_t b = 0; ok && b <= nb; ++b) {
            const uint32_t want = (uint32_t)(std::lower_bound(order.begin(), order.end(), b, [&](uint32_t a, uint32_t k) { return keys[a] < k; }) - order.begin());
            ok = off[b] == want;
        }
        printf("map bucketing (hist / column scan / scatter)      n=%6u  %s\n", n, ok ? "ok" : "MISMATCH");
        CHECK(ok, "map bucketing n=%u", n);
    }
}

// ---- run detection: k_run_count / k_run_emit -----------------------------------------------------------------------
static void test_runs(std::mt19937 &rng) {
    for (uint32_t n : {0u, 1u, 1023u, 1024u, 1025u, 5000u}) {
        std::vector<uint32_t> keys(n);
        uint32_t k = 5;
        for (uint32_t i = 0; i < n; ++i) {
            if (rng() % 3 == 0) k += 1 + rng() % 4;
            keys[i] = k;
        }
        if (n > 3000) std::fill(keys.begin() + 900, keys.begin() + 2300, keys[900]);  // one run across two tile borders
        const uint32_t ntile = std::max(1u, (n + 1023) / 1024);
        std::vector<uint32_t> tops(ntile + 4, 0), run_begin(n + 2, 0xFFFFFFFFu), nv(1, 0xFFFFFFFFu);
        simt::run_grid(ntile, 256, [&] { k_run_count(keys.data(), n, tops.data()); });
        simt::run_grid(ntile, 256, [&] { k_run_emit(keys.data(), n, tops.data(), ntile, run_begin.data(), nv.data()); });
        std::vector<uint32_t> want;
        for (uint32_t i = 0; i < n; ++i)
            if (i == 0 || keys[i] != keys[i - 1]) want.push_back(i);
        bool ok = nv[0] == want.size() && run_begin[want.size()] == n;
        for (size_t v = 0; ok && v < want.size(); ++v) ok = run_begin[v] == want[v];
        printf("run detection (tile totals -> run_begin)           n=%6u  %s\n", n, ok ? "ok" : "MISMATCH");
        CHECK(ok, "runs n=%u", n);
    }
}

// ---- R-GPF and the per-bin voxelisation against the oracle's per-bin functions -----------------------------------------------
static void test_per_bin(std::mt19937 &rng) {
    const erasor_params p = seq05_params();
    const DP P = make_dp(p);
    const uint32_t B = (uint32_t)P.B;
    std::uniform_real_distribution<float> u01(0.f, 1.f);
    // a few reverted bins of different sizes: sloped ground quantised to 1 cm (ties in z), clutter above, something below min_h
    // <= 1024: one key per thread of the level-synchronous sort; <= 2048: two; <= 4096: four (3000: its cloud of curr + ground points is
    // beyond 2048 as well -- staged twice); beyond 4096: the global-memory paths of both stages (block_esort, binvox_core)
    const uint32_t sizes[] = {12, 40, 300, 900, 1700, 2300, 3000, 4500};
    const uint32_t nbin = sizeof(sizes) / sizeof(sizes[0]);
    std::vector<uint32_t> moff(B + 3, 0), qoff(B + 3, 0), rev_list, rev_key;
    std::vector<float4> spts, sq;
    std::vector<std::vector<float4>> map_bin(nbin), cur_bin(nbin);
    for (uint32_t b = 0; b < nbin; ++b) {
        const float x0 = 10.f + 4.f * b, y0 = 2.f;
        for (uint32_t i = 0; i < sizes[b]; ++i) {
            const float x = x0 + 3.9f * u01(rng), y = y0 + 1.5f * u01(rng);
            float z;
            const float r = u01(rng);
            if (r < 0.7f) z = -1.70f + 0.02f * (x - x0) + 0.01f * (float)(int)(6.f * u01(rng));  // ground, 1 cm steps
            else if (r < 0.97f) z = -1.5f + 3.5f * u01(rng);                                      // a wall / a car
            else z = -1.9f - u01(rng);                                                            // below min_h
            if (b % 2 == 1) z = roundf(z * 64.f) / 64.f;  // every other bin: equal z everywhere (the order of equal keys is the introsort's)
            map_bin[b].push_back(make_float4(x, y, z, (rng() % 5 == 0) ? 252.f : 40.f));
        }
        for (uint32_t i = 0; i < sizes[b] / 3 + 2; ++i)
            cur_bin[b].push_back(make_float4(x0 + 3.9f * u01(rng), y0 + 1.5f * u01(rng), -1.7f + 0.05f * u01(rng), 40.f));
    }
    // the bins sit at keys 7, 107, 207, ... of the sorted VoI / the sorted query
    for (uint32_t key = 0, b = 0; key <= B; ++key) {
        moff[key] = (uint32_t)spts.size();
        qoff[key] = (uint32_t)sq.size();
        if (b < nbin && key == 7 + 100 * b) {
            spts.insert(spts.end(), map_bin[b].begin(), map_bin[b].end());
            sq.insert(sq.end(), cur_bin[b].begin(), cur_bin[b].end());
            rev_list.push_back(key);
            ++b;
        } else if (key % 50 == 3) {  // unrelated bins in between
            spts.push_back(make_float4(1.f, 1.f, -1.7f, 40.f));
        }
    }
    moff[B + 1] = moff[B + 2] = (uint32_t)spts.size();
    qoff[B + 1] = qoff[B + 2] = (uint32_t)sq.size();
    const size_t G = spts.size() + sq.size() + 64;
    DevState st;
    memset(&st, 0, sizeof(st));
    st.n_rev = nbin;
    Counters ctr;
    memset(&ctr, 0, sizeof(ctr));
    std::vector<uint32_t> gsK(G), gsV(G), gsL(G), gsR(G), gsH(G / 32 + 2 * B + 16), gsK2(G), gsV2(G), grank(G), glist(G), ng(nbin + 1, 0);
    std::vector<uint8_t> gflag(G, 9);
    std::vector<float> plane_n((size_t)nbin * P.gf_iter * 3 + 8, 0.f);
    std::vector<double> plane_d((size_t)nbin * P.gf_iter + 8, 0.0);
    std::vector<uint32_t> vox_off(nbin + 1, 0), nvox(nbin + 1, 0);
    for (uint32_t b = 0; b < nbin; ++b) vox_off[b + 1] = vox_off[b] + (uint32_t)(cur_bin[b].size() + sizes[b]);
    std::vector<float4> gsC(G), vox_out(vox_off[nbin] + 8);
    RevArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.moff = moff.data();
    ra.spts = spts.data();
    ra.qoff = qoff.data();
    ra.sq = sq.data();
    ra.gsK = gsK.data();
    ra.gsV = gsV.data();
    ra.gsL = gsL.data();
    ra.gsR = gsR.data();
    ra.gsH = gsH.data();
    ra.gsK2 = gsK2.data();
    ra.gsV2 = gsV2.data();
    ra.gsC = gsC.data();
    ra.gflag = gflag.data();
    ra.grank = grank.data();
    ra.glist = glist.data();
    ra.ng_arr = ng.data();
    ra.plane_n = plane_n.data();
    ra.plane_d = plane_d.data();
    ra.vox_out = vox_out.data();
    ra.nvox_out = nvox.data();
    ra.ctr = &ctr;
    ra.dbg = nullptr;
    simt::run_grid(2, 1024, [&] { k_rgpf2(P, rev_list.data(), &st, ra); });
    for (uint32_t b = 0; b < nbin; ++b) {
        const uint32_t M = sizes[b], o0 = moff[rev_list[b]];
        std::vector<uint8_t> mask(M);
        std::vector<float> normals(P.gf_iter * 3, 0.f);
        std::vector<double> ds(P.gf_iter, 0.0);
        uint32_t ndeg = 0;
        orc_extract_ground(&p, &map_bin[b][0].x, M, mask.data(), normals.data(), ds.data(), &ndeg);
        bool ok = true;
        uint32_t ngr = 0;
        for (uint32_t i = 0; i < M; ++i) {
            ok = ok && gflag[o0 + i] == mask[i];
            ngr += mask[i];
        }
        ok = ok && ng[b] == ngr;
        for (int t = 0; ok && t < P.gf_iter * 3; ++t) ok = memcmp(&plane_n[(size_t)b * P.gf_iter * 3 + t], &normals[t], 4) == 0;
        for (int t = 0; ok && t < P.gf_iter; ++t) ok = memcmp(&plane_d[(size_t)b * P.gf_iter + t], &ds[t], 8) == 0;
        printf("R-GPF (k_rgpf2) vs the oracle's extract_ground    M=%6u  ground=%u  %s\n", M, ngr, ok ? "ok (mask, every iteration's plane bit-exact)" : "MISMATCH");
        CHECK(ok, "rgpf bin %u", b);
    }
    // per-bin voxelisation of [curr bin | ground of the map bin] (erasor.cpp:512-528)
    simt::run_grid(2, 1024, [&] { k_binvox2(P, rev_list.data(), &st, vox_off.data(), ra); });
    for (uint32_t b = 0; b < nbin; ++b) {
        const uint32_t o0 = moff[rev_list[b]];
        std::vector<float4> in(cur_bin[b]);
        for (uint32_t k = 0; k < ng[b]; ++k) in.push_back(spts[o0 + glist[o0 + k]]);
        std::vector<float4> want(in.size() + 1);
        size_t nw = 0;
        orc_voxelize_preserving_labels(&in[0].x, in.size(), p.map_voxel_size, &want[0].x, in.size(), &nw);
        bool ok = nvox[b] == nw;
        for (size_t v = 0; ok && v < nw; ++v) ok = memcmp(&vox_out[vox_off[b] + v], &want[v], 16) == 0;
        printf("per-bin voxelisation (k_binvox2) vs the oracle    in=%6zu -> %zu voxels  %s\n", in.size(), nw, ok ? "ok (bit-exact, in order)" : "MISMATCH");
        CHECK(ok, "binvox bin %u (got %u voxels, want %zu)", b, nvox[b], nw);
    }
    CHECK(ctr.err == 0 && ctr.sort_qoverflow == 0, "error flags %u %u", ctr.err, ctr.sort_qoverflow);
    // the fused kernel of v3 (R-GPF and the per-bin voxelisation of a bin back to back in one workgroup) must leave exactly what the two
    // separate kernels left -- bins of every size class: level-synchronous sort, one wavefront per segment
    {
        std::vector<uint32_t> gsK_(2 * G), gsV_(2 * G), gsL_(2 * G), gsR_(2 * G), gsH_(2 * (G / 32 + 2 * B + 16)), gsK2_(2 * G), gsV2_(2 * G), grank_(G), glist_(G),
            ng_(nbin + 1, 0), nvox_(nbin + 1, 0);
        std::vector<uint8_t> gflag_(G, 9);
        std::vector<float> plane_n_(plane_n.size(), 0.f);
        std::vector<double> plane_d_(plane_d.size(), 0.0);
        std::vector<float4> gsC_(G), vox_out_(vox_out.size());
        Counters ctr_;
        memset(&ctr_, 0, sizeof(ctr_));
        uint32_t n_rare = 0;
        for (uint32_t b = 0; b < nbin; ++b) n_rare += (sizes[b] > PB_CAP || sizes[b] + cur_bin[b].size() > PB_CAP) ? 1u : 0u;
        RevArgs rb = ra;
        rb.gsK = gsK_.data();
        rb.gsV = gsV_.data();
        rb.gsL = gsL_.data();
        rb.gsR = gsR_.data();
        rb.gsH = gsH_.data();
        rb.gsK2 = gsK2_.data();
        rb.gsV2 = gsV2_.data();
        rb.gsC = gsC_.data();
        rb.gflag = gflag_.data();
        rb.grank = grank_.data();
        rb.glist = glist_.data();
        rb.ng_arr = ng_.data();
        rb.plane_n = plane_n_.data();
        rb.plane_d = plane_d_.data();
        rb.vox_out = vox_out_.data();
        rb.nvox_out = nvox_.data();
        rb.ctr = &ctr_;
        rb.vox_base = (uint32_t)G;
        rb.h_base = (uint32_t)(G / 32 + 2 * B + 16);
        simt::run_grid(3, 1024, [&] { k_revert_bins(P, rev_list.data(), &st, vox_off.data(), rb); });
        bool same = ng_ == ng && nvox_ == nvox && memcmp(plane_n_.data(), plane_n.data(), plane_n.size() * 4) == 0 &&
                    memcmp(plane_d_.data(), plane_d.data(), plane_d.size() * 8) == 0;
        for (uint32_t b = 0; same && b < nbin; ++b) {
            const uint32_t o0 = moff[rev_list[b]];
            same = memcmp(&gflag_[o0], &gflag[o0], sizes[b]) == 0 && memcmp(&grank_[o0], &grank[o0], sizes[b] * 4) == 0 &&
                   memcmp(&glist_[o0], &glist[o0], ng[b] * 4) == 0 && memcmp(&vox_out_[vox_off[b]], &vox_out[vox_off[b]], (size_t)nvox[b] * 16) == 0;
        }
        printf("k_revert_bins (%u of %u bins beyond the LDS-resident size) == k_rgpf2 + k_binvox2  %s\n", n_rare, nbin, same ? "ok" : "MISMATCH");
        CHECK(same && n_rare > 0 && n_rare < nbin && ctr_.err == 0 && ctr_.sort_qoverflow == 0, "fused per-bin launch");
    }
}


// ---- the map store: VoI split (a step's own and the one launched ahead), chunk scan, gather ---------------------------------
static void test_map_store(std::mt19937 &rng) {
    const erasor_params p = seq05_params();
    DP P = make_dp(p);
    P.voi_r2 = 30.0 * 30.0;
    std::uniform_real_distribution<float> u(-60.f, 60.f), uz(-1.6f, 2.5f);
    const uint32_t nF = 5 * CHUNK + 137, capO = 24 * CHUNK, o_begin = 7 * CHUNK + 311;
    std::vector<float4> F(nF + CHUNK);
    std::vector<float2> Oxy(capO), Ozi(capO);
    for (uint32_t i = 0; i < nF; ++i) F[i] = make_float4(u(rng) * 0.6f, u(rng) * 0.6f, uz(rng), (rng() % 6 == 0) ? 252.f : 40.f);
    uint32_t o_valid = 0;
    for (uint32_t i = 0; i < capO; ++i) {
        // (the outskirts are written in runs that are close in space: chunk j sits around x = (j % 8 - 4) * 20 m -- some chunks are far from
        // the VoI circle and can be skipped by their bounding box, some straddle it)
        Oxy[i] = make_float2(u(rng) * 0.25f + ((float)((i / CHUNK) % 8) - 4.f) * 20.f, u(rng));
        Ozi[i] = make_float2(uz(rng), (rng() % 6 == 0) ? 253.f : 44.f);
        if (i >= o_begin) {
            if (rng() % 9 == 0) reinterpret_cast<uint32_t *>(&Oxy[i])[0] = HOLE_BITS;  // tombstones of earlier steps
            else ++o_valid;
        }
    }
    const double xc = 3.25, yc = -2.5;
    const float T[12] = {0.96f, 0.28f, 0.f, -2.42f, -0.28f, 0.96f, 0.f, 3.31f, 0.f, 0.f, 1.f, 0.1f};
    Xf To2b;
    memcpy(To2b.m, T, sizeof(T));
    // scalar expectation, in the store's logical order [F | outskirts]
    struct Want { float4 ego; uint32_t key, src; };
    std::vector<Want> want;
    std::vector<float4> leaving;
    Counters rc;
    memset(&rc, 0, sizeof(rc));
    uint32_t voiF = 0, validO = 0;
    for (uint32_t i = 0; i < nF; ++i) {
        const double dx = (double)F[i].x - xc, dy = (double)F[i].y - yc;
        if (dx * dx + dy * dy < P.voi_r2) {
            const float4 e = xform(To2b, F[i]);
            want.push_back({e, bin_key(P, e.x, e.y, e.z, &rc), i});
            ++voiF;
        } else
            leaving.push_back(F[i]);
    }
    std::vector<uint32_t> tomb;
    for (uint32_t i = o_begin; i < capO; ++i) {
        if (__float_as_uint(Oxy[i].x) == HOLE_BITS) continue;
        const double dx = (double)Oxy[i].x - xc, dy = (double)Oxy[i].y - yc;
        if (dx * dx + dy * dy < P.voi_r2) {
            const float4 e = xform(To2b, make_float4(Oxy[i].x, Oxy[i].y, Ozi[i].x, Ozi[i].y));
            want.push_back({e, bin_key(P, e.x, e.y, e.z, &rc), nF + validO});
            tomb.push_back(i);
        }
        ++validO;
    }
    for (int variant = 0; variant < 4; ++variant) {
        const int ahead = variant & 1, warm = variant >> 1;  // warm: the chunk records exist already (built by an earlier pass): chunks get skipped
        std::vector<float2> oxy(Oxy), ozi(Ozi);
        std::vector<OMeta> ometa(capO / CHUNK + 8);
        memset(ometa.data(), 0, ometa.size() * sizeof(OMeta));
        const uint32_t nFchunks = (nF + CHUNK - 1) / CHUNK, o_chunk0 = o_begin / CHUNK, nOchunks = capO / CHUNK - o_chunk0, nchunks = nFchunks + nOchunks;
        std::vector<unsigned long long> vmask((size_t)(nchunks + 4) * CHUNK_TILES, 0), hmask((size_t)(nchunks + 4) * CHUNK_TILES, 0), lab(128, 1);
        std::vector<uint32_t> cinfo(nchunks + 8, 0), pvl(nchunks + 8, 0), phl(nchunks + 8, 0), topv(8, 9), toph(8, 9), mb_tot(64, 5);
        DevState st, init;
        memset(&init, 0, sizeof(init));
        init.nF = nF;
        init.o_begin = o_begin;
        init.O_static = 1000000;
        init.O_dynamic = 1000000;
        st = init;
        Counters ctr, qctr;
        memset(&ctr, 0, sizeof(ctr));
        memset(&qctr, 0, sizeof(qctr));
        for (int pass = warm ? 0 : 1; pass < 2; ++pass) {  // (warm: a first pass around a far-away centre builds the records)
            const double px = pass ? xc : xc + 500.0;
            if (pass) {
                std::fill(cinfo.begin(), cinfo.end(), 0xDEADBEEFu);
                std::fill(vmask.begin(), vmask.end(), ~0ull);
                std::fill(hmask.begin(), hmask.end(), ~0ull);
            }
            if (ahead)  // extents from the committed device state, upper-bound grid
                simt::run_grid(3, 256, [&] {
                    k_voi_split(F.data(), 0u, 0u, oxy.data(), 0u, 0u, 0u, px, yc, P.voi_r2, vmask.data(), hmask.data(), cinfo.data(), &st, capO / CHUNK, nchunks + 2,
                                ometa.data(), StepEnd{});
                });
            else
                simt::run_grid((nchunks + 3) / 4, 256, [&] {
                    k_voi_split(F.data(), nF, nFchunks, oxy.data(), o_begin, o_chunk0, nOchunks, px, yc, P.voi_r2, vmask.data(), hmask.data(), cinfo.data(), nullptr, 0u,
                                0u, ometa.data(), StepEnd{});
                });
        }
        uint32_t n_read = 0;
        for (uint32_t c = nFchunks; c < nchunks; ++c) n_read += cinfo[c] >> 31;
        if (getenv("DBG_CINFO")) { printf("   [dbg] variant %d cinfo:", variant); for (uint32_t c = 0; c < nchunks; ++c) printf(" %08x", cinfo[c]); printf("\n"); }
        CHECK(warm ? (n_read > 0 && n_read + 3 <= nOchunks) : n_read == nOchunks, "outskirts chunks read: %u of %u (warm=%d)", n_read, nOchunks, warm);
        simt::run_grid(1, 1024, [&] {
            // (ahead: the scan is launched ahead like the split -- extents and starting state from the committed device state)
            k_chunk_scan_one(cinfo.data(), ahead ? 0u : nchunks, pvl.data(), phl.data(), topv.data(), toph.data(), 1u, ahead ? 0u : nFchunks, &st, &ctr, init,
                             lab.data(), mb_tot.data(), 64u, ometa.data(), ahead ? capO / CHUNK : 0u, ahead ? nchunks + 8u : 0u);
        });
        CHECK(st.n_o_read == n_read, "n_o_read %u vs %u", st.n_o_read, n_read);
        {   // the two-level scan (maps beyond k_chunk_scan_one's 131072 chunks) opens the step with the same state and the same prefixes
            std::vector<uint32_t> pvl2(nchunks + 8, 0), phl2(nchunks + 8, 0), topv2(8, 0), toph2(8, 0), mb2(64, 5);
            std::vector<unsigned long long> lab2(128, 1);
            DevState st2 = init;
            Counters ctr2;
            memset(&ctr2, 0, sizeof(ctr2));
            const uint32_t ntop = (nchunks + 1023) / 1024;
            std::vector<uint32_t> topr2(8, 77);
            std::vector<OMeta> ometa2(ometa);
            simt::run_grid(ntop, 256, [&] { k_chunk_scan_local(cinfo.data(), nchunks, pvl2.data(), phl2.data(), topv2.data(), toph2.data(), topr2.data(), nullptr, 0u, 0u); });
            simt::run_grid(1, 1024, [&] {
                k_chunk_scan_top(topv2.data(), toph2.data(), ntop, pvl2.data(), phl2.data(), nchunks, nFchunks, &st2, &ctr2, init, lab2.data(), mb2.data(), 64u,
                                 topr2.data(), ometa2.data(), 0u, 0u);
            });
            CHECK(memcmp(ometa2.data(), ometa.data(), ometa.size() * sizeof(OMeta)) == 0, "both scans void the same chunk records");
            st2.t_open = st.t_open = 0;  // (a time stamp)
            bool same = memcmp(&st2, &st, sizeof(st)) == 0;
            for (uint32_t c = 0; same && c < nchunks; ++c)
                same = pvl2[c] + topv2[c >> 10] == pvl[c] + topv[c >> 10] && phl2[c] + toph2[c >> 10] == phl[c] + toph[c >> 10];
            CHECK(same, "two-level chunk scan == one-launch chunk scan (ahead=%d)", ahead);
        }
        bool ok = st.voi_total == want.size() && st.voiF == voiF && st.validF == nF && st.valid_total == nF + validO &&
                  st.n_leaving == leaving.size() && st.o_new_begin == o_begin - (uint32_t)leaving.size() && lab[5] == 0 && mb_tot[7] == 0;
        std::vector<float4> ego(want.size() + 8);
        std::vector<uint32_t> key(want.size() + 8), src(want.size() + 8);
        simt::run_grid(5, 256, [&] {
            k_voi_gather(F.data(), nF, nFchunks, oxy.data(), ozi.data(), o_chunk0, nOchunks, vmask.data(), hmask.data(), cinfo.data(), pvl.data(),
                         phl.data(), topv.data(), toph.data(), To2b, P, &st, &ctr, &qctr, ego.data(), key.data(), src.data(), ometa.data());
        });
        {   // the chunk records afterwards: void where leaving points arrived, counts corrected where entries were tombstoned, and every
            // known record describes its chunk (box covers the valid entries, count exact)
            bool rec = true;
            for (uint32_t a = o_chunk0; rec && a < capO / CHUNK; ++a) {
                const bool recv = st.o_new_begin < o_begin && a >= st.o_new_begin / CHUNK && a <= (o_begin - 1) / CHUNK;
                if (recv) { rec = ometa[a].known == 0; continue; }
                if (!ometa[a].known) continue;
                uint32_t cnt = 0;
                for (uint32_t i = a * CHUNK; rec && i < (a + 1) * CHUNK; ++i) {
                    if (i < o_begin || __float_as_uint(oxy[i].x) == HOLE_BITS) continue;
                    ++cnt;
                    rec = oxy[i].x >= ometa[a].xmin && oxy[i].x <= ometa[a].xmax && oxy[i].y >= ometa[a].ymin && oxy[i].y <= ometa[a].ymax;
                }
                rec = rec && cnt == ometa[a].valid;
            }
            for (uint32_t a = st.o_new_begin / CHUNK; rec && a < o_chunk0; ++a) rec = ometa[a].known == 0;
            CHECK(rec, "chunk records after the step (variant %d)", variant);
        }
        if (!ok) printf("   [dbg] state mismatch: voi_total %u (want %zu) voiF %u (%u) validF %u (%u) valid_total %u (%u) n_leaving %u (%zu)\n", st.voi_total, want.size(), st.voiF, voiF, st.validF, nF, st.valid_total, nF + validO, st.n_leaving, leaving.size());
        for (size_t i = 0; ok && i < want.size(); ++i) ok = memcmp(&ego[i], &want[i].ego, 16) == 0 && key[i] == want[i].key && src[i] == want[i].src;
        for (size_t i = 0; ok && i < leaving.size(); ++i) {  // [leaving (F order) | old outskirts]
            const uint32_t d = st.o_new_begin + (uint32_t)i;
            ok = oxy[d].x == leaving[i].x && oxy[d].y == leaving[i].y && ozi[d].x == leaving[i].z && ozi[d].y == leaving[i].w;
        }
        for (size_t i = 0; ok && i < tomb.size(); ++i) ok = __float_as_uint(oxy[tomb[i]].x) == HOLE_BITS;
        long long dyn = 0, stat = 0;  // parse_dynamic_obj counters of the outskirts: + leaving, - entering
        for (auto &q : leaving) (is_dynamic_label(q.w) ? dyn : stat) += 1;
        for (uint32_t i : tomb) (is_dynamic_label(Ozi[i].y) ? dyn : stat) -= 1;
        ok = ok && (long long)st.O_dynamic == 1000000 + dyn && (long long)st.O_static == 1000000 + stat;
        printf("map store: split%s%s / chunk scan / gather     VoI %zu (%u resident), %zu leaving, %zu entering, %u of %u outskirts chunks read  %s\n",
               ahead ? " (launched ahead)" : "                 ", warm ? ", records known" : "               ", want.size(), voiF, leaving.size(), tomb.size(),
               n_read, nOchunks, ok ? "ok" : "MISMATCH");
        CHECK(ok, "map store variant=%d", variant);
    }
}

// Round 4: the two-level chunk scan launched AHEAD (extents and starting state from the committed device state, grid an upper bound)
// against the same scan with the host's arguments -- config 4's 38 k chunks -- and against k_chunk_scan_one where that applies
static void test_chunk_scan_ahead(std::mt19937 &rng) {
    for (uint32_t nchunks : {1u, 1000u, 16384u, 16385u, 40000u, 49152u, 70001u}) {
        for (uint32_t nFchunks : {0u, 1u, 16383u, 16384u, 20000u, 32768u, 39999u, 70001u}) {
            if (nFchunks > nchunks) continue;
            std::vector<uint32_t> cinfo(nchunks + 2048, 0);
            uint32_t n_read = 0;
            for (uint32_t c = 0; c < nchunks; ++c) {
                const uint32_t h = rng() % 1025u, v = h ? rng() % (h + 1u) : 0u, r = (c >= nFchunks && rng() % 3u == 0) ? 1u : 0u;
                cinfo[c] = v | (h << 16) | (r << 31);
                n_read += r;
            }
            const uint32_t ntop = std::max(1u, (nchunks + 1023) / 1024), grid = (nchunks + 64 + 1023) / 1024, ob_chunks = 100000u;
            DevState init;
            memset(&init, 0, sizeof(init));
            init.o_begin = ob_chunks * CHUNK + 17u;
            init.nF = nFchunks ? (nFchunks - 1) * CHUNK + 5u : 0u;
            const uint32_t capO_chunks = ob_chunks + (nchunks - nFchunks);
            struct Run {
                std::vector<uint32_t> pvl, phl, topv, toph, topr, mb;
                std::vector<unsigned long long> lab;
                DevState st;
                Counters ctr;
            } a, b, c1;
            for (Run *r : {&a, &b, &c1}) {
                r->pvl.assign(nchunks + 2048, 0);
                r->phl.assign(nchunks + 2048, 0);
                r->topv.assign(grid + 8, 9);
                r->toph.assign(grid + 8, 9);
                r->topr.assign(grid + 8, 7);
                r->mb.assign(64, 5);
                r->lab.assign(128, 1);
                r->st = init;
                memset(&r->ctr, 0, sizeof(r->ctr));
            }
            simt::run_grid(ntop, 256, [&] {
                k_chunk_scan_local(cinfo.data(), nchunks, a.pvl.data(), a.phl.data(), a.topv.data(), a.toph.data(), a.topr.data(), nullptr, 0u, 0u);
            });
            simt::run_grid(1, 1024, [&] {
                k_chunk_scan_top(a.topv.data(), a.toph.data(), ntop, a.pvl.data(), a.phl.data(), nchunks, nFchunks, &a.st, &a.ctr, init, a.lab.data(),
                                 a.mb.data(), 64u, a.topr.data(), nullptr, 0u, 0u);
            });
            DevState junk;
            memset(&junk, 0x5a, sizeof(junk));
            simt::run_grid(grid, 256, [&] {
                k_chunk_scan_local(cinfo.data(), 0u, b.pvl.data(), b.phl.data(), b.topv.data(), b.toph.data(), b.topr.data(), &b.st, capO_chunks, grid * 1024u);
            });
            simt::run_grid(1, 1024, [&] {
                k_chunk_scan_top(b.topv.data(), b.toph.data(), grid, b.pvl.data(), b.phl.data(), 0u, 0u, &b.st, &b.ctr, junk, b.lab.data(), b.mb.data(), 64u,
                                 b.topr.data(), nullptr, capO_chunks, grid * 1024u);
            });
            a.st.t_open = b.st.t_open = 0;
            bool same = memcmp(&a.st, &b.st, sizeof(a.st)) == 0 && a.st.n_o_read == n_read;
            for (uint32_t c = 0; same && c < nchunks; ++c)
                same = a.pvl[c] + a.topv[c >> 10] == b.pvl[c] + b.topv[c >> 10] && a.phl[c] + a.toph[c >> 10] == b.phl[c] + b.toph[c >> 10];
            CHECK(same, "two-level chunk scan ahead: %u chunks, %u of them VoI-resident", nchunks, nFchunks);
            if (nchunks <= 16384) {
                simt::run_grid(1, 1024, [&] {
                    k_chunk_scan_one(cinfo.data(), nchunks, c1.pvl.data(), c1.phl.data(), c1.topv.data(), c1.toph.data(), ntop, nFchunks, &c1.st, &c1.ctr, init,
                                     c1.lab.data(), c1.mb.data(), 64u, nullptr, 0u, 0u);
                });
                c1.st.t_open = 0;
                bool s1 = memcmp(&a.st, &c1.st, sizeof(a.st)) == 0;
                for (uint32_t c = 0; s1 && c < nchunks; ++c)
                    s1 = a.pvl[c] + a.topv[c >> 10] == c1.pvl[c] + c1.topv[c >> 10] && a.phl[c] + a.toph[c >> 10] == c1.phl[c] + c1.toph[c >> 10];
                CHECK(s1, "one-launch chunk scan == two-level: %u chunks, %u of them VoI-resident", nchunks, nFchunks);
            }
        }
    }
    printf("two-level chunk scan launched ahead == with the host's arguments (1 .. 70001 chunks)  %s\n", g_fail ? "FAILED" : "ok");
}

// Round 4: bin_key decides most points in float32 and hands the rest to the float64 restatement of the reference (bin_key_exact).
// Sound iff every point the fast path accepts gets the exact key: random points, points ON and next to every ring / sector boundary
// (constructed in float64, offsets from 1e-9 to 1e-3 of a cell), the axes, signed zeros, tiny and huge coordinates.
static void test_bin_key(std::mt19937 &rng) {
    Counters ca, cb;
    for (int cfg = 0; cfg < 3; ++cfg) {
        erasor_params p = seq05_params();
        if (cfg == 1) { p.num_rings = 15; p.num_sectors = 60; p.max_range = 60.0; }
        if (cfg == 2) { p.num_rings = 7; p.num_sectors = 31; p.max_range = 9.7; p.max_h = 1.0; p.min_h = -0.25; }
        const DP P = make_dp(p);
        memset(&ca, 0, sizeof(ca));
        memset(&cb, 0, sizeof(cb));
        size_t n = 0, bad = 0;
        auto one = [&](float x, float y, float z) {
            const uint32_t a = bin_key(P, x, y, z, &ca), b = bin_key_exact(P, x, y, z, &cb);
            ++n;
            if (a != b && ++bad < 5) printf("  bin_key(%a, %a, %a) = %u, exact %u\n", x, y, z, a, b);
        };
        std::uniform_real_distribution<double> U(-1.0, 1.0), U01(0.0, 1.0);
        const double R = p.max_range;
        for (int i = 0; i < 3000000; ++i) one((float)(U(rng) * R * 1.05), (float)(U(rng) * R * 1.05), (float)(U(rng) * 4.0));
        const double offs[] = {0.0, 1e-9, 1e-8, 1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3};
        for (int sidx = 0; sidx <= p.num_sectors; ++sidx)      // sector boundaries at many radii
            for (double o : offs)
                for (int sg = -1; sg <= 1; sg += 2)
                    for (int k = 0; k < 60; ++k) {
                        const double th = sidx * P.sector_size + sg * o * P.sector_size, r = U01(rng) * R * 1.01;
                        one((float)(r * cos(th)), (float)(r * sin(th)), 0.5f);
                    }
        for (int ridx = 0; ridx <= p.num_rings; ++ridx)        // ring boundaries at many angles
            for (double o : offs)
                for (int sg = -1; sg <= 1; sg += 2)
                    for (int k = 0; k < 300; ++k) {
                        const double th = U01(rng) * 2 * PI_REF, r = ridx * P.ring_size + sg * o * P.ring_size;
                        one((float)(r * cos(th)), (float)(r * sin(th)), 0.5f);
                    }
        const float sp[] = {0.f, -0.f, 1e-45f, -1e-45f, 1e-30f, -1e-30f, 1e-3f, -1e-3f, 1.f, -1.f, 39.99999f, 40.f, 40.00001f, -40.f,
                            79.99999f, 80.f, 80.00001f, -80.f, 1e10f, -1e10f, 1e20f, -1e20f, 3e38f, -3e38f};
        for (float x : sp)
            for (float y : sp)
                for (float z : {-1.3f, -1.29999f, 0.f, 3.19999f, 3.2f, 100.f}) one(x, y, z);
        CHECK(bad == 0, "bin_key: %zu of %zu points differ from the exact key (config %d)", bad, n, cfg);
        CHECK(ca.n_ambiguous == cb.n_ambiguous && ca.n_neg_sector == cb.n_neg_sector, "bin_key: counters %u %u vs %u %u", ca.n_ambiguous,
              ca.n_neg_sector, cb.n_ambiguous, cb.n_neg_sector);
    }
    printf("bin_key (float32 decision, float64 fallback) == the float64 key  %s\n", g_fail ? "FAILED" : "ok");
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::mt19937 rng(20210310);
    test_bin_key(rng);
    test_chunk_scan_ahead(rng);
    test_runs(rng);
    test_map_store(rng);
    test_map_bucketing(rng);
    test_per_bin(rng);
    printf("%s\n", g_fail ? "FAILED" : "ALL OK");
    return g_fail ? 1 : 0;
}
