"""Known-answer and invariant tests that pin the CPU oracle.

The reference ships no tests or golden vectors ("parity unpinned"), so each case below states the
reference rule it pins (file:line under /root/reference) and a hand-derived expectation.
"""
import ctypes as C
import math

import numpy as np
import pytest

from erasor_amd import synth
from oracle import orc

PI_REF = 3.1415926535  # erasor.h:4 (truncated on purpose)
I4 = np.eye(4, dtype=np.float32).reshape(16)


def params(**kw):
    p = orc.params_default()
    synth.apply_params(p, "05")
    for k, v in kw.items():
        setattr(p, k, v)
    return p


# ---------------------------------------------------------------------------------------------
# 1. bin index (erasor.cpp:11-21, 104-110; erasor.h:63-64)
# ---------------------------------------------------------------------------------------------
def test_bin_index_edges():
    p = params(min_h=-1.25, max_h=3.25)  # exactly representable gates
    S = p.num_sectors
    assert orc.bin_of(p, 60.0, 0.0, 0.0) == 14 * S + 0          # r == max_r is inside (<=), ring clamped to R-1
    assert orc.bin_of(p, np.nextafter(np.float32(60), np.float32(61)), 0.0, 0.0) == -1
    assert orc.bin_of(p, 1.0, 0.0, -1.25) == -1                  # z == min_h excluded (strict)
    assert orc.bin_of(p, 1.0, 0.0, np.nextafter(np.float32(-1.25), np.float32(0))) == 0
    assert orc.bin_of(p, 1.0, 0.0, 3.25) == -1                   # z == max_h excluded (strict)
    assert orc.bin_of(p, 1.0, 0.0, np.nextafter(np.float32(3.25), np.float32(0))) == 0
    assert orc.bin_of(p, 0.0, 10.0, 0.0) == 2 * S + 15           # +y axis: theta = pi/2 -> sector 15 (60 sectors)
    assert orc.bin_of(p, -10.0, 0.0, 0.0) == 2 * S + 30          # -x axis with y = +0: theta = pi -> sector 30
    assert orc.bin_of(p, 10.0, -1e-30, 0.0) == 2 * S + 59        # y -> 0-: theta -> 2*PI_REF, clamped to S-1
    assert orc.bin_of(p, 3.9999, 0.0, 0.0) == 0 and orc.bin_of(p, 4.0, 0.0, 0.0) == 1 * S  # ring_size = 60/15 = 4


def test_truncated_pi_constant_is_used():
    # with the true pi, theta(-x axis)/sector_size would be exactly 30; PI_REF < pi makes it 30.0000000009
    q = orc.xy2theta(-1.0, 0.0) / (2 * PI_REF / 60)
    assert q > 30.0 and q - 30.0 < 1e-8
    assert orc.xy2theta(1.0, -1.0) == 2 * PI_REF + math.atan2(-1.0, 1.0)


def test_negative_zero_hazard_is_a_counted_clamp():
    # y == -0.0f, x < 0: `y >= 0` is true, atan2(-0, x<0) = -pi -> sector = -30 -> vector::at throws in the reference
    p = params()
    assert orc.bin_of(p, -10.0, -0.0, 0.0) == 2 * p.num_sectors + 0
    # inside a step the hazard is all but unreachable: pcl::transformPointCloud's `... + T03` turns -0.0 into +0.0
    o = orc.Oracle(p)
    o.set_map(np.array([[-10.0, -0.0, 0.0, 40.0]], np.float32))
    r = o.step(np.zeros((0, 4), np.float32), I4, I4, I4)
    assert r.n_neg_sector == 0 and r.n_voi == 1
    code, _ = o.get_voi_codes()
    assert code[0] == 2 * p.num_sectors + 30


# ---------------------------------------------------------------------------------------------
# helpers: a one-bin world.  R-POD 1 ring x 4 sectors, points placed in sector 0 (x>0, small y>0)
# ---------------------------------------------------------------------------------------------
def one_bin_params(**kw):
    d = dict(max_range=10.0, num_rings=1, num_sectors=4, min_h=-5.0, max_h=5.0, minimum_num_pts=3,
             scan_ratio_threshold=0.3, query_voxel_size=0.05, map_voxel_size=0.05, gf_num_lpr=2, num_lowest_pts=0)
    d.update(kw)
    return params(**d)


def column(n, z0, z1, x0=2.0, label=40.0, dy=0.3):
    """n points in sector 0, z spread linearly in [z0, z1], > leaf apart so voxelisation keeps them"""
    z = np.linspace(z0, z1, n) if n > 1 else np.array([z0])
    pts = np.stack([x0 + 0.31 * np.arange(n), np.full(n, dy), z, np.full(n, label)], 1)
    return pts.astype(np.float32)


def run(p, m, s):
    o = orc.Oracle(p)
    o.set_map(m)
    r = o.step(s, I4, I4, I4)
    return o, r


# ---------------------------------------------------------------------------------------------
# 3. Scan Ratio Test edge cases (erasor.cpp:448-486)
# ---------------------------------------------------------------------------------------------
def test_srt_equal_heights_merge():
    o, r = run(one_bin_params(), column(5, 0.0, 1.0), column(5, 0.0, 1.0, dy=0.6))
    assert o.get_status()[0] == 0.25 and r.n_reverted_bins == 0 and r.n_map_out == 5  # v3: scan points are NOT merged in


def test_srt_one_zero_height_is_dynamic_and_reverts():
    # curr flat (diff 0) -> ratio 0 < thr; map diff 1.0 >= 0 -> MAP_IS_HIGHER; 1.0 > 0.5 -> R-GPF runs
    o, r = run(one_bin_params(), column(6, 0.0, 1.0), column(4, 0.0, 0.0, dy=0.6))
    assert o.get_status()[0] == 0.5 and r.n_reverted_bins == 1


def test_srt_both_zero_height_is_nan_and_merges():
    o, r = run(one_bin_params(), column(4, 0.25, 0.25), column(4, 0.5, 0.5, dy=0.6))
    assert o.get_status()[0] == 0.25  # min(0/0, 0/0) = NaN; NaN < thr is false -> MERGE_BINS


def test_srt_minimum_num_pts_boundary():
    p = one_bin_params(minimum_num_pts=4)
    o, _ = run(p, column(6, 0.0, 1.0), column(3, 0.0, 0.0, dy=0.6))
    assert o.get_status()[0] == 0.0                      # 3 < 4 -> LITTLE_NUM
    o, r = run(p, column(6, 0.0, 1.0), column(4, 0.0, 0.0, dy=0.6))
    assert o.get_status()[0] == 0.5 and r.n_reverted_bins == 1


def test_scan_points_in_bins_without_map_points_are_dropped():
    m = column(5, 0.0, 1.0)
    s = np.array([[-2.0, -0.5, 0.0, 40.0], [-2.4, -0.5, 1.0, 40.0], [-2.8, -0.5, 0.5, 40.0]], np.float32)  # other sector
    o, r = run(one_bin_params(), m, s)
    assert r.n_map_out == 5 and r.n_query == 3 and np.array_equal(o.get_map(), m)


def test_curr_is_higher_keeps_map_bin():
    o, r = run(one_bin_params(), column(5, 0.0, 0.1), column(5, 0.0, 2.0, dy=0.6))
    assert o.get_status()[0] == 1.0 and r.n_reverted_bins == 0 and r.n_map_out == 5


# ---------------------------------------------------------------------------------------------
# 4. v3 revert gate is the hard-coded `> 0.5` (erasor.cpp:511); v2 uses bin_map.max_h > th_bin_max_h (:383)
# ---------------------------------------------------------------------------------------------
def test_v3_gate_exactly_half_metre_is_not_reverted():
    p = one_bin_params()
    o, r = run(p, column(6, 0.0, 0.5), column(4, 0.0, 0.0, dy=0.6))
    assert r.n_reverted_bins == 0 and o.get_status()[0] == 0.0   # MAP_IS_HIGHER but diff == 0.5 -> NOT_ASSIGNED
    hi = float(np.nextafter(np.float32(0.5), np.float32(1)))
    o, r = run(p, column(6, 0.0, hi), column(4, 0.0, 0.0, dy=0.6))
    assert r.n_reverted_bins == 1


def test_v2_gate_uses_th_bin_max_h_and_merges_scan_points():
    p = one_bin_params(version=2, th_bin_max_h=0.75)
    o, r = run(p, column(6, 0.0, 0.7), column(4, 0.0, 0.0, dy=0.6))      # map.max_h 0.7 <= 0.75 -> keep map bin
    assert r.n_reverted_bins == 0 and r.n_map_out == 6
    o, r = run(p, column(6, 0.0, 0.8), column(4, 0.0, 0.0, dy=0.6))      # 0.8 > 0.75 -> curr + ground(map)
    assert r.n_reverted_bins == 1
    o, r = run(p, column(5, 0.0, 1.0), column(5, 0.0, 1.0, dy=0.6))      # ratio 1 -> merge: curr THEN map (erasor.cpp:296-307)
    assert r.n_map_out == 10
    assert np.array_equal(o.get_map()[:5], column(5, 0.0, 1.0, dy=0.6)) and np.array_equal(o.get_map()[5:], column(5, 0.0, 1.0))


# ---------------------------------------------------------------------------------------------
# 5. R-GPF (erasor.cpp:204-294)
# ---------------------------------------------------------------------------------------------
def test_rgpf_few_points_lpr_zero_and_degenerate_plane():
    p = one_bin_params(num_lowest_pts=5, gf_num_lpr=10)
    # M = 3 <= num_lowest_pts -> lpr = 0 -> seeds: z < 0.5 -> none (all z >= 1) -> empty cloud into estimate_plane_
    pts = column(3, 1.0, 2.0)
    mask, normals, ds, ndeg = orc.extract_ground(p, pts)
    assert ndeg >= 1
    # defined behaviour: cov = 0 -> U = I -> normal (0,0,1), d = 0, threshold gf_dist_thr: nothing below 0.15
    assert np.array_equal(normals[0], [0, 0, 1]) and ds[0] == 0.0 and not mask.any()


@pytest.mark.parametrize("rng_m", [5.0, 30.0, 70.0])
def test_rgpf_recovers_a_tilted_plane(rng_m):
    p = params(gf_dist_thr=0.15)
    rng = np.random.default_rng(int(rng_m))
    n = 800
    x = rng_m + rng.uniform(0, 4, n)
    y = rng.uniform(-2, 2, n)
    z = 0.02 * x - 0.01 * y + rng.normal(0, 0.01, n)
    box = np.stack([rng_m + rng.uniform(1, 2, 100), rng.uniform(-0.5, 0.5, 100), 0.02 * rng_m + rng.uniform(0.4, 1.6, 100)], 1)
    pts = np.concatenate([np.stack([x, y, z], 1), box], 0)
    pts = np.concatenate([pts, np.full((len(pts), 1), 40.0)], 1).astype(np.float32)
    mask, normals, ds, ndeg = orc.extract_ground(p, pts)
    assert ndeg == 0
    assert mask[:n].mean() > 0.99 and mask[n:].sum() == 0           # ground recovered, box rejected
    # float64 fit of the final ground set bounds the float32 restatement
    g = pts[mask][:, :3].astype(np.float64)
    mean = g.mean(0)
    u, s, vt = np.linalg.svd(np.cov((g - mean).T))
    n64 = u[:, 2] * np.sign(u[2, 2])
    # the last recorded plane was fitted on the previous iteration's ground set; refit on that set is what we compare:
    assert abs(np.dot(normals[-1], n64)) > 1 - 1e-4
    assert normals[-1][2] > 0.99                                       # Jacobi keeps the normal pointing +z


def test_rgpf_outputs_are_in_source_order_and_nonground_only_from_last_iteration():
    p = params()
    rng = np.random.default_rng(3)
    pts = np.stack([10 + rng.uniform(0, 4, 300), rng.uniform(-2, 2, 300), rng.normal(0, 0.02, 300), np.full(300, 40.0)], 1).astype(np.float32)
    pts[::7, 2] += 1.0
    mask, normals, ds, _ = orc.extract_ground(p, pts)
    assert normals.shape == (3, 3) and len(ds) == 3
    assert mask.sum() + (~mask).sum() == 300 and (~mask)[::7].all()


# ---------------------------------------------------------------------------------------------
# third-party restatements: PCL covariance, Eigen JacobiSVD (bounded by float64 references)
# ---------------------------------------------------------------------------------------------
def test_mean_and_cov_is_single_pass_float32():
    rng = np.random.default_rng(0)
    c = np.concatenate([rng.normal([30, 5, 0.2], [1.0, 1.0, 0.02], (500, 3)), np.zeros((500, 1))], 1).astype(np.float32)
    cov, mean = orc.mean_and_cov(c)
    # restated by hand: nine sequential float32 accumulators, /n, E[ab]-E[a]E[b]
    a = np.zeros(9, np.float32)
    for x, y, z, _ in c:
        a += np.array([x * x, x * y, x * z, y * y, y * z, z * z, x, y, z], np.float32)
    a = a / np.float32(len(c))
    want = np.array([[a[0] - a[6] * a[6], a[1] - a[6] * a[7], a[2] - a[6] * a[8]],
                     [a[1] - a[6] * a[7], a[3] - a[7] * a[7], a[4] - a[7] * a[8]],
                     [a[2] - a[6] * a[8], a[4] - a[7] * a[8], a[5] - a[8] * a[8]]], np.float32)
    assert np.array_equal(cov, want) and np.array_equal(mean[:3], a[6:9]) and mean[3] == 1.0
    assert np.allclose(cov, np.cov(c[:, :3].astype(np.float64).T, bias=True), atol=2e-3)


def test_jacobi_svd_matches_float64_svd():
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.normal(size=(3, 3)) * rng.uniform(0.01, 3)
        cov = (a @ a.T).astype(np.float32)
        U, sv = orc.jacobi_svd3(cov)
        u64, s64, _ = np.linalg.svd(cov.astype(np.float64))
        assert np.all(np.diff(sv) <= 0)                                           # descending
        assert np.allclose(sv, s64, rtol=2e-5, atol=1e-6 * s64[0])
        assert np.allclose(U @ U.T, np.eye(3), atol=1e-5)
        if s64[1] - s64[2] > 1e-3 * s64[0]:
            assert abs(np.dot(U[:, 2], u64[:, 2])) > 1 - 1e-4
    U, sv = orc.jacobi_svd3(np.zeros((3, 3), np.float32))                      # zero matrix: U = I, sv = 0
    assert np.array_equal(U, np.eye(3, dtype=np.float32)) and not sv.any()
    U, sv = orc.jacobi_svd3(np.diag([1.0, 4.0, 0.25]).astype(np.float32))     # already diagonal: pure sorting
    assert np.array_equal(sv, [4.0, 1.0, 0.25]) and np.array_equal(U[:, 2], [0, 0, 1]) and np.array_equal(U[:, 0], [0, 1, 0])


# ---------------------------------------------------------------------------------------------
# 6. voxelize_preserving_labels (utils.cpp:80-114; PCL 1.8 VoxelGrid)
# ---------------------------------------------------------------------------------------------
def test_voxelize_label_is_nearest_input_label_never_an_average():
    c = np.array([[0.01, 0.01, 0.01, 40.0], [0.19, 0.01, 0.01, 252.0 + (7 << 16)], [0.18, 0.02, 0.01, 252.0 + (7 << 16)],
                  [1.01, 0.0, 0.0, 50.0]], np.float32)
    out = orc.voxelize_preserving_labels(c, 0.2)
    assert len(out) == 2
    labels = set(out[:, 3].tolist())
    assert labels <= {40.0, 252.0 + (7 << 16), 50.0}
    # centroid of voxel 0 = float32 running sum in std::sort order / 3
    pts, pi, idx, ovf = orc.voxel_grid(c, 0.2)
    order = pi[idx == idx.min()]
    s = np.zeros(3, np.float32)
    for k in order:
        s = s + c[k, :3]
    assert np.array_equal(pts[0, :3], s / np.float32(3)) and not ovf
    assert out[0, 3] == 252.0 + (7 << 16)   # the centroid (0.1267,..) is closest to the 0.18 point


def test_voxelize_single_point_voxels_are_identity_and_sorted_by_voxel_index():
    rng = np.random.default_rng(5)
    c = np.concatenate([rng.uniform(-20, 20, (400, 3)), rng.integers(0, 300, (400, 1))], 1).astype(np.float32)
    out = orc.voxelize_preserving_labels(c, 0.05)   # every point alone in its voxel (800^3 cells < INT_MAX)
    assert len(out) == 400
    assert np.array_equal(np.sort(out.view([("", np.float32)] * 4), axis=0), np.sort(c.view([("", np.float32)] * 4), axis=0))
    # output order = ascending voxel index i + j*dx + k*dx*dy: z layer is the major key, then y, then x
    inv = np.float32(1.0) / np.float32(0.05)
    ijk = [np.floor(out[:, a] * inv) for a in range(3)]
    key = (ijk[2] - ijk[2].min()) * 1e8 + (ijk[1] - ijk[1].min()) * 1e4 + (ijk[0] - ijk[0].min())
    assert np.all(np.diff(key) > 0)


def test_voxelize_index_overflow_returns_the_input():
    c = np.array([[0, 0, 0, 1.0], [3000, 3000, 3000, 2.0], [1, 1, 1, 3.0]], np.float32)
    out = orc.voxelize_preserving_labels(c, 0.001)   # (3e6)^3 > INT_MAX -> "output = *input_"
    assert np.array_equal(out, c)


def test_std_sort_is_the_unstable_libstdcxx_one():
    # 17+ equal keys: introsort's partition moves them; a stable sort would keep 0..n-1
    k = np.zeros(40, np.uint32)
    _, v = orc.std_sort_u32(k, np.arange(40, dtype=np.uint32))
    assert sorted(v.tolist()) == list(range(40)) and v.tolist() != list(range(40))


# ---------------------------------------------------------------------------------------------
# 7. assembly order + conservation (erasor.cpp:311-312,616,622; OMU.cpp:281,290,431-433)
# ---------------------------------------------------------------------------------------------
def test_assembly_order_and_conservation_on_a_synthetic_sequence():
    from scenarios import small
    sc = small()
    o = orc.Oracle(sc["params"])
    o.set_map(sc["map"])
    prev = sc["map"]
    for f in range(3):
        r = o.step(sc["scans"][f], sc["T_l2b"], sc["T_b2o"][f], sc["T_o2b"][f])
        m = o.get_map()
        assert r.n_map_in == r.n_voi + r.n_outskirts                      # fetch_VoI conservation (OMU.cpp:431-433)
        assert r.n_map_out == r.n_static_estimate + r.n_complement + r.n_outskirts
        se, co = o.get_cloud(2), o.get_cloud(3)
        head = orc.transform(np.concatenate([se, co], 0), sc["T_b2o"][f])  # body2origin (OMU.cpp:286)
        assert np.array_equal(m[: len(head)].view(np.uint32), head.view(np.uint32))
        # outskirts: untouched points, original relative order, at the tail
        xc, yc = float(sc["T_b2o"][f][3]), float(sc["T_b2o"][f][7])
        d2 = (prev[:, 0].astype(np.float64) - xc) ** 2 + (prev[:, 1].astype(np.float64) - yc) ** 2
        outs = prev[~(d2 < 60.0 ** 2)]
        assert np.array_equal(m[len(head):].view(np.uint32), outs.view(np.uint32))
        # ground of reverted bins appears twice: voxelised inside the bin and raw in ground_viz (erasor.cpp:523,531,616)
        gv = o.get_cloud(6)
        assert r.n_ground == len(gv) and np.array_equal(se[len(se) - len(gv):], gv)
        # rejected U ground = the reverted bins' map points
        cnt, _, _ = o.get_bins(0)
        bins, _, _ = o.get_planes()
        assert cnt[bins].sum() == r.n_ground + r.n_map_rejected
        ns, nd = (~synth.is_dynamic(m[:, 3])).sum(), synth.is_dynamic(m[:, 3]).sum()
        assert (r.n_static, r.n_dynamic) == (ns, nd)
        prev = m


def test_is_dynamic_obj_close_wrap_uses_num_rings():
    # erasor.cpp:578,580: j<0 -> j+num_rings, j>=num_sectors -> j-num_rings (a reference bug, kept: status parity only)
    p = params(max_range=8.0, num_rings=2, num_sectors=8, min_h=-5.0, max_h=5.0, minimum_num_pts=2, scan_ratio_threshold=0.3,
               query_voxel_size=0.05)

    def col(sector, n, z1, dy):
        ang = (sector + 0.5) * 2 * PI_REF / 8
        r = 1.0 + 0.2 * np.arange(n)
        return np.stack([r * np.cos(ang) + dy, r * np.sin(ang), np.linspace(0, z1, n), np.full(n, 40.0)], 1).astype(np.float32)

    # sector 0: merge candidate; sector 1 (= -1 + num_rings): CURR_IS_HIGHER; true neighbour sector 7 is NOT looked at
    m = np.concatenate([col(0, 4, 1.0, 0.0), col(1, 4, 0.1, 0.0), col(7, 4, 0.1, 0.0)], 0)
    s = np.concatenate([col(0, 4, 1.0, 0.01), col(1, 4, 2.0, 0.01), col(7, 4, 0.1, 0.01)], 0)
    o, _ = run(p, m, s)
    st = o.get_status()
    assert st[0 * 8 + 1] == 1.0 and st[0 * 8 + 0] == 0.8     # BLOCKED through the (correct) +1 neighbour
    m2 = np.concatenate([col(0, 4, 1.0, 0.0), col(7, 4, 0.1, 0.0)], 0)
    s2 = np.concatenate([col(0, 4, 1.0, 0.01), col(7, 4, 2.0, 0.01)], 0)
    o, _ = run(p, m2, s2)
    st = o.get_status()
    assert st[7] == 1.0 and st[0] == 0.25                     # sector 7 dynamic, but theta=-1 wraps to 1 (num_rings=2): not seen


# ---------------------------------------------------------------------------------------------
# 9. label decode (utils.cpp:3,64-65) and pose -> matrix (utils.cpp:35-55)
# ---------------------------------------------------------------------------------------------
def test_label_decode_numeric_cast_low_16_bits():
    lab = np.array([252, 259, 251, 260, 252 + (5 << 16), 40 + (252 << 16), 10], np.float32)
    assert synth.is_dynamic(lab).tolist() == [True, True, False, False, True, False, False]
    m = np.stack([np.arange(7) * 100.0 + 1000, np.zeros(7), np.zeros(7), lab], 1).astype(np.float32)  # all outskirts
    o = orc.Oracle(params())
    o.set_map(m)
    r = o.step(np.zeros((0, 4), np.float32), I4, I4, I4)
    assert (r.n_static, r.n_dynamic) == (4, 3)


def test_geopose2eigen_and_transform_formula():
    T = orc.geopose2eigen([1.5, -2.0, 0.25, 0, 0, math.sin(0.3), math.cos(0.3)]).reshape(4, 4)
    assert np.allclose(T[:3, :3], [[math.cos(0.6), -math.sin(0.6), 0], [math.sin(0.6), math.cos(0.6), 0], [0, 0, 1]], atol=1e-7)
    assert T[:3, 3].tolist() == [1.5, -2.0, 0.25] and T[3].tolist() == [0, 0, 0, 1]
    p = np.array([[3.0, 4.0, 5.0, 77.0]], np.float32)
    q = orc.transform(p, T.reshape(16))[0]
    t = T.astype(np.float32)
    want = [np.float32(np.float32(np.float32(t[r, 0] * p[0, 0] + t[r, 1] * p[0, 1]) + t[r, 2] * p[0, 2]) + t[r, 3]) for r in range(3)]
    assert q[:3].tolist() == [float(w) for w in want] and q[3] == 77.0
    Ti = orc.invert4(T.reshape(16)).reshape(4, 4)
    assert np.allclose(Ti.astype(np.float64) @ T.astype(np.float64), np.eye(4), atol=1e-6)


def test_unsupported_version_is_rejected():
    p = params(version=4)   # OMU.cpp:273-275 "Other version is not implemented!"
    o = orc.Oracle(p)
    o.set_map(np.zeros((1, 4), np.float32))
    with pytest.raises(RuntimeError):
        o.step(np.zeros((0, 4), np.float32), I4, I4, I4)


# ---------------------------------------------------------------------------------------------
# large-scale (submap) mode: OMU.cpp:332-379
# ---------------------------------------------------------------------------------------------
def test_large_scale_submap_semantics():
    from scenarios import small
    sc = small()
    p = params(is_large_scale=1, submap_size=25.0)
    o = orc.Oracle(p)
    o.set_map(sc["map"])
    n0 = len(sc["map"])
    x0, y0 = float(sc["T_b2o"][0][3]), float(sc["T_b2o"][0][7])
    r = o.step(sc["scans"][0], sc["T_l2b"], sc["T_b2o"][0], sc["T_o2b"][0])
    m = sc["map"]
    box = (np.abs(x0 - m[:, 0].astype(np.float64)) < 25.0) & (np.abs(y0 - m[:, 1].astype(np.float64)) < 25.0)   # full submap_size, not half
    assert r.n_map_in == box.sum()                                   # map_arranged_ is the submap
    full = o.get_map()                                               # save = submap + complement (OMU.cpp:181)
    assert len(full) == r.n_map_out + (n0 - box.sum())
    assert np.array_equal(full[r.n_map_out:], m[~box])               # complement: untouched, original order
    # re-centring only after moving more than submap_size / 2 (OMU.cpp:342-345)
    sizes = []
    for f in range(1, 12):
        r = o.step(sc["scans"][f], sc["T_l2b"], sc["T_b2o"][f], sc["T_o2b"][f])
        sizes.append(o.map_size())
        moved = max(abs(float(sc["T_b2o"][f][3]) - x0), abs(float(sc["T_b2o"][f][7]) - y0))
        if moved > 12.5:
            break
    assert moved > 12.5, "scenario does not travel far enough to re-centre the submap"
    assert r.n_static + r.n_dynamic == r.n_map_out                   # counters are over the submap only


# ---------------------------------------------------------------------------------------------
# mapgen restatement (oracle/orc.py:Mapgen, src/mapgen/mapgen.hpp:198-305)
# ---------------------------------------------------------------------------------------------
def test_mapgen_self_filter_boundary_and_transform_order():
    """mapgen.hpp:219-228: a point is dropped iff pow(x,2)+pow(y,2) (double) < (float)pow(2.7,2); z is ignored.
    :231-237: lidar->origin-of-body (z + 1.73) THEN the pose."""
    thr = float(np.float32(2.7 ** 2))  # 7.28999996185302734375
    inside = np.float32(np.sqrt(thr) * (1 - 1e-6))
    outside = np.float32(np.sqrt(thr) * (1 + 1e-6))
    scan = np.array([[inside, 0, 50, 1], [0, inside, -50, 2], [outside, 0, 0, 3], [0, -outside, 0, 4], [2.0, 2.0, 0, 5],  # 8 >= thr: kept
                     [1.9, 1.9, 0, 6]], np.float32)                                                                      # 7.22 < thr: dropped
    assert float(inside) ** 2 < thr <= float(outside) ** 2
    T = orc.geopose2eigen([10.0, -5.0, 0.5, 0, 0, 0.7071068, 0.7071068])  # yaw +90 deg
    g = orc.Mapgen(0.2)
    n = g.accum(scan, np.asarray(T, np.float32).reshape(4, 4))
    kept = g.cloud_curr[np.argsort(g.cloud_curr[:, 3])]
    assert n == 3 and kept[:, 3].tolist() == [3.0, 4.0, 5.0]
    # label 3: (outside, 0, 0) -> z + 1.73 -> rotate +90 deg about z (x -> y), then translate
    assert np.allclose(kept[0, :3], [10.0, -5.0 + float(outside), 0.5 + 1.73], atol=1e-5)
    assert np.allclose(kept[2, :3], [10.0 - 2.0, -5.0 + 2.0, 0.5 + 1.73], atol=1e-5)


def test_mapgen_large_scale_submap_counter():
    """mapgen.hpp:241-256: the first scan initialises the map; from the second on `cnt_voxel++ % 500 == 0` closes a submap,
    i.e. right at the second scan and then every 500 accumulated scans; saveNaiveMap concatenates submaps + remainder."""
    rng = np.random.default_rng(0)
    I = np.eye(4, dtype=np.float32)
    g = orc.Mapgen(0.5, is_large_scale=True)
    scans = []
    for k in range(4):
        s = np.zeros((200, 4), np.float32)
        s[:, :2] = rng.uniform(5, 30, (200, 2))
        s[:, 2] = rng.uniform(-1, 1, 200)
        s[:, 3] = 40 + k
        scans.append(s)
        g.accum(s, I)
        if k == 0:
            assert len(g.cloud_maps) == 0 and len(g.cloud_map) == len(g.cloud_curr)
        elif k == 1:
            assert len(g.cloud_maps) == 1 and len(g.cloud_map) == 0   # cnt_voxel == 0: re-voxelised at leafsize, pushed, cleared
        else:
            assert len(g.cloud_maps) == 1 and len(g.cloud_map) > 0
    first_two = np.concatenate([orc.voxelize_preserving_labels(orc.transform(orc.transform(s, np.array(
        [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1.73], [0, 0, 0, 1]], np.float32)), I), 0.2) for s in scans[:2]])
    assert np.array_equal(g.cloud_maps[0], orc.voxelize_preserving_labels(first_two, 0.5))
    assert len(g.naive_map()) == len(g.cloud_maps[0]) + len(g.cloud_map)
    small = orc.Mapgen(0.5, is_large_scale=False)
    for s in scans:
        small.accum(s, I)
    assert len(small.cloud_maps) == 0 and small.accum_count == 3 and len(small.save()) <= len(small.cloud_map)
