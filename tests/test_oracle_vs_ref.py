"""Pins the CPU oracle (oracle/erasor_oracle.cpp) to the reference's OWN source text.

oracle/_ref/liberasor_ref.so is /root/reference's erasor.cpp, erasor_utils.cpp, OfflineMapUpdater.cpp and
mapgen.hpp compiled UNMODIFIED (oracle/ref.mk) against stand-in ros/pcl/Eigen/tf headers (oracle/stubs/).  Every
test here drives the reference's real objects — erasor::OfflineMapUpdater through its ROS callback, class ERASOR,
erasor_utils::*, class mapgen — and demands bit-identical results from the oracle on every scenario the GPU suite
uses.  What remains restated (not pinned by source) is exactly oracle/third_party_restated.h, which both sides share:
PCL VoxelGrid / covariance / transform, Eigen JacobiSVD / products / 4x4 inverse, tf quaternion->matrix; the 1-NN of
voxelize_preserving_labels is implemented twice (exact kd-tree in the stub, voxel-grid search in the oracle).
"""
import ctypes as C

import numpy as np
import pytest

import scenarios
from erasor_amd import synth
from oracle import orc, ref

pytestmark = pytest.mark.skipif(not (ref.available() or ref.build()), reason="oracle/_ref not built and /root/reference absent")

L2B = [0, 0, synth.LIDAR_HEIGHT, 0, 0, 0, 1]
ID7 = [0, 0, 0, 0, 0, 0, 1]
I4 = np.eye(4, dtype=np.float32).reshape(16)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def same(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def compare_state(o, r, ro, tag=""):
    """everything the reference exposes after one callback_node vs the oracle"""
    for which in range(8):  # query_voi, map_voi, static_estimate, complement, map_rejected, curr_rejected, ground_viz, map
        assert same(o.get_cloud(which), r.get_cloud(which)), (tag, "cloud", which)
    R, S = o.params.num_rings, o.params.num_sectors
    st = o.get_status()
    like = r.polygon_likelihood()
    if o.params.version == 3:
        # r_pod_selected[r][theta].status (erasor.h:30) and the published SRT polygons (erasor.cpp:496-563, theta-major)
        assert np.array_equal(bits(st), bits(r.get_status())), (tag, "status")
        assert np.array_equal(like, st.reshape(R, S).T.reshape(-1).astype(np.float32)), (tag, "polygon likelihood")
    else:
        # v2 never assigns Bin::status (bin_merged is even left uninitialised, erasor.cpp:420): the SRT outcome only
        # exists as the likelihood of the polygons it pushes (erasor.cpp:345-425): one per bin unless the bin passed the
        # min-points test without both sides occupied
        cc, _, _ = o.get_bins(1)
        mc, _, _ = o.get_bins(0)
        pushed = (cc < o.params.minimum_num_pts) | ((cc > 0) & (mc > 0))
        order = np.arange(R * S).reshape(R, S).T.reshape(-1)
        assert np.array_equal(like, st[order][pushed[order]].astype(np.float32)), (tag, "v2 polygon likelihood")
    for which in (0, 1):  # r_pod_map, r_pod_curr: count, min_h, max_h
        for a, b in zip(o.get_bins(which), r.get_bins(which)):
            assert np.array_equal(a, b), (tag, "bins", which)
    _, normal, d = o.get_planes()
    rn, rd, rth = r.get_planes()
    assert same(normal.reshape(-1, 3), rn), (tag, "plane normals")
    if d.size:
        assert d.reshape(-1)[-1] == rd and o.params.gf_dist_thr - rd == rth, (tag, "plane d")
    assert r.label_counts() == (ro.n_static, ro.n_dynamic), (tag, "label counters")
    assert len(r.get_cloud(100)) == ro.n_outskirts


def run_sequence(sc, steps, over=None, l2b=L2B):
    p = orc.Params()
    C.memmove(C.byref(p), C.byref(sc["params"]), C.sizeof(p))
    for k, v in (over or {}).items():
        setattr(p, k, v)
    o = orc.Oracle(p)
    o.set_map(sc["map"])
    r = ref.RefUpdater(p, sc["map"], l2b)
    tot_rev = tot_rej = 0
    for k in range(steps):
        ro = o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
        r.step(sc["scans"][k], sc["poses"][k])
        Tl, Tb = r.get_matrices()
        assert same(Tl, sc["T_l2b"]) and same(Tb, sc["T_b2o"][k])  # geoPose2eigen (utils.cpp:35-55) as the reference runs it
        compare_state(o, r, ro, tag=(sc["seq"], k))
        # the dynamic-point mask as indices: the reference only has the rejected cloud; tie the two together
        tot_rev += ro.n_reverted_bins
        tot_rej += ro.n_map_rejected
    return o, r, tot_rev, tot_rej


# ---------------------------------------------------------------------------------------------
# whole steps through OfflineMapUpdater::callback_node (OMU.cpp:203-330)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seq,version,steps", [("05", 3, 10), ("00", 3, 4), ("07", 3, 4), ("01", 3, 3), ("02", 3, 3),
                                               ("05", 2, 4), ("00", 2, 3), ("ouster", 3, 3), ("large_scale_05", 3, 3)])
def test_sequences_bit_identical_to_reference_source(seq, version, steps):
    sc = scenarios.small(seq=seq, version=version)
    _, _, rev, rej = run_sequence(sc, steps)
    if seq in ("05", "00", "07"):
        assert rev > 0 and rej > 0  # R-GPF, per-bin voxelisation and the rejected cloud were really exercised


def test_ouster128_shape():
    sc = scenarios.small(seq="ouster", lidar="ouster128", az=2048, n_frames=4, length=120.0)
    run_sequence(sc, 3, l2b=L2B)


@pytest.mark.parametrize("submap_size", [160.0, 25.0, 8.0, 500.0])
def test_large_scale_submap_mode(submap_size):
    """reassign_submap / set_submap (OMU.cpp:332-379) incl. the function-static half_size; get_cloud(7) = submap + complement"""
    sc = scenarios.small()
    o, r, _, _ = run_sequence(sc, 8, over={"is_large_scale": 1, "submap_size": submap_size})
    assert same(o.get_map(), r.get_map())
    n = C.c_size_t(0)
    orc.lib().orc_submap_size(o.h, C.byref(n))
    assert n.value == len(r.get_cloud(101))


@pytest.mark.parametrize("rings,sectors", [(40, 120), (100, 130), (8, 60)])
def test_other_rpod_grids(rings, sectors):
    sc = scenarios.small()
    run_sequence(sc, 3, over={"num_rings": rings, "num_sectors": sectors})


def test_save_static_map_matches():
    """save_static_map (OMU.cpp:174-196): whole-map voxelize_preserving_labels of map_arranged_"""
    sc = scenarios.small()
    o, r, _, _ = run_sequence(sc, 3)
    want = r.save_static_map(0.2)
    got = orc.voxelize_preserving_labels(o.get_map(), np.float32(0.2))
    assert same(got, want)


def test_revisit_and_zero_scan():
    sc = scenarios.small()
    p = sc["params"]
    o = orc.Oracle(p)
    o.set_map(sc["map"])
    r = ref.RefUpdater(p, sc["map"], L2B)
    order = [0, 0, 5, 1, 0]
    for k in order:
        ro = o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
        r.step(sc["scans"][k], sc["poses"][k])
        compare_state(o, r, ro, tag=("revisit", k))
    empty = np.zeros((0, 4), np.float32)
    ro = o.step(empty, sc["T_l2b"], sc["T_b2o"][2], sc["T_o2b"][2])
    r.step(empty, sc["poses"][2])
    compare_state(o, r, ro, tag="empty scan")


def test_rejected_indices_agree_with_the_reference_rejected_cloud():
    """the oracle's dynamic-point mask (indices into the pre-step map) reproduces the reference's map_rejected_ cloud:
    map_rejected_ = T_b2o * (T_o2b * map_in[idx]) (OMU.cpp:435-437, 284-287)"""
    sc = scenarios.small()
    p = sc["params"]
    o = orc.Oracle(p)
    o.set_map(sc["map"])
    r = ref.RefUpdater(p, sc["map"], L2B)
    for k in range(4):
        before = o.get_map()
        o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
        r.step(sc["scans"][k], sc["poses"][k])
        idx = o.get_rejected_indices().astype(np.int64)
        assert len(idx) > 0
        via_idx = orc.transform(orc.transform(before[idx], sc["T_o2b"][k]), sc["T_b2o"][k])
        assert same(via_idx, r.get_cloud(4))


# ---------------------------------------------------------------------------------------------
# class ERASOR on egocentric clouds: one-bin known answers + reference exception behaviour
# ---------------------------------------------------------------------------------------------
def one_bin_params(**kw):
    p = orc.params_default()
    synth.apply_params(p, "05")
    d = dict(max_range=10.0, num_rings=1, num_sectors=4, min_h=-5.0, max_h=5.0, minimum_num_pts=3,
             scan_ratio_threshold=0.3, query_voxel_size=0.05, map_voxel_size=0.05, gf_num_lpr=2, num_lowest_pts=0)
    d.update(kw)
    for k, v in d.items():
        setattr(p, k, v)
    return p


def column(n, z0, z1, x0=2.0, label=40.0, dy=0.3):
    z = np.linspace(z0, z1, n) if n > 1 else np.array([z0])
    return np.stack([x0 + 0.31 * np.arange(n), np.full(n, dy), z, np.full(n, label)], 1).astype(np.float32)


ONE_BIN_CASES = [
    ("equal heights merge", {}, column(5, 0.0, 1.0), column(5, 0.0, 1.0, dy=0.6)),
    ("curr flat -> revert", {}, column(6, 0.0, 1.0), column(4, 0.0, 0.0, dy=0.6)),
    ("both flat -> NaN merge", {}, column(4, 0.25, 0.25), column(4, 0.5, 0.5, dy=0.6)),
    ("min pts 3<4", {"minimum_num_pts": 4}, column(6, 0.0, 1.0), column(3, 0.0, 0.0, dy=0.6)),
    ("min pts 4>=4", {"minimum_num_pts": 4}, column(6, 0.0, 1.0), column(4, 0.0, 0.0, dy=0.6)),
    ("curr higher", {}, column(5, 0.0, 0.1), column(5, 0.0, 2.0, dy=0.6)),
    ("gate exactly 0.5", {}, column(6, 0.0, 0.5), column(4, 0.0, 0.0, dy=0.6)),
    ("gate just above 0.5", {}, column(6, 0.0, 0.5000001), column(4, 0.0, 0.0, dy=0.6)),
    ("v2 th_bin_max_h", {"version": 2, "th_bin_max_h": 0.9}, column(6, 0.0, 1.0), column(4, 0.0, 0.0, dy=0.6)),
    ("v2 below th_bin_max_h", {"version": 2, "th_bin_max_h": 1.5}, column(6, 0.0, 1.0), column(4, 0.0, 0.0, dy=0.6)),
    ("v2 merge", {"version": 2}, column(5, 0.0, 1.0), column(5, 0.0, 1.0, dy=0.6)),
    ("v2 curr higher rejected", {"version": 2, "th_bin_max_h": 0.5}, column(5, 0.0, 0.1), column(5, 0.0, 2.0, dy=0.6)),
    ("v2 only curr", {"version": 2}, np.zeros((0, 4), np.float32), column(5, 0.0, 2.0, dy=0.6)),
    ("rgpf 2 pts, lpr 0", {"num_lowest_pts": 5, "minimum_num_pts": 1}, column(2, 0.0, 1.0), column(3, 0.0, 0.0, dy=0.6)),
    ("rgpf no seeds -> degenerate plane", {"gf_th_seeds_height": -10.0}, column(6, 0.0, 1.0), column(4, 0.0, 0.0, dy=0.6)),
    ("equal z ties in std::sort", {}, np.concatenate([column(20, 0.0, 0.0), column(5, 0.2, 1.5, x0=2.1, dy=0.9)]),
     column(4, 0.0, 0.0, dy=0.6)),
]


@pytest.mark.parametrize("name,over,m,s", ONE_BIN_CASES, ids=[c[0] for c in ONE_BIN_CASES])
def test_one_bin_known_answers_against_reference_source(name, over, m, s):
    p = one_bin_params(**over)
    o = orc.Oracle(p)
    o.set_map(m)
    r = ref.RefUpdater(p, m, ID7)
    ro = o.step(s, I4, I4, I4)
    r.step(s, ID7)
    compare_state(o, r, ro, tag=name)


def test_erasor_class_direct_calls_match_oracle_step_products():
    """ERASOR::set_inputs / compare_* / get_static_estimate / get_outliers called directly (erasor.h:109-147)"""
    sc = scenarios.small()
    p = sc["params"]
    o = orc.Oracle(p)
    o.set_map(sc["map"])
    r = ref.RefUpdater(p, sc["map"][:10], L2B)
    for k in range(2):
        o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
        # feed the reference's ERASOR the oracle's egocentric inputs; egocentric outputs must come back identical
        assert r.erasor_run(o.get_cloud(1), o.get_cloud(0), 3) == 0
        assert same(r.erasor_get(0), o.get_cloud(2)) and same(r.erasor_get(1), o.get_cloud(3))
        assert same(r.get_cloud(6), o.get_cloud(6))
        ego_rej = orc.transform(o.get_cloud(4), sc["T_o2b"][k])  # not bit-reversible in general; compare sizes + labels
        assert len(r.erasor_get(2)) == len(ego_rej)
        assert np.array_equal(bits(o.get_status()), bits(r.get_status()))
    assert r.lib.ref_erasor_get_max_range(r.h) == p.max_range


def test_negative_zero_hazard_reference_throws_oracle_counts():
    """y == -0.0f, x < 0: `y >= 0` holds, atan2 = -pi, sector = -30 -> vector::at throws (erasor.cpp:12-13,109-112,136)"""
    p = one_bin_params(num_sectors=60)
    pt = np.array([[-5.0, -0.0, 0.0, 40.0]], np.float32)
    r = ref.RefUpdater(p, pt, ID7)
    assert r.erasor_run(pt, np.zeros((0, 4), np.float32), 3) == -100
    assert b"out_of_range" in r.lib.ref_last_error()
    assert r.erasor_run(np.zeros((0, 4), np.float32), pt, 3) == -100  # query side too (erasor.cpp:112)
    assert orc.bin_of(p, -5.0, -0.0, 0.0) == 0  # defined clamp, counted in n_neg_sector
    # through callback_node the hazard is unreachable: transformPointCloud's `+ T03` turns -0.0 into +0.0
    o = orc.Oracle(p)
    o.set_map(pt)
    ro = o.step(np.zeros((0, 4), np.float32), I4, I4, I4)
    r.step(np.zeros((0, 4), np.float32), ID7)
    compare_state(o, r, ro, tag="-0.0 via step")
    assert ro.n_neg_sector == 0


def test_bin_edges_against_reference_binning():
    """xy2theta / xy2radius / gates (erasor.cpp:11-21,104-110) probed point by point through ERASOR::set_inputs"""
    p = one_bin_params(num_rings=15, num_sectors=60, max_range=60.0, min_h=-1.25, max_h=3.25)
    rng = np.random.default_rng(7)
    nf = np.float32
    pts = [(60.0, 0.0, 0.0), (np.nextafter(nf(60), nf(61)), 0.0, 0.0), (1.0, 0.0, -1.25), (1.0, 0.0, np.nextafter(nf(-1.25), nf(0))),
           (1.0, 0.0, 3.25), (1.0, 0.0, np.nextafter(nf(3.25), nf(0))), (0.0, 10.0, 0.0), (-10.0, 0.0, 0.0), (10.0, -1e-30, 0.0),
           (3.9999, 0.0, 0.0), (4.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, -7.0, 1.0), (-3.0, -1e-38, 0.5), (42.42641, 42.42641, 0.0)]
    # sector / ring boundaries approached from both sides in float32
    for s in range(0, 60, 7):
        th = s * (2 * 3.1415926535 / 60)
        for rad in (3.999999, 4.0, 27.3, 59.99999):
            x, y = nf(rad * np.cos(th)), nf(rad * np.sin(th))
            pts += [(x, y, 0.1), (np.nextafter(x, nf(100)), y, 0.1), (x, np.nextafter(y, nf(-100)), 0.1)]
    pts = np.array(pts, np.float32)
    extra = np.concatenate([rng.uniform(-65, 65, (4000, 2)), rng.uniform(-2, 4, (4000, 1))], 1).astype(np.float32)
    pts = np.concatenate([pts, extra])
    pts = pts[~((pts[:, 1] == 0) & np.signbit(pts[:, 1]) & (pts[:, 0] < 0))]
    cloud = np.concatenate([pts, np.full((len(pts), 1), 40, np.float32)], 1)
    r = ref.RefUpdater(p, cloud[:1], ID7)
    want = np.array([orc.bin_of(p, *c[:3]) for c in cloud])
    # one point at a time is slow through ctypes; instead bin the whole cloud and compare per-bin counts + min/max,
    # then spot-check the hand-placed edge points individually
    assert r.erasor_run(cloud, np.zeros((0, 4), np.float32), 3) == 0
    cnt, mn, mx = r.get_bins(0)
    exp = np.bincount(want[want >= 0], minlength=900)
    assert np.array_equal(cnt, exp)
    assert len(r.erasor_get(1)) == int((want < 0).sum())  # complement
    for c in cloud[:60]:
        assert r.erasor_run(c[None], np.zeros((0, 4), np.float32), 3) == 0
        cnt, _, _ = r.get_bins(0)
        b = orc.bin_of(p, *c[:3])
        assert (cnt.sum() == 0 and b == -1) or (cnt[b] == 1 and cnt.sum() == 1), (c, b)


def test_is_dynamic_obj_close_wrap_quirk():
    """theta wrap uses num_rings (erasor.cpp:578,580): with 15 rings x 60 sectors, sector -1 maps to 14, sector 60 to 45"""
    p = one_bin_params(num_rings=15, num_sectors=60, max_range=60.0, minimum_num_pts=3)
    ring, S = 2, 60
    def col_at(sector, n, z1, dy=0.0):
        th = (sector + 0.5) * (2 * 3.1415926535 / S)
        rad = 10.0 + 0.25 * np.arange(n)
        z = np.linspace(0, z1, n)
        return np.stack([rad * np.cos(th + dy), rad * np.sin(th + dy), z, np.full(n, 40.0)], 1).astype(np.float32)
    # CURR_IS_HIGHER in sector 14 (map flat, scan tall); MERGE in sectors 0 and 59 and 15
    m = np.concatenate([col_at(14, 5, 0.1), col_at(0, 5, 1.0), col_at(59, 5, 1.0), col_at(15, 5, 1.0), col_at(45, 5, 0.1)])
    s = np.concatenate([col_at(14, 5, 2.0, 0.01), col_at(0, 5, 1.0, 0.01), col_at(59, 5, 1.0, 0.01), col_at(15, 5, 1.0, 0.01),
                        col_at(45, 5, 2.0, 0.01)])
    o = orc.Oracle(p)
    o.set_map(m)
    r = ref.RefUpdater(p, m, ID7)
    ro = o.step(s, I4, I4, I4)
    r.step(s, ID7)
    compare_state(o, r, ro, tag="wrap quirk")
    st = r.get_status()
    assert st[ring * S + 14] == 1.0 and st[ring * S + 15] == 0.8          # true neighbour is blocked
    assert st[ring * S + 0] == 0.8                                       # sector 0 looks at -1 -> 14 (num_rings!) -> blocked
    assert st[ring * S + 59] == 0.8                                      # sector 59 looks at 60 -> 45 -> blocked
    assert r.is_dynamic_obj_close(ring, 0) and r.is_dynamic_obj_close(ring, 59) and not r.is_dynamic_obj_close(ring, 30)


# ---------------------------------------------------------------------------------------------
# erasor_utils free functions (utils.cpp)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("leaf", [0.05, 0.2, 0.5, 1.0])
def test_voxelize_preserving_labels_free_function(leaf):
    sc = scenarios.small()
    for cloud in (sc["scans"][0], sc["scans"][3][:5000], sc["map"][:30000]):
        assert same(orc.voxelize_preserving_labels(cloud, leaf), ref.voxelize_preserving_labels(cloud, leaf))


def test_voxelize_edge_cases():
    rng = np.random.default_rng(3)
    empty = np.zeros((0, 4), np.float32)
    assert len(ref.voxelize_preserving_labels(empty, 0.2)) == 0 and len(orc.voxelize_preserving_labels(empty, 0.2)) == 0
    one = np.array([[1, 2, 3, 40]], np.float32)
    assert same(orc.voxelize_preserving_labels(one, 0.2), ref.voxelize_preserving_labels(one, 0.2))
    # duplicates and exact 1-NN ties (lattice points: many equidistant neighbours with different labels)
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(3), indexing="ij"), -1).reshape(-1, 3) * 0.25
    lab = rng.integers(1, 300, len(g))
    lattice = np.concatenate([g, lab[:, None]], 1).astype(np.float32)
    lattice = np.concatenate([lattice, lattice[::3]])  # exact duplicates with other labels behind them
    lattice[-len(lattice[::3]):, 3] += 1
    for leaf in (0.5, 1.0, 0.25):
        assert same(orc.voxelize_preserving_labels(lattice, leaf), ref.voxelize_preserving_labels(lattice, leaf))
    # VoxelGrid index overflow: PCL warns and returns the input unchanged (utils.cpp:88-91)
    wide = np.concatenate([rng.uniform(-4000, 4000, (500, 3)), rng.integers(1, 99, (500, 1))], 1).astype(np.float32)
    a, b = orc.voxelize_preserving_labels(wide, 0.001), ref.voxelize_preserving_labels(wide, 0.001)
    assert len(a) == 500 and same(a, b)
    dup = np.concatenate([wide, wide[:50]])
    dup[-50:, 3] = 7
    assert same(orc.voxelize_preserving_labels(dup, 0.001), ref.voxelize_preserving_labels(dup, 0.001))


def test_geopose2eigen_and_labels():
    rng = np.random.default_rng(5)
    for _ in range(200):
        q = rng.normal(size=4)
        if rng.random() < 0.3:
            q /= np.linalg.norm(q)
        pose = np.concatenate([rng.uniform(-500, 500, 3), q])
        assert same(orc.geopose2eigen(pose), ref.geopose2eigen(pose))
    lab = np.array([0, 1, 40, 251, 252, 259, 260, 252 + (3 << 16), 259 + (200 << 16), 65535, 65536 + 252, 16777216.0, 3.9, 252.9],
                   np.float32)
    cloud = np.concatenate([rng.normal(size=(len(lab), 3)), lab[:, None]], 1).astype(np.float32)
    ns, nd = ref.count_stat_dyn(cloud)
    d, s = ref.parse_dynamic_obj(cloud)
    mask = synth.is_dynamic(lab)
    assert (ns, nd) == (int((~mask).sum()), int(mask.sum()))
    assert same(d, cloud[mask]) and same(s, cloud[~mask])


# ---------------------------------------------------------------------------------------------
# class mapgen (src/mapgen/mapgen.hpp:198-305)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("large", [False, True])
def test_mapgen_matches_reference_source(large):
    sc = scenarios.small()
    w = synth.World(seed=20210311, length=150.0)
    lid = synth.Lidar.hdl64(400)
    og = orc.Mapgen(0.2, is_large_scale=large)
    rg = ref.RefMapgen(0.2, is_large_scale=large)
    for f in range(0, 12, 2):
        pose = w.pose(f)
        scan = w.cast(pose, lid, f)
        n_o = og.accum(scan, orc.geopose2eigen(pose))
        n_r = rg.accum(scan, pose)
        assert n_o == n_r
        assert same(og.cloud_curr, rg.get(0)) and same(og.cloud_map, rg.get(1))
    dense, vox = rg.save()
    assert same(og.naive_map(), dense) and same(og.save(), vox)
    del sc
