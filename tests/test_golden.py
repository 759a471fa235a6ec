"""Frozen fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the CPU oracle must keep
reproducing them (no GPU), and the HIP path must reproduce them through the C ABI (-m gpu)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "seq*.npz")))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def fill_params(p, vals, fields):
    for f, v in zip(fields, vals):
        cur = getattr(p, f)
        setattr(p, f, int(v) if isinstance(cur, int) else float(v))
    return p


def replay(engine, z, check_fine):
    engine.set_map(z["map0"])
    keys = [k for k in z["res_keys"]]
    for k in range(int(z["n_steps"])):
        r = engine.step(z["scan%d" % k], z["T_l2b"], z["T_b2o%d" % k], z["T_o2b%d" % k])
        got = r.as_dict()
        want = dict(zip(keys, z["res%d" % k].tolist()))
        for name in keys:
            if name in ("n_ambiguous", "n_sort_fallback"):
                continue
            assert got[name] == want[name], (k, name, got[name], want[name])
        assert np.array_equal(engine.get_rejected_indices(), z["rejidx%d" % k])
        assert np.array_equal(engine.get_status(), z["status%d" % k])
        b, n, d = engine.get_planes()
        assert np.array_equal(b, z["plane_bins%d" % k])
        # north_star tolerance on plane coefficients is 1e-5; the implementation is in fact bit-exact
        assert np.allclose(n, z["plane_n%d" % k], atol=1e-5, rtol=0) and np.allclose(d, z["plane_d%d" % k], atol=1e-5, rtol=0)
        assert np.array_equal(bits(n), bits(z["plane_n%d" % k])) and np.array_equal(bits(d), bits(z["plane_d%d" % k]))
        assert np.array_equal(bits(engine.get_cloud(0)), bits(z["query%d" % k]))
    m = engine.get_map()
    assert m.shape == z["map_final"].shape and np.array_equal(bits(m), bits(z["map_final"]))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_reproduces_golden(path):
    from oracle import orc
    z = np.load(path)
    fields = [f for f, _ in orc.Params._fields_ if f != "reserved_"]
    p = fill_params(orc.params_default(), z["params"], fields)
    replay(orc.Oracle(p), z, True)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_hip_reproduces_golden(path):
    import erasor_amd
    z = np.load(path)
    fields = [f for f, _ in erasor_amd.Params._fields_ if f != "reserved_"]
    p = fill_params(erasor_amd.params_default(), z["params"], fields)
    replay(erasor_amd.Erasor(p), z, True)


def test_fixtures_exist():
    assert len(FIXTURES) >= 3 and len(REF_FIXTURES) >= 4


# ---------------------------------------------------------------------------------------------
# vectors produced by the reference itself (tests/golden/make_ref_golden.py -> oracle/_ref = the reference's
# unmodified sources): every product of OfflineMapUpdater::callback_node, per step
# ---------------------------------------------------------------------------------------------
REF_FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_*.npz")))


def replay_ref(engine, z, voxelize):
    names = [str(s) for s in z["param_names"]]
    R, S = int(z["params"][names.index("num_rings")]), int(z["params"][names.index("num_sectors")])
    min_pts = int(z["params"][names.index("minimum_num_pts")])
    engine.set_map(z["map0"])
    for k in range(int(z["n_steps"])):
        r = engine.step(z["scan%d" % k], z["T_l2b"], z["T_b2o%d" % k], z["T_o2b%d" % k])
        for which, key in ((0, "query"), (2, "static_estimate"), (4, "map_rejected"), (5, "curr_rejected"), (6, "ground")):
            got, want = engine.get_cloud(which), z["%s%d" % (key, k)]
            assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (k, key)
        st = engine.get_status()
        order = np.arange(R * S).reshape(R, S).T.reshape(-1)  # the reference pushes polygons theta-major
        cnt = {}
        for which in (0, 1):
            c, mn, mx = engine.get_bins(which)
            cnt[which] = c
            assert np.array_equal(c, z["bins%d_cnt%d" % (which, k)])
            occ = c > 0
            assert np.array_equal(mn[occ], z["bins%d_min%d" % (which, k)][occ]) and np.array_equal(mx[occ], z["bins%d_max%d" % (which, k)][occ])
        if int(z["version"]) == 3:
            assert np.array_equal(st, z["status%d" % k])
            assert np.array_equal(z["likelihood%d" % k], st[order].astype(np.float32))
        else:  # v2 only publishes the SRT outcome as polygon likelihoods (erasor.cpp:345-425)
            pushed = (cnt[1] < min_pts) | ((cnt[1] > 0) & (cnt[0] > 0))
            assert np.array_equal(z["likelihood%d" % k], st[order][pushed[order]].astype(np.float32))
        _, n, d = engine.get_planes()
        want_n = z["plane_n%d" % k]
        # north_star: plane coefficients within 1e-5 — they are in fact bit-identical
        assert n.reshape(-1, 3).shape == want_n.shape and np.allclose(n.reshape(-1, 3), want_n, atol=1e-5, rtol=0)
        assert np.array_equal(bits(n.reshape(-1, 3)), bits(want_n))
        if d.size:
            assert abs(d.reshape(-1)[-1] - z["plane_last_d%d" % k][0]) <= 1e-5 and d.reshape(-1)[-1] == z["plane_last_d%d" % k][0]
        assert (r.n_static, r.n_dynamic) == tuple(z["labels%d" % k].tolist())
        assert r.n_map_rejected == len(z["map_rejected%d" % k]) and r.n_reverted_bins * 3 == len(want_n)
    m = engine.get_map()
    assert m.shape == z["map_final"].shape and np.array_equal(bits(m), bits(z["map_final"]))
    saved = voxelize(m, 0.2)
    assert saved.shape == z["saved_0_2"].shape and np.array_equal(bits(saved), bits(z["saved_0_2"]))


def params_from(z, P):
    p = P
    for f, v in zip([str(s) for s in z["param_names"]], z["params"]):
        cur = getattr(p, f)
        setattr(p, f, int(v) if isinstance(cur, int) else float(v))
    return p


@pytest.mark.parametrize("path", REF_FIXTURES, ids=[os.path.basename(p) for p in REF_FIXTURES])
def test_oracle_reproduces_reference_vectors(path):
    from oracle import orc
    z = np.load(path)
    replay_ref(orc.Oracle(params_from(z, orc.params_default())), z, orc.voxelize_preserving_labels)


@pytest.mark.gpu
@pytest.mark.parametrize("path", REF_FIXTURES, ids=[os.path.basename(p) for p in REF_FIXTURES])
def test_hip_reproduces_reference_vectors(path):
    """data-only: nothing but the committed vectors and the C ABI"""
    import erasor_amd
    z = np.load(path)
    e = erasor_amd.Erasor(params_from(z, erasor_amd.params_default()))
    replay_ref(e, z, e.voxelize_preserving_labels)
