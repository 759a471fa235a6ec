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
    assert len(FIXTURES) >= 3
