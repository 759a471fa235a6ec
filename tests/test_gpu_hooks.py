"""GPU tests that need the library's TEST HOOKS (device libm probe, the exact / radix sorts on their own, a forced outskirts rebuild).

Round 4: the product library (erasor_amd/liberasor_hip.so) no longer exports them.  The same sources are compiled a second time with
-DERASOR_HIP_TEST_HOOKS into tests/_build/liberasor_hip_hooks.so (erasor_amd/csrc/Makefile); for this module -- and only here -- the
ctypes wrapper loads that copy, and the hook calls are made by tests/hooks.py, not by the package.
"""
import ctypes as C
import os

import numpy as np
import pytest

import hooks
import scenarios
from test_gpu_parity import bits, compare_step, make_pair, same  # noqa: F401  (helpers only)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_mod():
    import erasor_amd
    erasor_amd.build()
    with hooks.hooks_library():
        yield erasor_amd


def test_device_libm_as_used_by_the_binning(gpu_mod):
    g = gpu_mod.Erasor(gpu_mod.params_default())
    rng = np.random.default_rng(1)
    x = rng.uniform(-80, 80, 300000).astype(np.float32).astype(np.float64)
    y = rng.uniform(-80, 80, 300000).astype(np.float32).astype(np.float64)
    s, d, a = hooks.probe_math(g, x, y)
    same(s, np.sqrt(x * x + y * y), "sqrt f64 (correctly rounded)")
    same(d, x / y, "div f64 (correctly rounded)")
    at = np.arctan2(y, x)
    ulp = np.abs(a - at) / np.spacing(np.abs(at))
    assert ulp.max() <= 4.0   # OCML vs glibc differ by <= 2 ulp: the reason for the n_ambiguous guard band (1e-11 >> 1e-15)


def test_bin_key_float32_decision_never_disagrees_with_the_float64_key(gpu_mod):
    """Round 4: bin_key decides a point's ring / sector in float32 (hardware rcp / sqrt, an 8-term atan polynomial) unless it lies within
    5e-4 m / 1e-5 rad of a boundary of the cell it lands in, where the reference's float64 arithmetic (bin_key_exact: erasor.cpp:124-139
    restated) takes over.  Sound iff no accepted point ever differs: random points, points on and next to every boundary (offsets
    1e-9 ... 1e-3 of a cell, built in float64), axes, signed zeros, tiny / huge coordinates; and against the oracle's own keys."""
    from oracle import orc
    for rings, sectors, rng_m in ((20, 108, 80.0), (15, 60, 60.0), (7, 31, 9.7)):
        p = gpu_mod.params_default()
        p.num_rings, p.num_sectors, p.max_range = rings, sectors, rng_m
        g = gpu_mod.Erasor(p)
        rng = np.random.default_rng(rings)
        ring, sector = rng_m / rings, 2 * 3.1415926535 / sectors
        pts = [np.column_stack([rng.uniform(-1.05 * rng_m, 1.05 * rng_m, (2000000, 2)), rng.uniform(-4, 4, 2000000), np.zeros(2000000)])]
        offs = np.array([0.0, 1e-9, 1e-8, 1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3])
        offs = np.concatenate([offs, -offs])
        th = (np.arange(sectors + 1)[:, None, None] + offs[None, :, None]) * sector + np.zeros((1, 1, 200))
        r = rng.uniform(0, 1.01 * rng_m, th.shape)
        pts.append(np.column_stack([(r * np.cos(th)).ravel(), (r * np.sin(th)).ravel(), np.full(th.size, 0.5), np.zeros(th.size)]))
        r = (np.arange(rings + 1)[:, None, None] + offs[None, :, None]) * ring + np.zeros((1, 1, 2000))
        th = rng.uniform(0, 2 * np.pi, r.shape)
        pts.append(np.column_stack([(r * np.cos(th)).ravel(), (r * np.sin(th)).ravel(), np.full(th.size, 0.5), np.zeros(th.size)]))
        sp = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-30, -1e-30, 1e-3, -1e-3, 1, -1, 39.99999, 40, 40.00001, -40, 79.99999, 80, 80.00001, -80,
                       1e10, -1e10, 1e20, -1e20, 3e38, -3e38], np.float32)
        zs = np.array([-1.3, -1.29999, 0, 3.19999, 3.2, 100], np.float32)
        gx, gy, gz = np.meshgrid(sp, sp, zs, indexing="ij")
        pts.append(np.column_stack([gx.ravel(), gy.ravel(), gz.ravel(), np.zeros(gx.size, np.float32)]))
        pts = np.ascontiguousarray(np.concatenate(pts), np.float32)
        kf, ke, ctr = hooks.probe_bin_keys(g, pts)
        bad = np.flatnonzero(kf != ke)
        assert len(bad) == 0, (rings, sectors, len(bad), pts[bad[:5]], kf[bad[:5]], ke[bad[:5]])
        assert ctr[0] == ctr[2] and ctr[1] == ctr[3], ctr
        po = orc.Params()   # (same layout, distinct ctypes class)
        C.memmove(C.byref(po), C.byref(p), C.sizeof(po))
        n_spec = gx.size
        for i in rng.choice(len(pts) - n_spec, 3000, replace=False):   # the oracle itself, one call per point
            ko = orc.bin_of(po, *pts[i, :3])
            ko = rings * sectors if ko < 0 else (ko % sectors) * rings + ko // sectors   # (the oracle numbers ring-major, the device sector-major)
            assert ko == int(kf[i]), (i, pts[i], ko, kf[i])


@pytest.mark.parametrize("n,key_range", [(0, 5), (1, 5), (16, 3), (17, 3), (100, 7), (5000, 50), (5000, 1 << 30), (70000, 300),
                                         (200000, 40000), (300000, 5)])
def test_exact_std_sort_emulation(gpu_mod, n, key_range):
    from oracle import orc
    g = gpu_mod.Erasor(gpu_mod.params_default())
    rng = np.random.default_rng(n + key_range)
    k = rng.integers(0, key_range, n).astype(np.uint32)
    v = np.arange(n, dtype=np.uint32)
    gk, gv, _ = hooks.exact_sort_u32(g, k, v)
    ok, ov = orc.std_sort_u32(k, v)
    same(gk, ok, "keys")
    same(gv, ov, "tie order (libstdc++ introsort permutation)")


def test_exact_sort_heapsort_fallback_on_median_of_3_killer(gpu_mod):
    from oracle import orc
    g = gpu_mod.Erasor(gpu_mod.params_default())
    for n in (1000, 4096, 30000):
        a = np.zeros(n, np.uint32)
        k = n // 2
        for i in range(1, k + 1):
            if i & 1:
                a[i - 1] = i
                a[i] = k + i
            a[k + i - 1] = 2 * i
        gk, gv, nf = hooks.exact_sort_u32(g, a, np.arange(n, dtype=np.uint32))
        ok, ov = orc.std_sort_u32(a, np.arange(n, dtype=np.uint32))
        same(gk, ok)
        same(gv, ov)
        assert nf > 0, "depth limit was never hit: the fallback is not exercised"


def test_exact_sort_long_segments_reach_the_final_kernel(gpu_mod):
    """With the level budget cut to one wide level (test hook), segments of tens of thousands of keys reach the final
    kernel's global-memory path: its bounded per-piece queue must neither overflow nor change the permutation."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, erasor_amd\n"
        "sys.path.insert(0, 'tests')\n"
        "import hooks\n"
        "from oracle import orc\n"
        "ctx = hooks.hooks_library(); ctx.__enter__()\n"
        "g = erasor_amd.Erasor(erasor_amd.params_default())\n"
        "for n, kr in ((200000, 3000), (150001, 1 << 32), (40000, 17)):\n"
        "    k = np.random.default_rng(n).integers(0, kr, n).astype(np.uint32)\n"
        "    v = np.arange(n, dtype=np.uint32)\n"
        "    gk, gv, _ = hooks.exact_sort_u32(g, k, v)\n"
        "    ok, ov = orc.std_sort_u32(k, v)\n"
        "    assert np.array_equal(gk, ok) and np.array_equal(gv, ov), (n, kr)\n"
        "print('LONG-SEGMENTS-OK')\n"
    )
    env = dict(os.environ, ERASOR_HIP_SORT_LEVEL_CAP="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert "LONG-SEGMENTS-OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("n,B", [(0, 900), (5, 900), (12453, 900), (100000, 2160), (300001, 2160)])
def test_stable_radix_bucketing(gpu_mod, n, B):
    g = gpu_mod.Erasor(gpu_mod.params_default())
    k = np.random.default_rng(n).integers(0, B + 1, n).astype(np.uint32)
    ko, po = hooks.radix_sort_u32(g, k, max(1, int(np.ceil(np.log2(B + 1)))))
    order = np.argsort(k, kind="stable").astype(np.uint32)
    same(ko, k[order])
    same(po, order)



def test_outskirts_rebuild_is_invisible(gpu_mod):
    """tombstones + front growth are an HBM layout detail: forcing the compaction must not change anything"""
    sc = scenarios.small()
    g, o = make_pair(gpu_mod, sc["params"])
    g.set_map(sc["map"])
    o.set_map(sc["map"])
    for f in range(6):
        ro = o.step(sc["scans"][f], sc["T_l2b"], sc["T_b2o"][f], sc["T_o2b"][f])
        rg = g.step(sc["scans"][f], sc["T_l2b"], sc["T_b2o"][f], sc["T_o2b"][f])
        if f % 2 == 1:
            hooks.debug_rebuild_outskirts(g)
            same(g.get_map(), o.get_map(), "map after forced rebuild")
        compare_step(g, o, rg, ro, full=False)


