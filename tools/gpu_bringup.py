"""Stage-by-stage GPU bring-up: HIP path vs CPU oracle, keeps going after a mismatch and says where.
Run on the GPU box:  python tools/gpu_bringup.py [n_steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import erasor_amd  # noqa: E402
from oracle import orc  # noqa: E402
import scenarios  # noqa: E402

bad = 0


def cmp(name, a, b, exact=True, tol=0.0):
    global bad
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        print("  [FAIL] %-22s shape %s vs %s" % (name, a.shape, b.shape))
        bad += 1
        return False
    if a.size == 0:
        print("  [ ok ] %-22s (empty)" % name)
        return True
    if exact:
        if a.dtype.kind == "f":
            eq = (a.view(np.uint32 if a.dtype == np.float32 else np.uint64) == b.view(np.uint32 if b.dtype == np.float32 else np.uint64)) | (a == b)
        else:
            eq = a == b
        nbad = int((~eq).sum())
        if nbad:
            idx = np.argwhere(~eq)[:5]
            print("  [FAIL] %-22s %d / %d elements differ; first at %s: got %s want %s" % (name, nbad, a.size, idx.tolist(), a[tuple(idx[0])], b[tuple(idx[0])]))
            bad += 1
            return False
        print("  [ ok ] %-22s n=%s bit-exact" % (name, a.shape))
        return True
    err = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))
    ok = err <= tol
    print("  [%s] %-22s max|diff|=%.3g (tol %.1g)" % (" ok " if ok else "FAIL", name, err, tol))
    bad += 0 if ok else 1
    return ok


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    print(erasor_amd.lib().erasor_hip_version().decode())
    sc = scenarios.small()
    P = scenarios.to_product_params(sc["params"])
    g = erasor_amd.Erasor(P)
    print("== device libm probe")
    rng = np.random.default_rng(1)
    x = rng.uniform(-80, 80, 200000).astype(np.float32).astype(np.float64)
    y = rng.uniform(-80, 80, 200000).astype(np.float32).astype(np.float64)
    s, d, a = g.probe_math(x, y)
    cmp("sqrt(x2+y2) f64", s, np.sqrt(x * x + y * y))
    cmp("x/y f64", d, x / y)
    at = np.arctan2(y, x)
    ulp = np.abs(a - at) / np.spacing(np.abs(at))
    print("  atan2 f64: max ulp diff vs libm = %.2f, mismatching = %d / %d" % (ulp.max(), int((a != at).sum()), len(a)))
    print("== exact std::sort emulation")
    for n, rngk in ((0, 5), (1, 5), (10, 3), (100, 7), (5000, 50), (5000, 1 << 30), (20000, 300), (200000, 40000), (300000, 7)):
        k = rng.integers(0, rngk, n).astype(np.uint32)
        v = np.arange(n, dtype=np.uint32)
        t = time.time()
        gk, gv, nf = g.exact_sort_u32(k, v)
        dt = time.time() - t
        ok, ov = orc.std_sort_u32(k, v)
        r = cmp("esort n=%d r=%d" % (n, rngk), np.stack([gk, gv]), np.stack([ok, ov]))
        print("        (%.1f ms wall, %d heapsort fallbacks)" % (dt * 1e3, nf))
    print("== stable radix bucketing")
    for n, B in ((0, 900), (5, 900), (3000, 900), (12453, 900), (100000, 2160), (300001, 2160)):
        k = rng.integers(0, B + 1, n).astype(np.uint32)
        bits = max(1, int(np.ceil(np.log2(B + 1))))
        ko, po = g.radix_sort_u32(k, bits)
        order = np.argsort(k, kind="stable").astype(np.uint32)
        cmp("radix n=%d B=%d" % (n, B), np.stack([ko, po]), np.stack([k[order], order]))
    print("== voxelize_preserving_labels (standalone)")
    for leaf in (0.2, 0.5):
        vg = g.voxelize_preserving_labels(sc["scans"][0], leaf)
        vo = orc.voxelize_preserving_labels(sc["scans"][0], leaf)
        cmp("voxelize leaf=%.1f" % leaf, vg, vo)
    print("== steps")
    o = orc.Oracle(sc["params"])
    o.set_map(sc["map"])
    g.set_map(sc["map"])
    cmp("map after set_map", g.get_map(), o.get_map())
    for f in range(nsteps):
        print("-- step %d" % f)
        ro = o.step(sc["scans"][f], sc["T_l2b"], sc["T_b2o"][f], sc["T_o2b"][f])
        t = time.time()
        rg = g.step(sc["scans"][f], sc["T_l2b"], sc["T_b2o"][f], sc["T_o2b"][f])
        dt = time.time() - t
        do, dg = ro.as_dict(), rg.as_dict()
        for k in do:
            if k in ("n_ambiguous", "n_sort_fallback"):
                continue
            if do[k] != dg[k]:
                print("  [FAIL] result.%s: got %d want %d" % (k, dg[k], do[k]))
                globals()["bad"] += 1
        print("  step wall %.2f ms; result %s" % (dt * 1e3, dg))
        cmp("query_voi", g.get_cloud(0), o.get_cloud(0))
        cmp("map_voi", g.get_cloud(1), o.get_cloud(1))
        for w, nm in ((0, "map"), (1, "curr")):
            cg, mng, mxg = g.get_bins(w)
            co, mno, mxo = o.get_bins(w)
            cmp("bins %s count" % nm, cg, co)
            cmp("bins %s min_h" % nm, mng, mno)
            cmp("bins %s max_h" % nm, mxg, mxo)
        cmp("status", g.get_status(), o.get_status())
        bg, ng_, dg_ = g.get_planes()
        bo, no_, do_ = o.get_planes()
        cmp("plane bins", bg, bo)
        cmp("plane normals", ng_, no_)
        cmp("plane d", dg_, do_)
        cmp("map_rejected", g.get_cloud(4), o.get_cloud(4))
        cmp("rejected indices", g.get_rejected_indices(), o.get_rejected_indices())
        cmp("ground_viz", g.get_cloud(6), o.get_cloud(6))
        cmp("static_estimate", g.get_cloud(2), o.get_cloud(2))
        cmp("complement", g.get_cloud(3), o.get_cloud(3))
        cmp("map", g.get_map(), o.get_map())
    print("TOTAL FAILURES: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
