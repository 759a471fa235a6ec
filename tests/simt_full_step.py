"""Helper of tests/test_full_step_on_cpu.py (run as a subprocess): drives the library built by that test -- the product's
erasor_hip.hip + kernels compiled unmodified against tests/cpp/simt_emu -- through the ctypes wrapper and compares every
step with the oracle.  TEST INFRASTRUCTURE: the product never loads this library (erasor_amd.lib() loads liberasor_hip.so only)."""
import os
import sys
import time

os.environ.setdefault("ERASOR_HIP_OVERLAP", "1")  # (like tests/conftest.py: the path with the most moving parts)

import numpy as np

sys.path.insert(0, sys.argv[2])
sys.path.insert(0, sys.argv[2] + "/tests")
import erasor_amd  # noqa: E402

erasor_amd.LIB_PATH = sys.argv[1]  # the emulated build, for this process only
erasor_amd._lib = None
import scenarios  # noqa: E402
from oracle import orc  # noqa: E402

n_steps = int(sys.argv[3])
sc = scenarios.small(n_frames=6, az=120, length=60.0)
g = erasor_amd.Erasor(scenarios.to_product_params(sc["params"]))
o = orc.Oracle(sc["params"])
g.set_map(sc["map"])
o.set_map(sc["map"])
scans = [np.ascontiguousarray(s, np.float32) for s in sc["scans"]]
ok = True
g.prefetch(scans[0], sc["T_l2b"], sc["T_b2o"][0])  # nodes announced with their pose: look-ahead and the split launched ahead
for k in range(n_steps):
    if k + 1 < n_steps:
        g.prefetch(scans[k + 1], sc["T_l2b"], sc["T_b2o"][k + 1])
    t0 = time.time()
    rg = g.step(scans[k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
    dt = time.time() - t0
    ro = o.step(scans[k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
    dg, do = rg.as_dict(), ro.as_dict()
    bad = {f: (dg[f], do[f]) for f in do if dg[f] != do[f] and f not in ("n_ambiguous", "n_sort_fallback")}
    same = {"map": np.array_equal(g.get_map().view(np.uint32), o.get_cloud(7).view(np.uint32))}
    for which, name in ((1, "map_voi"), (2, "static_estimate"), (4, "map_rejected"), (5, "curr_rejected")):
        a, b = g.get_cloud(which), o.get_cloud(which)
        same[name] = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    same["status"] = np.array_equal(g.get_status(), o.get_status())
    bg, ng, dg_ = g.get_planes()
    bo, no, do_ = o.get_planes()
    same["planes"] = np.array_equal(bg, bo) and np.array_equal(ng.view(np.uint32), no.view(np.uint32)) and np.array_equal(dg_, do_)
    good = not bad and all(same.values())
    ok = ok and good
    print("step %d on the CPU stand-in in %.0f s: %d-pt VoI, %d reverted bins, %d rejected; differing result fields %s; bit-exact %s -> %s"
          % (k, dt, dg["n_voi"], dg["n_reverted_bins"], dg["n_map_rejected"], bad, same, "ok" if good else "MISMATCH"), flush=True)
# the call pattern of bench.py: scans resident in "device" memory (read in place), two nodes announced ahead with their poses
g2 = erasor_amd.Erasor(scenarios.to_product_params(sc["params"]))
o2 = orc.Oracle(sc["params"])
g2.set_map(sc["map"])
o2.set_map(sc["map"])
ptr = [s.ctypes.data for s in scans]
n3 = n_steps + 3  # (long enough for the chains announced after the first finished step: they take the one-launch query bucketing)
for j in range(2):
    g2.prefetch_device(ptr[j], len(scans[j]), sc["T_l2b"], sc["T_b2o"][j], sc["T_o2b"][j])
for k in range(n3):
    if k + 2 < n3:
        g2.prefetch_device(ptr[k + 2], len(scans[k + 2]), sc["T_l2b"], sc["T_b2o"][k + 2], sc["T_o2b"][k + 2])
    rg = g2.step_device(ptr[k], len(scans[k]), sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
    ro = o2.step(scans[k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
    dg, do = rg.as_dict(), ro.as_dict()
    bad = {f: (dg[f], do[f]) for f in do if dg[f] != do[f] and f not in ("n_ambiguous", "n_sort_fallback")}
    ok = ok and not bad
same_map = np.array_equal(g2.get_map().view(np.uint32), o2.get_cloud(7).view(np.uint32))
l2, u2 = g2.ahead_split_counts()
print("overlapped steps: launched %d, taken %d" % g2.overlap_counts())
print("bench.py's call pattern (device scans, two nodes ahead), %d steps: result blocks equal %s, final map bit-exact %s; splits ahead %d launched / %d used"
      % (n3, ok, same_map, l2, u2))
ok = ok and same_map and l2 == n3 - 1
# the same nodes through erasor_hip_run_nodes (the node loop in native code), split over two calls like bench.py (warm-up, timed part)
import ctypes as C
g3 = erasor_amd.Erasor(scenarios.to_product_params(sc["params"]))
g3.set_map(sc["map"])
P_, N_, Tb_, To_ = erasor_amd.Erasor.node_arrays(ptr[:n3], [len(s) for s in scans[:n3]], sc["T_b2o"][:n3], sc["T_o2b"][:n3])
ann = C.c_size_t(0)
rs = g3.run_nodes(P_, N_, sc["T_l2b"], Tb_, To_, 0, 1, 2, ann) + g3.run_nodes(P_, N_, sc["T_l2b"], Tb_, To_, 1, n3 - 1, 2, ann)
same3 = np.array_equal(g3.get_map().view(np.uint32), o2.get_cloud(7).view(np.uint32)) and len(rs) == n3 and ann.value == n3
print("erasor_hip_run_nodes (two calls, %d nodes, two ahead): final map bit-exact %s" % (n3, same3))
ok = ok and same3
# round 6: the same nodes announced SIX ahead: chains beyond the third in line are held back until two (then three) of them share one
# set of launches (erasor_hip_chain_batch); a held chain whose partner never comes goes off alone when its step is near
for nb in (2, 3):
    g4 = erasor_amd.Erasor(scenarios.to_product_params(sc["params"]))
    g4.set_map(sc["map"])
    g4.chain_batch(nb, 2)
    ann4 = C.c_size_t(0)
    rs4 = g4.run_nodes(P_, N_, sc["T_l2b"], Tb_, To_, 0, n3, 6, ann4)
    sets, chains = g4.chain_batch_counts()
    same4 = np.array_equal(g4.get_map().view(np.uint32), o2.get_cloud(7).view(np.uint32)) and all(
        a.as_dict() == b.as_dict() for a, b in zip(rs4, rs))
    print("erasor_hip_run_nodes, six ahead, chains in sets of %d: %d sets / %d chains shared their launches; results and final map bit-exact %s"
          % (nb, sets, chains, same4))
    ok = ok and same4 and sets >= 1 and chains >= 2
launched, used = g.ahead_split_counts()
print("VoI splits launched ahead: %d, used: %d" % (launched, used))
ok = ok and launched == n_steps - 1  # (whether the next step could use it depends on scratch growth in the first steps)
print("FULL-STEP-OK" if ok else "FULL-STEP-FAILED")
