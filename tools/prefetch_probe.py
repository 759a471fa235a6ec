"""timing of every API call in a prefetch / step / read-back loop (debug aid)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import erasor_amd, scenarios
sc = scenarios.small()
g = erasor_amd.Erasor(scenarios.to_product_params(sc["params"]))
def T(label, f, *a):
    t = time.time(); r = f(*a); dt = time.time() - t
    print("%-28s %8.1f ms" % (label, dt * 1e3), flush=True)
    return r
T("set_map", g.set_map, sc["map"])
scans = [np.ascontiguousarray(s, np.float32) for s in sc["scans"]]
held = T("prefetch 0", g.prefetch, scans[0], sc["T_l2b"])
for k in range(6):
    nxt = T("prefetch %d" % (k + 1), g.prefetch, scans[k + 1], sc["T_l2b"])
    T("step %d" % k, g.step, held, sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
    for w in (0, 1, 2, 4, 5):
        T("  get_cloud %d" % w, g.get_cloud, w)
    T("  get_map", g.get_map)
    T("  get_planes", g.get_planes)
    T("  get_bins", g.get_bins, 0)
    held = nxt
print("--- dropped prefetch", flush=True)
T("prefetch scans[0]", g.prefetch, scans[0], sc["T_l2b"])
k = 6
T("step %d (other scan)" % k, g.step, scans[k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
print("--- standalone call while pending", flush=True)
k = 7
held = T("prefetch %d" % k, g.prefetch, scans[k], sc["T_l2b"])
T("voxelize standalone", g.voxelize_preserving_labels, sc["scans"][0], 0.3)
T("step %d" % k, g.step, held, sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
T("get_rejected_indices", g.get_rejected_indices)
T("get_status", g.get_status)
T("count_static_dynamic", g.count_static_dynamic)
