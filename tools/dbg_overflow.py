import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import scenarios, erasor_amd
from erasor_amd import synth
from oracle import orc
which = sys.argv[1]
sc = scenarios.small()
def eq(a, b): return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
t = time.time()
if which == 'a':
    g0 = erasor_amd.Erasor(erasor_amd.params_default())
    rng = np.random.default_rng(4)
    wide = np.concatenate([rng.uniform(-4000, 4000, (5000, 3)), rng.integers(1, 99, (5000, 1))], 1).astype(np.float32)
    wide = np.concatenate([wide, wide[:300]]); wide[-300:, 3] = 7
    print('a', eq(g0.voxelize_preserving_labels(wide, 0.001), orc.voxelize_preserving_labels(wide, 0.001)))
elif which == 'b':
    g0 = erasor_amd.Erasor(erasor_amd.params_default())
    print('b', eq(g0.voxelize_preserving_labels(sc["map"][:50000], 0.01), orc.voxelize_preserving_labels(sc["map"][:50000], 0.01)))
elif which in ('c', 'd'):
    p = orc.params_default()
    if which == 'c': synth.apply_params(p, "05", query_voxel_size=0.02)
    else: synth.apply_params(p, "05", map_voxel_size=1e-4)
    g = erasor_amd.Erasor(scenarios.to_product_params(p)); o = orc.Oracle(p)
    g.set_map(sc["map"]); o.set_map(sc["map"])
    for k in range(2):
        rg = g.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k]); ro = o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
        print(which, k, rg.n_voxel_overflow, ro.n_voxel_overflow, rg.n_query, ro.n_query, eq(g.get_map(), o.get_map()))
print('done %s %.1fs' % (which, time.time() - t))
