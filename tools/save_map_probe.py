"""whole-map voxelize_preserving_labels (save_static_map, OMU.cpp:186) on the GPU vs the oracle"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import erasor_amd
from erasor_amd import synth
from oracle import orc
streets = int(sys.argv[1]) if len(sys.argv) > 1 else 5
w = synth.World(seed=20210305 + 5, length=1000.0, n_streets=streets, street_gap=50.0, n_moving=10, n_peds=6)
m = w.sample_map(spacing=0.2, frames=range(0, 320, 2))
print("map", m.shape)
g = erasor_amd.Erasor(erasor_amd.params_default())
for leaf in (0.2, 0.4):
    t = time.time(); a = g.voxelize_preserving_labels(m, leaf); tg = time.time() - t
    t = time.time(); a = g.voxelize_preserving_labels(m, leaf); tg2 = time.time() - t
    t = time.time(); b = orc.voxelize_preserving_labels(m, leaf); to = time.time() - t
    ok = a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()
    print("leaf %.1f: gpu %.3f s (2nd %.3f s, incl. PCIe both ways), oracle %.2f s, out %s, bit-exact %s" % (leaf, tg, tg2, to, a.shape, ok))
