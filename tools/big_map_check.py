"""One-off scale check: the full step on a 4x tiled (~39 M-point) map, HIP path vs the oracle, plus timing.
Usage (GPU box): python tools/big_map_check.py [copies=4] [steps=3]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import erasor_amd
from erasor_amd import synth
from oracle import orc
import ctypes as C

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = synth.World(seed=20210305 + 5, length=1000.0, n_streets=5, street_gap=50.0, n_moving=10, n_peds=6)
lidar = synth.Lidar.hdl64(2000)
m0 = w.sample_map(spacing=0.2, frames=range(0, 320, 2), step=1.0)
tiles = [m0]
for c in range(1, copies):
    t = m0.copy()
    t[:, 1] += np.float32(400.0 * c)  # far outside every VoI: pure outskirts
    tiles.append(t)
m = np.concatenate(tiles)
print("map points:", len(m))
P = erasor_amd.params_default()
synth.apply_params(P, "05")
P.max_range, P.num_rings, P.num_sectors = 80.0, 20, 108
Tl = erasor_amd.geopose2eigen([0, 0, synth.LIDAR_HEIGHT, 0, 0, 0, 1])
jr = np.random.default_rng(1234)
scans, Tb, To = [], [], []
for k in range(steps):
    p7 = w.pose(k, 1.0, x0=0.0, jitter_rng=jr)
    scans.append(w.cast(p7, lidar, k))
    Tb.append(erasor_amd.geopose2eigen(p7))
    To.append(erasor_amd.invert_rigid(Tb[-1]))
g = erasor_amd.Erasor(P)
t = time.time(); g.set_map(m); print("set_map %.2f s" % (time.time() - t))
po = orc.Params(); C.memmove(C.byref(po), C.byref(P), C.sizeof(po))
o = orc.Oracle(po)
o.set_map(m)
for k in range(steps):
    t = time.time(); rg = g.step(scans[k], Tl, Tb[k], To[k]); tg = time.time() - t
    t = time.time(); ro = o.step(scans[k], Tl, Tb[k], To[k]); to = time.time() - t
    same = all(getattr(rg, f) == getattr(ro, f) for f in ("n_voi", "n_query", "n_static_estimate", "n_complement", "n_map_rejected", "n_map_out", "n_static", "n_dynamic", "n_reverted_bins"))
    print("step %d: gpu %.2f ms (host scan upload included), oracle %.0f ms, counters equal: %s, n_voi %d, map_out %d" % (k, tg * 1e3, to * 1e3, same, rg.n_voi, rg.n_map_out))
a, b = g.get_map(), o.get_map()
print("final maps bit-identical:", a.shape == b.shape and bool((a.view(np.uint32) == b.view(np.uint32)).all()))
