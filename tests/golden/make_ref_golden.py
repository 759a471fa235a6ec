"""Regenerates tests/golden/ref_*.npz: frozen synthetic inputs + the outputs of the REFERENCE ITSELF.

The outputs come from oracle/_ref/liberasor_ref.so = /root/reference's erasor.cpp, erasor_utils.cpp and
OfflineMapUpdater.cpp compiled unmodified (oracle/ref.mk) and driven through OfflineMapUpdater::callback_node.
/root/reference only exists in the build container, so the vectors are committed; tests/test_golden.py replays them
through the CPU oracle (no GPU) and through the HIP path via the C ABI (-m gpu).
Run from the repo root:  python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from erasor_amd import synth  # noqa: E402
from oracle import orc, ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PARAM_FIELDS = [f for f, _ in orc.Params._fields_ if not f.startswith("reserved")]
L2B = [0, 0, synth.LIDAR_HEIGHT, 0, 0, 0, 1]


def make(name, seq, version, seed, n_steps, az, length, **over):
    w = synth.World(seed=seed, length=length, n_moving=5, n_peds=3)
    m, scans, poses = w.accumulate_map(range(0, 2 * n_steps, 2), synth.Lidar.hdl64(az))
    p = orc.params_default()
    synth.apply_params(p, seq, version=version)
    for k, v in over.items():
        setattr(p, k, v)
    r = ref.RefUpdater(p, m, L2B)
    out = {"map0": m, "l2b_pose": np.array(L2B, np.float64), "n_steps": n_steps, "version": version,
           "param_names": np.array(PARAM_FIELDS), "params": np.array([float(getattr(p, f)) for f in PARAM_FIELDS])}
    rev = 0
    for k in range(n_steps):
        r.step(scans[k], poses[k])
        Tl, Tb = r.get_matrices()
        out["scan%d" % k] = scans[k]
        out["pose%d" % k] = np.asarray(poses[k], np.float64)
        out["T_l2b"] = Tl                       # the reference's own geoPose2eigen results (utils.cpp:35-55)
        out["T_b2o%d" % k] = Tb
        out["T_o2b%d" % k] = orc.invert4(Tb)   # = the stub's Matrix4f::inverse() (restated: double cofactors, OMU.cpp:436)
        out["query%d" % k] = r.get_cloud(0)
        out["static_estimate%d" % k] = r.get_cloud(2)
        out["map_rejected%d" % k] = r.get_cloud(4)
        out["curr_rejected%d" % k] = r.get_cloud(5)
        out["ground%d" % k] = r.get_cloud(6)
        out["likelihood%d" % k] = r.polygon_likelihood()      # SRT status polygons, push order
        if version == 3:
            out["status%d" % k] = r.get_status()
        for which in (0, 1):
            c, mn, mx = r.get_bins(which)
            out["bins%d_cnt%d" % (which, k)], out["bins%d_min%d" % (which, k)], out["bins%d_max%d" % (which, k)] = c, mn, mx
        n, d, th = r.get_planes()
        out["plane_n%d" % k], out["plane_last_d%d" % k] = n, np.array([d, th])
        out["labels%d" % k] = np.array(r.label_counts(), np.int64)
        out["n_map%d" % k] = len(r.get_map())
        rev += len(n)
    out["map_final"] = r.get_map()
    out["saved_0_2"] = r.save_static_map(0.2)   # save_static_map (OMU.cpp:174-196)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "map", m.shape, "->", out["map_final"].shape, "scan", scans[0].shape, "plane fits", rev)


if __name__ == "__main__":
    assert ref.build(), "needs /root/reference"
    make("ref_seq05_v3", "05", 3, 20210411, 3, 240, 90.0)
    make("ref_seq00_v3", "00", 3, 20210412, 2, 240, 90.0)
    make("ref_seq05_v2", "05", 2, 20210413, 2, 240, 90.0)
    make("ref_large_scale_v3", "large_scale_05", 3, 20210414, 3, 240, 90.0, is_large_scale=1, submap_size=25.0)
