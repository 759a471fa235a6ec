"""Regenerates tests/golden/*.npz: frozen synthetic inputs + the CPU oracle's outputs.

The reference has no golden vectors and cannot be compiled here, so these fixtures freeze the oracle
(oracle/erasor_oracle.cpp) at the commit that was reviewed against the reference line by line; they
catch regressions of the oracle and give the HIP path a data-only target on the GPU box.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from erasor_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PARAM_FIELDS = [f for f, _ in orc.Params._fields_ if f != "reserved_"]


def make(name, seq, version, seed, n_steps, az, length):
    w = synth.World(seed=seed, length=length, n_moving=5, n_peds=3)
    lid = synth.Lidar.hdl64(az)
    m, scans, poses = w.accumulate_map(range(0, 2 * n_steps, 2), lid)
    p = orc.params_default()
    synth.apply_params(p, seq, version=version)
    Tl = orc.geopose2eigen([0, 0, synth.LIDAR_HEIGHT, 0, 0, 0, 1])
    o = orc.Oracle(p)
    o.set_map(m)
    out = {"map0": m, "T_l2b": Tl, "n_steps": n_steps, "params": np.array([float(getattr(p, f)) for f in PARAM_FIELDS])}
    for k in range(n_steps):
        Tb = orc.geopose2eigen(poses[k])
        To = orc.invert4(Tb)
        r = o.step(scans[k], Tl, Tb, To)
        out["scan%d" % k] = scans[k]
        out["T_b2o%d" % k] = Tb
        out["T_o2b%d" % k] = To
        out["res%d" % k] = np.array(list(r.as_dict().values()), np.int64)
        out["rejidx%d" % k] = o.get_rejected_indices()
        out["status%d" % k] = o.get_status()
        b, n, d = o.get_planes()
        out["plane_bins%d" % k], out["plane_n%d" % k], out["plane_d%d" % k] = b, n, d
        out["query%d" % k] = o.get_cloud(0)
    out["map_final"] = o.get_map()
    out["res_keys"] = np.array(list(r.as_dict().keys()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "map", m.shape, "->", out["map_final"].shape, "scan", scans[0].shape)


if __name__ == "__main__":
    make("seq05_v3", "05", 3, 20210401, 3, 240, 90.0)
    make("seq07_v3", "07", 3, 20210402, 2, 240, 90.0)
    make("seq05_v2", "05", 2, 20210403, 2, 240, 90.0)
