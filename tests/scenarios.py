"""Shared synthetic scenarios for the parity tests (deterministic; cached per process)."""
import functools

import numpy as np

from erasor_amd import synth
from oracle import orc


@functools.lru_cache(maxsize=8)
def small(seq="05", n_frames=12, az=500, length=200.0, version=3, seed=20210310, lidar="hdl64"):
    """~70 k-pt accumulated map, ~28 k-pt scans, seq-shaped parameters."""
    w = synth.World(seed=seed, length=length)
    lid = synth.Lidar.hdl64(az) if lidar == "hdl64" else synth.Lidar.ouster128(az)
    frames = list(range(0, 2 * n_frames, 2))
    m, scans, poses = w.accumulate_map(frames, lid, step=1.0)
    p = orc.params_default()
    synth.apply_params(p, seq, version=version)
    Tl = orc.geopose2eigen([0, 0, synth.LIDAR_HEIGHT, 0, 0, 0, 1])  # config/seq_05.yaml:32
    Tb = [orc.geopose2eigen(p7) for p7 in poses]
    To = [orc.invert4(t) for t in Tb]
    return dict(map=m, scans=scans, poses=poses, params=p, T_l2b=Tl, T_b2o=Tb, T_o2b=To, seq=seq, version=version)


def to_product_params(p):
    """copy an oracle Params into the product's Params (same layout, distinct ctypes class)"""
    import ctypes as C
    import erasor_amd
    q = erasor_amd.Params()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(q))
    return q
