#!/usr/bin/env python3
"""Writes bench.py's default workload (config 2: the ~10 M-point synthetic map, HDL-64 scans, seq05 parameters on the 20 x 108 @ 80 m
R-POD) as raw files for the C++ drop-in benchmark (erasor_offline_demo --bench <dir> ...):
   params.bin (erasor_params), map.bin (float32 xyzi), scan_%06d.bin, poses.bin (7 float64 per node: x y z qx qy qz qw), l2b.bin (7 float64)
usage: tools/export_cpp_bench.py <dir> [n_frames] [--small]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the same WORKLOADS / make_params / world as the Python benchmark)
import erasor_amd  # noqa: E402
from erasor_amd import synth  # noqa: E402

out = sys.argv[1]
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 30
small = "--small" in sys.argv
os.makedirs(out, exist_ok=True)
args = bench.parse_args.__wrapped__() if hasattr(bench.parse_args, "__wrapped__") else None
wl = bench.WORKLOADS["seq05"]


class A:  # what make_params / make_lidar read
    large_scale_mode = "off"
    az_steps = 360 if small else 0


P = bench.make_params(wl, A)
lidar = bench.make_lidar(wl, A)
if small:
    world = synth.World(seed=20210305 + 5, length=200.0)
    m = world.sample_map(spacing=0.3, frames=range(0, 40, 2), step=1.0)
else:
    world = synth.World(seed=20210305 + 5, length=1000.0, n_streets=5, street_gap=50.0, n_moving=10, n_peds=6)
    m = world.sample_map(spacing=wl["spacing"], frames=range(0, 320, 2), step=1.0)
open(os.path.join(out, "params.bin"), "wb").write(bytes(C.string_at(C.addressof(P), C.sizeof(P))))
np.ascontiguousarray(m, np.float32).tofile(os.path.join(out, "map.bin"))
jr = np.random.default_rng(1234)
poses = []
for k in range(n_frames):
    p7 = world.pose(k, 1.0, x0=300.0 if not small else 0.0, jitter_rng=jr)
    s = world.cast(p7, lidar, k)
    np.ascontiguousarray(s, np.float32).tofile(os.path.join(out, "scan_%06d.bin" % k))
    poses.append([float(v) for v in p7])
np.asarray(poses, np.float64).tofile(os.path.join(out, "poses.bin"))
np.asarray([0, 0, synth.LIDAR_HEIGHT, 0, 0, 0, 1], np.float64).tofile(os.path.join(out, "l2b.bin"))
print("exported %d-point map, %d scans (~%d points each) to %s" % (len(m), n_frames, len(s), out))
