"""The reference-compatible C++ surface (erasor_amd/csrc/shim) on the GPU: erasor::OfflineMapUpdater driven
by the ROS-free offline driver, and the ERASOR class used on its own — both against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import scenarios

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "erasor_amd", "erasor_offline_demo")


def write_pcd_binary(path, c):
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
                 "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (len(c), len(c))).encode())
        f.write(np.ascontiguousarray(c, np.float32).tobytes())


def ensure_demo():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "erasor_amd", "csrc", "shim"), "-s"])
    assert os.path.exists(DEMO)


def test_offline_map_updater_shim_matches_oracle(tmp_path):
    from oracle import orc
    ensure_demo()
    sc = scenarios.small()
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "pcds"))
    write_pcd_binary(os.path.join(d, "map.pcd"), sc["map"])
    n = 5
    with open(os.path.join(d, "poses.csv"), "w") as f:
        f.write("index,timestamp,x,y,z,qx,qy,qz,qw\n")
        for k in range(n):
            write_pcd_binary(os.path.join(d, "pcds", "%06d.pcd" % k), sc["scans"][k])
            f.write("%d,%d,%s\n" % (k, k, ",".join("%.17g" % v for v in sc["poses"][k])))
    out = subprocess.run([DEMO, d, str(n), "3", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    o = orc.Oracle(sc["params"])
    o.set_map(sc["map"])
    for k in range(n):
        o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
    want = o.get_map()
    got = np.fromfile(os.path.join(d, "map_final.bin"), np.float32).reshape(-1, 4)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # save_static_map(0.2): voxelize_preserving_labels over the whole map, ASCII PCD with 8 significant digits (OMU.cpp:186-193)
    saved = np.loadtxt(os.path.join(d, "05_result.pcd"), skiprows=11, dtype=np.float64).reshape(-1, 4)
    ref = orc.voxelize_preserving_labels(want, 0.2)
    assert saved.shape == ref.shape
    assert np.allclose(saved[:, :3], ref[:, :3], rtol=2e-7, atol=1e-7) and np.array_equal(saved[:, 3], ref[:, 3].astype(np.float64))


def test_removal_interval_gate(tmp_path):
    """callback_node only works on every removal_interval-th message (OMU.cpp:206-209)"""
    ensure_demo()
    sc = scenarios.small()
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "pcds"))
    write_pcd_binary(os.path.join(d, "map.pcd"), sc["map"])
    with open(os.path.join(d, "poses.csv"), "w") as f:
        f.write("header\n")
        for k in range(4):
            write_pcd_binary(os.path.join(d, "pcds", "%06d.pcd" % k), sc["scans"][k])
            f.write("%d,%d,%s\n" % (k, k, ",".join("%.17g" % v for v in sc["poses"][k])))
    out = subprocess.run([DEMO, d, "4", "3", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    from oracle import orc
    o = orc.Oracle(sc["params"])
    o.set_map(sc["map"])
    for k in (1, 3):  # stack_count 2 and 4
        o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
    got = np.fromfile(os.path.join(d, "map_final.bin"), np.float32).reshape(-1, 4)
    assert np.array_equal(got.view(np.uint32), o.get_map().view(np.uint32))


@pytest.mark.parametrize("version", [3, 2])
def test_erasor_class_on_egocentric_clouds(tmp_path, version):
    """ERASOR::set_inputs / compare_* / get_static_estimate / get_outliers exactly as OMU.cpp:266-284 calls them"""
    from oracle import orc
    ensure_demo()
    sc = scenarios.small(version=version)
    o = orc.Oracle(sc["params"])
    o.set_map(sc["map"])
    o.step(sc["scans"][0], sc["T_l2b"], sc["T_b2o"][0], sc["T_o2b"][0])
    map_voi, query_voi = o.get_cloud(1), o.get_cloud(0)
    d = str(tmp_path)
    map_voi.tofile(os.path.join(d, "map_voi.bin"))
    query_voi.tofile(os.path.join(d, "query_voi.bin"))
    out = subprocess.run([DEMO, "--erasor-class", os.path.join(d, "map_voi.bin"), os.path.join(d, "query_voi.bin"), os.path.join(d, "out"), str(version)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rd = lambda nm: np.fromfile(os.path.join(d, "out_%s.bin" % nm), np.float32).reshape(-1, 4)  # noqa: E731
    for nm, which in (("arranged", 2), ("complement", 3), ("ground_viz", 6)):
        want = o.get_cloud(which)
        got = rd(nm)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), nm
    # debug_map_rejected is egocentric inside ERASOR (the oracle hands it out in the map frame): same count here
    assert len(rd("map_rejected")) == len(o.get_cloud(4))
    bad = subprocess.run([DEMO, "--erasor-class", os.path.join(d, "map_voi.bin"), os.path.join(d, "query_voi.bin"), os.path.join(d, "o2"), "4"],
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode == 1 and "not implemented" in bad.stderr   # std::invalid_argument, as OMU.cpp:274
