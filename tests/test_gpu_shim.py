"""The reference-compatible C++ surface (erasor_amd/csrc/shim) on the GPU: erasor::OfflineMapUpdater driven
by the ROS-free offline driver, and the ERASOR class used on its own — both against the CPU oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import scenarios

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# where liberasor_hip.so / liberasor_shim.so / erasor_offline_demo live.  ERASOR_TEST_SHIM_DIR: a directory that holds the
# tests' CPU stand-in build of liberasor_hip.so (tests/test_full_step_on_cpu.py); the shim and the driver are then built there too
LIBDIR = os.environ.get("ERASOR_TEST_SHIM_DIR") or os.path.join(ROOT, "erasor_amd")
DEMO = os.path.join(LIBDIR, "erasor_offline_demo")


def write_pcd_binary(path, c):
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
                 "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (len(c), len(c))).encode())
        f.write(np.ascontiguousarray(c, np.float32).tobytes())


def ensure_demo():
    cmd = ["make", "-C", os.path.join(ROOT, "erasor_amd", "csrc", "shim"), "-s"]
    if os.environ.get("ERASOR_TEST_SHIM_DIR"):
        cmd.append("LIBDIR=" + LIBDIR)
    subprocess.check_call(cmd)
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
    # public R-PODs (erasor.h:143-145) and is_dynamic_obj_close (erasor.h:132, erasor.cpp:573-595)
    p = sc["params"]
    R, S = p.num_rings, p.num_sectors
    rows = np.loadtxt(os.path.join(d, "out_rpod.txt")).reshape(-1, 11)
    idx = (rows[:, 0] * S + rows[:, 1]).astype(int)
    (mc, mmin, mmax), (cc, cmin, cmax) = o.get_bins(0), o.get_bins(1)
    assert np.array_equal(rows[:, 2], mc[idx]) and np.array_equal(rows[:, 5], cc[idx])
    assert np.array_equal(rows[:, 3], mmin[idx]) and np.array_equal(rows[:, 4], mmax[idx])
    assert np.array_equal(rows[:, 6], cmin[idx]) and np.array_equal(rows[:, 7], cmax[idx])
    st = o.get_status()
    assert np.array_equal(rows[:, 9], st[idx])
    # flattening the R-PODs the way r_pod2pc does gives back: the binned part of map_voi / query_voi, and the bins part of the estimate
    code, _ = o.get_voi_codes()
    theta_major = lambda cl, cd: cl[np.argsort((cd % S) * R + cd // S, kind="stable")]  # noqa: E731
    want_map = theta_major(map_voi[code >= 0], code[code >= 0])
    got_map = rd("rpod_map")
    assert got_map.shape == want_map.shape and np.array_equal(got_map.view(np.uint32), want_map.view(np.uint32))
    n_ground = len(o.get_cloud(6))
    want_sel = o.get_cloud(2)[: len(o.get_cloud(2)) - n_ground]
    got_sel = rd("rpod_selected")
    assert got_sel.shape == want_sel.shape and np.array_equal(got_sel.view(np.uint32), want_sel.view(np.uint32))
    assert len(rd("rpod_curr")) == int(cc.sum())
    if version == 3:  # BLOCKED bins are exactly the MERGE-candidates with a CURR_IS_HIGHER neighbour
        st2 = st.reshape(R, S)
        close = rows[:, 10].astype(bool)
        for r, t, c in zip(rows[:, 0].astype(int), rows[:, 1].astype(int), close):
            if st2[r, t] == 0.8:
                assert c
            if st2[r, t] == 0.25:
                assert not c
    bad = subprocess.run([DEMO, "--erasor-class", os.path.join(d, "map_voi.bin"), os.path.join(d, "query_voi.bin"), os.path.join(d, "o2"), "4"],
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode == 1 and "not implemented" in bad.stderr   # std::invalid_argument, as OMU.cpp:274


def test_free_function_voxelize_preserving_labels(tmp_path):
    """erasor_utils::voxelize_preserving_labels(Ptr, Cloud&, double) (utils.hpp:103) through the shim"""
    from oracle import orc
    ensure_demo()
    sc = scenarios.small()
    d = str(tmp_path)
    for k, (cloud, leaf) in enumerate(((sc["scans"][0], 0.2), (sc["map"][:40000], 0.5))):
        src, dst = os.path.join(d, "in%d.bin" % k), os.path.join(d, "out%d.bin" % k)
        np.ascontiguousarray(cloud, np.float32).tofile(src)
        out = subprocess.run([DEMO, "--voxelize", src, repr(leaf), dst], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        got = np.fromfile(dst, np.float32).reshape(-1, 4)
        want = orc.voxelize_preserving_labels(cloud, leaf)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


ADAPTER = os.path.join(LIBDIR if os.environ.get("ERASOR_TEST_SHIM_DIR") else os.path.join(ROOT, "tests", "cpp"), "ros1_adapter_check")


def build_adapter_check():
    """ros1_adapter.cpp + the shim, compiled with -DERASOR_SHIM_WITH_PCL -DERASOR_SHIM_WITH_ROS against the stand-in ROS / PCL
    headers of oracle/stubs (this image has neither)"""
    shim = os.path.join(ROOT, "erasor_amd", "csrc", "shim")
    cmd = ["g++", "-O1", "-std=c++17", "-DERASOR_SHIM_WITH_PCL", "-DERASOR_SHIM_WITH_ROS", "-I" + os.path.join(ROOT, "oracle", "stubs"), "-I" + shim,
           "-o", ADAPTER, os.path.join(ROOT, "tests", "cpp", "ros1_adapter_check.cpp"), os.path.join(shim, "erasor_shim.cpp"),
           os.path.join(shim, "erasor_io.cpp"), "-L" + LIBDIR, "-lerasor_hip",
           "-Wl,-rpath," + LIBDIR]
    subprocess.check_call(cmd)


@pytest.mark.parametrize("version,interval,hold", [(3, 1, 0), (3, 2, 0), (2, 1, 0), (3, 1, 1)])
def test_ros1_adapter_node(tmp_path, version, interval, hold):
    """The thin ROS1 node (erasor_amd/csrc/shim/ros1_adapter.cpp): erasor::node messages in through the subscribed callback,
    the reference's topics out (OMU.cpp:5-23, 316-326; SRT polygons erasor.cpp:496-570, 630-670) — against the oracle."""
    from oracle import orc
    build_adapter_check()
    sc = scenarios.small(version=version)
    d = str(tmp_path)
    write_pcd_binary(os.path.join(d, "map.pcd"), sc["map"])
    n = 4
    with open(os.path.join(d, "poses.txt"), "w") as f:
        for k in range(n):
            np.ascontiguousarray(sc["scans"][k], np.float32).tofile(os.path.join(d, "scan%d.bin" % k))
            f.write(" ".join("%.17g" % v for v in sc["poses"][k]) + "\n")
    # hold = 1: /MapUpdater/lookahead_hold -- node k is processed when node k+1 arrives (announced first: look-ahead under ROS), the
    # last one when /saveflag arrives; the publications are the same, one message later
    out = subprocess.run([ADAPTER, d, str(n), str(version), str(interval), str(hold)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    o = orc.Oracle(sc["params"])
    o.set_map(sc["map"])
    p = sc["params"]
    R, S = p.num_rings, p.num_sectors
    rd = lambda nm, j: np.fromfile(os.path.join(d, "out_%s_%d.bin" % (nm, j)), np.float32)  # noqa: E731
    same = lambda a, b: a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))  # noqa: E731
    j = 0
    for k in range(n):
        if (k + 1) % interval != 0:
            continue  # "PASS!" (OMU.cpp:206-209)
        ro = o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
        assert same(rd("map_rejected", j).reshape(-1, 4), o.get_cloud(4))
        assert same(rd("curr_rejected", j).reshape(-1, 4), o.get_cloud(5))
        assert same(rd("map_body", j).reshape(-1, 4), o.get_cloud(1))
        assert same(rd("pc_curr_body", j).reshape(-1, 4), o.get_cloud(0))
        assert same(rd("pc2_curr", j).reshape(-1, 4), orc.transform(o.get_cloud(0), sc["T_b2o"][k]))
        m = o.get_map()
        from erasor_amd import synth
        dyn = synth.is_dynamic(m[:, 3])
        assert same(rd("static", j).reshape(-1, 4), m[~dyn]) and same(rd("dynamic", j).reshape(-1, 4), m[dyn])
        assert (len(m[~dyn]), len(m[dyn])) == (ro.n_static, ro.n_dynamic)
        like = rd("likelihood", j)
        assert np.array_equal(like, o.get_status().reshape(R, S).T.reshape(-1).astype(np.float32))
        poly = rd("polygon0", j).reshape(-1, 3)  # bin ring 1, sector 2: set_polygons(1, 2, 3), erasor.cpp:630-670
        rs, ss = p.max_range / R, 2 * 3.1415926535 / S
        ang = [2 * ss, 2 * ss, 2 * ss + ss / 3, 2 * ss + 2 * ss / 3, 3 * ss, 3 * ss, 3 * ss - ss / 3, 3 * ss - 2 * ss / 3]
        rad = [rs, 2 * rs, 2 * rs, 2 * rs, 2 * rs, rs, rs, rs]
        want = np.array([[r_ * np.cos(a), r_ * np.sin(a), p.max_h + 0.5] for r_, a in zip(rad, ang)], np.float32)
        assert poly.shape == want.shape and np.allclose(poly, want, atol=1e-5)
        j += 1
    assert ("processed %d" % j) in out.stdout
    # /saveflag -> save_static_map(0.2) -> <save_path>/<data_name>_result.pcd (OMU.cpp:169-196)
    saved = np.loadtxt(os.path.join(d, "05_result.pcd"), skiprows=11, dtype=np.float64).reshape(-1, 4)
    ref_saved = orc.voxelize_preserving_labels(o.get_map(), 0.2)
    assert saved.shape == ref_saved.shape and np.array_equal(saved[:, 3], ref_saved[:, 3].astype(np.float64))


def test_config_driver_like_main_in_your_env(tmp_path):
    """erasor_offline_demo --config <rosparam.yaml>: the reference's own driver flow (main_in_your_env.cpp:61-127) —
    YAML in the reference's layout, <data_dir>/poses_lidar2body.csv (float parse, Quaternionf -> matrix ->
    eigen2geoPose -> node.odom -> geoPose2eigen), <data_dir>/pcds/%06d.pcd from init_idx, save_static_map(0.2)."""
    import ctypes as C
    from oracle import orc
    from erasor_amd import synth
    ensure_demo()
    sc = scenarios.small()
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "pcds"))
    os.makedirs(os.path.join(d, "out"))
    write_pcd_binary(os.path.join(d, "dense_global_map.pcd"), sc["map"])
    n, init_idx = 6, 1
    with open(os.path.join(d, "poses_lidar2body.csv"), "w") as f:
        f.write("index, timestamp, x, y, z, qx, qy, qz, qw\n")
        for k in range(n):
            write_pcd_binary(os.path.join(d, "pcds", "%06d.pcd" % k), sc["scans"][k])
            f.write("%d, %.2f, %s\n" % (k, 0.1 * k, ", ".join("%.9f" % v for v in sc["poses"][k])))
    sp = synth.SEQ_PARAMS["05"]
    with open(os.path.join(d, "cfg.yaml"), "w") as f:
        f.write("erasor:\n")
        for key in ("max_range", "num_rings", "num_sectors", "min_h", "max_h", "th_bin_max_h", "scan_ratio_threshold", "minimum_num_pts",
                    "gf_dist_thr", "gf_iter", "gf_num_lpr", "gf_th_seeds_height"):
            f.write("    %s: %s # from config/seq_05.yaml\n" % (key, sp[key]))
        f.write("    rejection_ratio: 0\n    version: 3\n\nMapUpdater:\n    data_name: \"05\"\n    env: \"outdoor\"\n")
        f.write("    save_path: \"%s\"\n    query_voxel_size: 0.2\n    map_voxel_size: 0.05\n    removal_interval: 1\n\n" % os.path.join(d, "out"))
        f.write("data_dir: \"%s\"\ninit_idx: %d\ninterval: 2\nvoxel_size: 0.075\n" % (d, init_idx))
        f.write("tf:\n     lidar2body: [0.0, 0.0, %s, 0, 0.0, 0.0, 1.0] # xyz q_x, q_y, q_z, q_w in order\n\nverbose: true\n" % synth.LIDAR_HEIGHT)
    out = subprocess.run([DEMO, "--config", os.path.join(d, "cfg.yaml")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Static map building complete!" in out.stdout
    # the oracle, fed with the transforms the driver derives from the csv
    shim = C.CDLL(os.path.join(LIBDIR, "liberasor_shim.so"))
    shim.erasor_shim_load_poses.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    shim.erasor_shim_load_poses.restype = C.c_long
    T = np.zeros((n, 16), np.float32)
    geo = np.zeros((n, 7), np.float64)
    R = np.zeros((n, 16), np.float32)
    assert shim.erasor_shim_load_poses(os.path.join(d, "poses_lidar2body.csv").encode(), T.ctypes.data, geo.ctypes.data, R.ctypes.data, n) == n
    o = orc.Oracle(sc["params"])
    o.set_map(sc["map"])
    for k in range(init_idx, n):
        Tb = np.ascontiguousarray(R[k].reshape(4, 4))
        o.step(sc["scans"][k], sc["T_l2b"], Tb, orc.invert4(Tb))
    ref = orc.voxelize_preserving_labels(o.get_map(), 0.2)
    saved = np.loadtxt(os.path.join(d, "out", "05_result.pcd"), skiprows=11, dtype=np.float64).reshape(-1, 4)
    assert saved.shape == ref.shape
    # ASCII PCD with PCL's 8 significant digits (OMU.cpp:193)
    assert np.allclose(saved[:, :3], ref[:, :3], rtol=2e-7, atol=1e-7) and np.array_equal(saved[:, 3], ref[:, 3].astype(np.float64))


def _write_sequence_dir(d, sc, n, init_idx, data_name):
    """<d>/dense_global_map.pcd, pcds/, poses_lidar2body.csv, cfg.yaml in the reference's layout; returns the yaml's path"""
    from erasor_amd import synth
    os.makedirs(os.path.join(d, "pcds"))
    os.makedirs(os.path.join(d, "out"))
    write_pcd_binary(os.path.join(d, "dense_global_map.pcd"), sc["map"])
    with open(os.path.join(d, "poses_lidar2body.csv"), "w") as f:
        f.write("index, timestamp, x, y, z, qx, qy, qz, qw\n")
        for k in range(n):
            write_pcd_binary(os.path.join(d, "pcds", "%06d.pcd" % k), sc["scans"][k])
            f.write("%d, %.2f, %s\n" % (k, 0.1 * k, ", ".join("%.9f" % v for v in sc["poses"][k])))
    sp = synth.SEQ_PARAMS["05"]
    with open(os.path.join(d, "cfg.yaml"), "w") as f:
        f.write("erasor:\n")
        for key in ("max_range", "num_rings", "num_sectors", "min_h", "max_h", "th_bin_max_h", "scan_ratio_threshold", "minimum_num_pts",
                    "gf_dist_thr", "gf_iter", "gf_num_lpr", "gf_th_seeds_height"):
            f.write("    %s: %s\n" % (key, sp[key]))
        f.write("    rejection_ratio: 0\n    version: 3\n\nMapUpdater:\n    data_name: \"%s\"\n    env: \"outdoor\"\n" % data_name)
        f.write("    save_path: \"%s\"\n    query_voxel_size: 0.2\n    map_voxel_size: 0.05\n    removal_interval: 1\n\n" % os.path.join(d, "out"))
        f.write("data_dir: \"%s\"\ninit_idx: %d\ninterval: 2\nvoxel_size: 0.075\n" % (d, init_idx))
        f.write("tf:\n     lidar2body: [0.0, 0.0, %s, 0, 0.0, 0.0, 1.0]\n\nverbose: false\n" % synth.LIDAR_HEIGHT)
    return os.path.join(d, "cfg.yaml")


def test_sequences_from_a_queue_and_replicas_in_one_process(tmp_path):
    """round 4, BASELINE config 3 / VERDICT r03 item 5 in the C++ driver: (a) --queue: three sequences over two workers, the next sequence
    goes to whichever worker is idle (erasor::WorkQueue); every saved map equals what --config saves for that sequence alone.  (b)
    --replicas 2: ONE map file read, erasor_hip_replicate_map (two handles on this box's one device: peer copies), the nodes dealt
    over the replicas."""
    import json
    ensure_demo()
    sc = scenarios.small()
    cfgs = []
    for j in range(3):
        d = str(tmp_path / ("seq%d" % j))
        os.makedirs(d)
        cfgs.append(_write_sequence_dir(d, sc, 4 + j, 0, "0%d" % j))
    out = subprocess.run([DEMO, "--queue", "2", "100"] + cfgs, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    j = json.loads(out.stdout.strip().split("\n")[-1])
    assert j["jobs"] == 3 and j["failed"] == 0 and j["workers"] == 2
    lines = [l for l in out.stdout.split("\n") if l.startswith("job ")]
    assert len(lines) == 3 and {int(l.split("worker ")[1].split()[0]) for l in lines} <= {0, 1}
    assert ["%d nodes" % (4 + k) in lines[k] for k in range(3)] == [True] * 3
    # each sequence alone, into another directory: the same saved map
    for k in range(3):
        d2 = str(tmp_path / ("alone%d" % k))
        os.makedirs(d2)
        cfg2 = _write_sequence_dir(d2, sc, 4 + k, 0, "0%d" % k)
        o2 = subprocess.run([DEMO, "--config", cfg2], capture_output=True, text=True, timeout=600)
        assert o2.returncode == 0, o2.stdout + o2.stderr
        a = open(os.path.join(str(tmp_path / ("seq%d" % k)), "out", "0%d_result.pcd" % k)).read()
        b = open(os.path.join(d2, "out", "0%d_result.pcd" % k)).read()
        assert a == b, "sequence %d: the queue's saved map differs from the stand-alone run's" % k
    out = subprocess.run([DEMO, "--replicas", "2", cfgs[2], "6"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    j = json.loads(out.stdout.strip().split("\n")[-1])
    assert j["replicas"] == 2 and j["transport"] in ("peer copies", "rccl")
    reps = [l for l in out.stdout.split("\n") if l.startswith("replica ")]
    assert len(reps) == 2 and "3 nodes" in reps[0] and "3 nodes" in reps[1]


@pytest.mark.parametrize("large", [0, 1])
def test_mapgen_class_and_driver(tmp_path, large):
    """class mapgen of the shim (setValue / accumPointCloud / getPointClouds / saveNaiveMap, mapgen.hpp:182-303) driven
    like src/mapgen/main.cpp: the saved map is what OfflineMapUpdater::load_global_map reads next."""
    import ctypes as C
    from oracle import orc
    ensure_demo()
    sc = scenarios.small()
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "pcds"))
    n = 5
    with open(os.path.join(d, "poses_lidar2body.csv"), "w") as f:
        f.write("index, timestamp, x, y, z, qx, qy, qz, qw\n")
        for k in range(n):
            write_pcd_binary(os.path.join(d, "pcds", "%06d.pcd" % k), sc["scans"][k])
            f.write("%d, %.2f, %s\n" % (k, 0.1 * k, ", ".join("%.9f" % v for v in sc["poses"][k])))
    out = subprocess.run([DEMO, "--mapgen", d, str(n), "0.2", str(large)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    shim = C.CDLL(os.path.join(LIBDIR, "liberasor_shim.so"))
    shim.erasor_shim_load_poses.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    shim.erasor_shim_load_poses.restype = C.c_long
    T = np.zeros((n, 16), np.float32)
    geo = np.zeros((n, 7), np.float64)
    R = np.zeros((n, 16), np.float32)
    assert shim.erasor_shim_load_poses(os.path.join(d, "poses_lidar2body.csv").encode(), T.ctypes.data, geo.ctypes.data, R.ctypes.data, n) == n
    o = orc.Mapgen(np.float32(0.2), bool(large))
    for k in range(n):
        o.accum(sc["scans"][k], R[k].reshape(4, 4))
    got_map = np.fromfile(os.path.join(d, "mapgen_cloud_map.bin"), np.float32).reshape(-1, 4)
    got_curr = np.fromfile(os.path.join(d, "mapgen_cloud_curr.bin"), np.float32).reshape(-1, 4)
    assert np.array_equal(got_map.view(np.uint32), o.cloud_map.view(np.uint32))
    assert np.array_equal(got_curr.view(np.uint32), o.cloud_curr.view(np.uint32))
    name = os.path.join(d, "05_0_to_%d_w_interval1_voxel_0.200000.pcd" % (n - 1))  # main.cpp:34-38
    assert os.path.exists(name), os.listdir(d)
    saved = np.loadtxt(name, skiprows=11, dtype=np.float64).reshape(-1, 4)
    ref = o.save()
    assert saved.shape == ref.shape
    assert np.allclose(saved[:, :3], ref[:, :3], rtol=2e-7, atol=1e-7) and np.array_equal(saved[:, 3], ref[:, 3].astype(np.float64))
    naive = np.loadtxt(os.path.join(d, "05_original.pcd"), skiprows=11, dtype=np.float64).reshape(-1, 4)
    assert naive.shape == o.naive_map().shape


def test_cpp_bench_mode_of_the_offline_driver(tmp_path):
    """erasor_offline_demo --bench: OfflineMapUpdater::callback_node with host clouds in and the rejected clouds out (with and
    without the next node announced) and the C ABI loop with device-resident scans, timed in C++ -- here on a small export of
    bench.py's workload: the three passes must agree on what they rejected and on the final map size, and print one JSON line."""
    import json
    d = str(tmp_path / "cppbench")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "export_cpp_bench.py"), d, "9", "--small"])
    out = subprocess.run([DEMO, "--bench", d, "5", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    j = json.loads(out.stdout.strip().split("\n")[-1])
    assert j["passes_agree"] is True and j["nodes_timed"] == 5 and j["ms_per_callback_nodes_announced_deep"] > 0
    assert j["ms_per_callback"] > 0 and j["ms_per_callback_next_node_announced"] > 0 and j["ms_per_step_device_resident_two_ahead"] > 0
