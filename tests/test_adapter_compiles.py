"""No-GPU check that the ROS1 adapter and the shim compile as ROS/PCL code: -DERASOR_SHIM_WITH_PCL -DERASOR_SHIM_WITH_ROS against the
stand-in ros / pcl / Eigen / message headers of oracle/stubs (the ones the reference's own sources are compiled against)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ros1_adapter_and_shim_compile_against_ros_pcl_api():
    shim = os.path.join(ROOT, "erasor_amd", "csrc", "shim")
    for src in ("ros1_adapter.cpp", "erasor_shim.cpp", "erasor_io.cpp"):
        cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DERASOR_SHIM_WITH_PCL", "-DERASOR_SHIM_WITH_ROS",
               "-I" + os.path.join(ROOT, "oracle", "stubs"), "-I" + shim, os.path.join(shim, src)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-4000:]
