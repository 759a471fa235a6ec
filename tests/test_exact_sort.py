"""Pins erasor_amd/csrc/exact_sort_core.h (the parallel formulation of libstdc++ std::sort that the HIP
kernels execute) against the real std::sort of this toolchain, on the host."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parallel_formulation_matches_libstdcxx_sort(tmp_path):
    exe = str(tmp_path / "esort_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "cpp", "esort_check.cpp")])
    out = subprocess.run([exe, "300"], capture_output=True, text=True, timeout=600)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
    # the heapsort fallback (median-of-3 killer inputs) must have been exercised
    n_heap = int(out.stdout.split("exercised:")[1].split(")")[0])
    assert n_heap > 0
