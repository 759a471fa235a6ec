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


def test_device_sort_code_runs_lane_by_lane_on_the_cpu(tmp_path):
    """exact_sort.hip.h itself -- wave_partition, wave_small_subtree, block_partition, block_esort, exactly as the kernels
    compile them -- executed on the CPU by tests/cpp/simt_emu (one thread per lane, cross-lane intrinsics as wavefront
    barriers) and pinned to the real std::sort: sizes around every threshold (16 / 64 / 2048 keys), ties, sorted and
    reversed input, the median-of-3 adversary (heapsort fallback), a 256-thread workgroup."""
    exe = str(tmp_path / "esort_simt_check")
    subprocess.check_call(["g++", "-O1", "-std=c++20", "-pthread", "-I" + os.path.join(HERE, "cpp", "simt_emu"), "-o", exe,
                           os.path.join(HERE, "cpp", "esort_simt_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr
    assert "MISMATCH" not in out.stdout
    fallback = [int(line.split("heapsorts=")[1].split()[0]) for line in out.stdout.splitlines() if "adversary" in line]
    assert fallback and all(f > 0 for f in fallback), "the heapsort fallback was not exercised"
