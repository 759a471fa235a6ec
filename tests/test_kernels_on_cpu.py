"""The DEVICE kernels, unmodified, on the CPU.

tests/cpp/simt_emu/hip/hip_runtime.h stands in for <hip/hip_runtime.h>: one OS thread per lane, 64 lanes per wavefront,
ballots / shuffles / readlane as tagged exchanges between lanes, __syncthreads and the intra-wavefront fence as barriers,
__shared__ as the single workgroup's static storage.  tests/cpp/kernels_simt_check.cpp compiles erasor_amd/csrc/kernels.hip.h
against it and runs whole kernels, workgroup by workgroup:

  * the map store's VoI split (a step's own and the one a step launches ahead for its successor), the chunk scan and the
    gather against a scalar walk over [VoI-resident region | outskirts];
  * the map bucketing (histogram, column scan, scatter) against std::stable_sort; the run detection against a loop;
  * R-GPF (k_rgpf2) and the per-bin voxelisation (k_binvox2) against the oracle's per-bin functions, bit for bit.

It is test infrastructure like the oracle it links: nothing of it is on the product path, and it proves nothing about
timing -- but a kernel change can be checked for parity here before a GPU is spent on it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_kernels_run_lane_by_lane_on_the_cpu(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import orc
    orc.build()
    exe = str(tmp_path / "kernels_simt_check")
    subprocess.check_call(["g++", "-O1", "-std=c++20", "-pthread", "-ffp-contract=off", "-I" + os.path.join(HERE, "cpp", "simt_emu"), "-o", exe,
                           os.path.join(HERE, "cpp", "kernels_simt_check.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-lerasor_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=1200)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr
    assert "MISMATCH" not in out.stdout and "FAILED" not in out.stdout
    for what in ("run detection", "map store: split (launched ahead)", "map bucketing", "R-GPF (k_rgpf2)", "per-bin voxelisation (k_binvox2)"):
        assert what in out.stdout, what
