"""The whole product path on the CPU: erasor_hip.hip (handle, map store, stream / event orchestration, look-ahead, the VoI
split launched ahead) and every kernel of kernels.hip.h, compiled UNMODIFIED against tests/cpp/simt_emu -- a stand-in for the
HIP runtime in which a launch runs its grid synchronously, workgroup by workgroup, the 64 lanes of a wavefront as fibers --
driven through the C ABI and compared with the oracle.

  * two steps of a small sequence with the nodes announced ahead (result block, map, VoI, static estimate, rejected clouds,
    bin statuses and planes: all bit-exact);
  * a part of the GPU parity suite itself (tests/test_gpu_parity.py, `-m gpu`) re-run against that library: the one-bin
    known answers, the edge cases (empty / one-point / ragged inputs), the stable bucketing, the exact sort's heapsort
    fallback, the API's error behaviour.  ERASOR_SIMT_MORE=1 adds the standalone voxelisation, the exact-sort sweep, the
    VoxelGrid index-overflow pass-through and the NaN refusal (~5 more minutes); ERASOR_SIMT_ALL=1 runs the whole file except
    the full-size cases (49 tests: every synthetic sequence, v2 / v3, submap mode, look-ahead, mapgen, PR / RR; ~40 minutes).

This is a checker, not a product: the library is built into the test's temporary directory, loaded by helper processes only
(ERASOR_TEST_SIMT_LIB), and is four to five orders of magnitude slower than the device.  `erasor_amd.lib()` loads
`liberasor_hip.so` and nothing else; a kernel change can be checked for parity here before a GPU is spent on it."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.timeout(7200)  # (pytest.ini's 300 s is for the device; the extended modes here take minutes to an hour)


@pytest.fixture(scope="module")
def simt_lib(tmp_path_factory):
    lib = str(tmp_path_factory.mktemp("simt") / "liberasor_hip_simt.so")
    subprocess.check_call(["g++", "-x", "c++", "-O1", "-std=c++20", "-pthread", "-ffp-contract=off", "-fPIC", "-shared", "-DERASOR_HIP_TEST_HOOKS",
                           "-I" + os.path.join(HERE, "cpp", "simt_emu"), "-o", lib, os.path.join(ROOT, "erasor_amd", "csrc", "erasor_hip.hip")])
    sys.path.insert(0, ROOT)
    from oracle import orc
    orc.build()
    return lib


def test_two_steps_of_the_real_host_code_and_kernels_match_the_oracle(simt_lib):
    out = subprocess.run([sys.executable, os.path.join(HERE, "simt_full_step.py"), simt_lib, ROOT, "2"], capture_output=True, text=True, timeout=900)
    sys.stdout.write(out.stdout)
    assert "FULL-STEP-OK" in out.stdout, out.stdout + out.stderr[-3000:]


def test_part_of_the_gpu_parity_suite_passes_on_the_cpu_stand_in(simt_lib):
    # (round 5: `api_error_two_handles...` -- five look-ahead steps of two handles, 4 minutes here since steps overlap -- runs with
    # ERASOR_SIMT_MORE / _ALL and on the GPU)
    keys = ["one_bin_known", "edge_cases", "stable_radix", "heapsort_fallback", "api_error_behaviour"]
    if os.environ.get("ERASOR_SIMT_MORE"):
        keys += ["voxelize_preserving_labels_standalone", "exact_std_sort", "voxelgrid_index_overflow", "non_finite", "device_libm", "api_error_two_handles"]
    expr = " or ".join(keys)
    if os.environ.get("ERASOR_SIMT_ALL"):  # everything but the full-size cases: 49 tests, ~40 minutes on 8 cores
        expr = "not (full_size or config4 or whole_map or long_segments or map_grows)"
    env = dict(os.environ, ERASOR_TEST_SIMT_LIB=simt_lib)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_gpu_hooks.py"), "-m", "gpu",
                          "-q", "-x", "-k", expr,
                          "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=7200 if os.environ.get("ERASOR_SIMT_ALL") else 2400, cwd=ROOT, env=env)
    tail = out.stdout[-1500:]
    sys.stdout.write(tail)
    assert out.returncode == 0 and " passed" in tail and "failed" not in tail, out.stdout[-4000:] + out.stderr[-2000:]
    n_passed = int(tail.split(" passed")[0].split()[-1])
    assert n_passed >= 9, tail  # 1 + 1 + 5 + 1 + 1


def test_the_references_own_vectors_are_reproduced_on_the_cpu_stand_in(simt_lib):
    """tests/golden/ref_*.npz were written by the reference's unmodified sources (oracle/_ref, tests/golden/make_ref_golden.py);
    tests/test_golden.py replays them through the C ABI with nothing but the committed data (-m gpu).  The same test, against
    the device code on the CPU stand-in: one vector by default, all seven golden files with ERASOR_SIMT_MORE=1."""
    expr = "hip_reproduces" if os.environ.get("ERASOR_SIMT_MORE") or os.environ.get("ERASOR_SIMT_ALL") else "reference_vectors and ref_seq05_v3"
    env = dict(os.environ, ERASOR_TEST_SIMT_LIB=simt_lib)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_golden.py"), "-m", "gpu", "-q", "-x", "-k", expr, "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=2400, cwd=ROOT, env=env)
    tail = out.stdout[-1500:]
    sys.stdout.write(tail)
    assert out.returncode == 0 and " passed" in tail and "failed" not in tail, out.stdout[-4000:] + out.stderr[-2000:]


@pytest.mark.skipif(not (os.environ.get("ERASOR_SIMT_MORE") or os.environ.get("ERASOR_SIMT_ALL")), reason="~8 minutes: ERASOR_SIMT_MORE=1")
def test_the_cpp_shim_suite_passes_on_the_cpu_stand_in(simt_lib, tmp_path):
    """tests/test_gpu_shim.py (the reference-compatible C++ surface: OfflineMapUpdater through the offline driver, class ERASOR,
    erasor_utils, class mapgen, the ROS1 node against stand-in ROS / PCL headers) with liberasor_shim.so and the driver built
    against the stand-in library: all eleven tests."""
    import shutil
    d = str(tmp_path / "shim")
    os.makedirs(d)
    shutil.copy(simt_lib, os.path.join(d, "liberasor_hip.so"))
    env = dict(os.environ, ERASOR_TEST_SIMT_LIB=os.path.join(d, "liberasor_hip.so"), ERASOR_TEST_SHIM_DIR=d)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_shim.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=3600, cwd=ROOT, env=env)
    tail = out.stdout[-1500:]
    sys.stdout.write(tail)
    assert out.returncode == 0 and "11 passed" in tail, out.stdout[-4000:] + out.stderr[-2000:]
