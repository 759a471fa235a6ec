import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# round 5: the library overlaps consecutive steps only where it pays (dense bins, see ERASOR_HIP_OVERLAP in erasor_hip.hip); the test
# scenes are small, so the suite FORCES the overlapped path -- it is the one with the most moving parts -- and a few cases run without
# (test_alternative_launch_paths_keep_parity).  Read once by the library, before its first step.
os.environ.setdefault("ERASOR_HIP_OVERLAP", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # tests/test_full_step_on_cpu.py re-runs `-m gpu` tests in a helper process against the SAME host code and kernels compiled
    # for the CPU stand-in of the HIP runtime (tests/cpp/simt_emu): in that process -- and only there -- the ctypes wrapper
    # loads that library.  Test infrastructure; the product knows nothing of it.
    simt = os.environ.get("ERASOR_TEST_SIMT_LIB")
    if simt:
        import erasor_amd
        erasor_amd.LIB_PATH = simt
        erasor_amd._lib = None
        erasor_amd.build = lambda force=False: simt
