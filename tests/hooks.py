"""Test infrastructure: the library's test hooks, called through a copy of the library that was built with them
(-DERASOR_HIP_TEST_HOOKS -> tests/_build/liberasor_hip_hooks.so).  The product library does not export these entry points and the
package has no wrapper for them any more (round 4)."""
import contextlib
import ctypes as C
import os

import numpy as np

import erasor_amd

HERE = os.path.dirname(os.path.abspath(__file__))
HOOKS_LIB = os.path.join(HERE, "_build", "liberasor_hip_hooks.so")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@contextlib.contextmanager
def hooks_library():
    """inside the block erasor_amd loads the hooks build instead of the product library (under ERASOR_TEST_SIMT_LIB the CPU stand-in
    build, which is compiled with the hooks, is in place already)"""
    if os.environ.get("ERASOR_TEST_SIMT_LIB"):
        yield
        return
    # (built here, by the tests that need it -- `make hooks`; the product build, plain `make`, neither compiles nor writes it)
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(HERE), "erasor_amd", "csrc"), "-s", "hooks"])
    keep = (erasor_amd.LIB_PATH, erasor_amd._lib)
    erasor_amd.LIB_PATH, erasor_amd._lib = HOOKS_LIB, None
    try:
        yield
    finally:
        erasor_amd.LIB_PATH, erasor_amd._lib = keep


def probe_math(g, x, y):
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    o = [np.zeros_like(x) for _ in range(3)]
    g._check(erasor_amd.lib().erasor_hip_probe_math(g._h, _p(x), _p(y), C.c_size_t(len(x)), _p(o[0]), _p(o[1]), _p(o[2])))
    return o


def probe_bin_keys(g, pts):
    pts = np.ascontiguousarray(pts, np.float32)
    kf = np.zeros(len(pts), np.uint32)
    ke = np.zeros(len(pts), np.uint32)
    ctr = np.zeros(4, np.uint32)
    g._check(erasor_amd.lib().erasor_hip_probe_bin_keys(g._h, _p(pts), C.c_size_t(len(pts)), _p(kf), _p(ke), _p(ctr)))
    return kf, ke, ctr


def exact_sort_u32(g, keys, vals):
    keys = np.ascontiguousarray(keys, np.uint32).copy()
    vals = np.ascontiguousarray(vals, np.uint32).copy()
    nf = C.c_uint32(0)
    g._check(erasor_amd.lib().erasor_hip_exact_sort_u32(g._h, _p(keys), _p(vals), C.c_size_t(len(keys)), C.byref(nf)))
    return keys, vals, int(nf.value)


def radix_sort_u32(g, keys, bits):
    keys = np.ascontiguousarray(keys, np.uint32)
    ko = np.zeros_like(keys)
    po = np.zeros_like(keys)
    g._check(erasor_amd.lib().erasor_hip_radix_sort_u32(g._h, _p(keys), C.c_size_t(len(keys)), C.c_int(bits), _p(ko), _p(po)))
    return ko, po


def debug_rebuild_outskirts(g):
    g._check(erasor_amd.lib().erasor_hip_debug_rebuild_outskirts(g._h))
