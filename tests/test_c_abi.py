"""The C-ABI library loads without a GPU and exports every symbol include/erasor_hip.h declares."""
import ctypes as C
import os
import re

import pytest

import erasor_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "erasor_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(erasor_hip_\w+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    erasor_amd.build()
    lib = C.CDLL(erasor_amd.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s


def test_version_and_default_params_match_reference_defaults():
    l = erasor_amd.lib()
    assert b"gfx950" in l.erasor_hip_version()
    p = erasor_amd.params_default()
    # erasor.h:47-61, OMU.cpp:66-81
    assert (p.max_range, p.num_rings, p.num_sectors) == (10.0, 20, 60)
    assert (p.max_h, p.min_h, p.th_bin_max_h, p.scan_ratio_threshold) == (3.0, 0.0, 0.39, 0.22)
    assert (p.num_lowest_pts, p.minimum_num_pts, p.rejection_ratio) == (5, 4, 0.33)
    assert (p.gf_dist_thr, p.gf_iter, p.gf_num_lpr, p.gf_th_seeds_height, p.map_voxel_size) == (0.05, 3, 10, 0.5, 0.2)
    assert (p.version, p.query_voxel_size, p.removal_interval) == (3, 0.05, 2)


def test_struct_layouts_match_the_header():
    # sizes as laid out by a C compiler for include/erasor_hip.h (checked by compiling a probe)
    import subprocess, tempfile
    code = '#include <stdio.h>\n#include "erasor_hip.h"\nint main(){printf("%zu %zu\\n", sizeof(erasor_params), sizeof(erasor_step_result));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write(code)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, src])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert a == C.sizeof(erasor_amd.Params)
    assert b == C.sizeof(erasor_amd.StepResult)


def test_no_gpu_means_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = erasor_amd.params_default()
    with pytest.raises(erasor_amd.ErasorError) as e:
        erasor_amd.Erasor(p)
    assert e.value.rc == erasor_amd.E_NO_DEVICE


def test_product_never_touches_the_oracle():
    """the product path must not import / link / call anything under oracle/"""
    pkg = os.path.join(ROOT, "erasor_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt, "%s mentions the oracle" % os.path.join(dirpath, f)
                # ... nor of the CPU stand-in of the HIP runtime the tests compile the device code against (tests/cpp/simt_emu):
                # the package loads liberasor_hip.so and nothing else
                assert "simt" not in txt.lower(), "%s mentions the tests' CPU stand-in" % os.path.join(dirpath, f)
