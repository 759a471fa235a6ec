"""erasor_amd.evalmap (PR / RR / F1) against vectors produced by the reference's own scripts/analysis_runner.py"""
import os

import numpy as np

from erasor_amd import evalmap

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("gt_static", "gt_dynamic", "est_static", "est_dynamic", "preserved_static", "preserved_dynamic", "PR", "RR", "F1")


def test_port_matches_reference_evaluator_vectors():
    z = np.load(os.path.join(HERE, "golden", "eval_golden.npz"))
    for case in range(4):
        r = evalmap.evaluate_clouds(z["gt%d" % case], z["est%d" % case], 0.2)
        want = dict(zip(KEYS, z["res%d" % case].tolist()))
        for k in KEYS[:6]:
            assert r[k] == int(want[k]), (case, k, r[k], want[k])
        for k in KEYS[6:]:
            assert abs(r[k] - want[k]) < 1e-9, (case, k, r[k], want[k])


def test_label_decode_and_edge_cases():
    lab = np.array([252, 259, 251, 260, 252 + (5 << 16), 40 + (252 << 16)], np.float32)
    assert np.isin(evalmap.labels(lab), evalmap.DYNAMIC_CLASSES).tolist() == [True, True, False, False, True, False]
    gt = np.array([[0, 0, 0, 40], [1, 0, 0, 252], [5, 5, 5, 40]], np.float32)
    r = evalmap.evaluate_clouds(gt, gt[:1], 0.2)         # only the first static point survives
    assert (r["PR"], r["RR"]) == (50.0, 100.0)
    r = evalmap.evaluate_clouds(gt, gt, 0.2)             # nothing removed: PR 100, RR 0
    assert (r["PR"], r["RR"], r["F1"]) == (100.0, 0.0, 0.0)
