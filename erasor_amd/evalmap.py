"""Preservation Rate / Rejection Rate of a static map against a labelled ground-truth map.

Same protocol as the reference's evaluator (scripts/analysis_runner.py:74-105; README.md:196): for every GT point the
nearest estimated point (1-NN, Euclidean); a GT point is "preserved" if that distance is < voxelsize*sqrt(3)/2;
PR = preserved static / GT static, RR = 1 - preserved dynamic / GT dynamic (both in %), F1 of PR/100 and RR/100.
Labels: numeric cast of intensity, & 0xFFFF, dynamic classes 252..259 (analysis_runner.py:14,44-47).
Pinned against the reference implementation by tests/golden/eval_golden.npz (tests/golden/make_eval_golden.py).
"""
import numpy as np
from scipy.spatial import cKDTree

DYNAMIC_CLASSES = np.arange(252, 260)


def labels(intensity):
    return np.asarray(intensity).astype(np.uint32) & 0xFFFF


def evaluate(gt_xyz, gt_sem, est_xyz, est_sem, voxelsize=0.2):
    gt_xyz = np.asarray(gt_xyz, np.float32)
    est_xyz = np.asarray(est_xyz, np.float32)
    gt_dyn_all = np.isin(gt_sem, DYNAMIC_CLASSES)
    ns_gt, nd_gt = int((~gt_dyn_all).sum()), int(gt_dyn_all.sum())
    est_dyn_all = np.isin(est_sem, DYNAMIC_CLASSES)
    # (workers=-1: the query runs on every host core -- the result does not depend on it; a 10 M-point map takes seconds instead of a minute)
    dists, idx = cKDTree(est_xyz.astype(np.float64)).query(gt_xyz.astype(np.float64), k=1, workers=-1)
    is_in = dists < voxelsize * np.sqrt(3) / 2
    gt_is_dyn = gt_dyn_all[is_in]
    est_is_dyn = est_dyn_all[idx[is_in]]
    kept_s = int(np.sum((~gt_is_dyn) & (~est_is_dyn)))
    kept_d = int(np.sum(gt_is_dyn & est_is_dyn))
    pr = kept_s / ns_gt * 100.0
    rr = (nd_gt - kept_d) / nd_gt * 100.0 if nd_gt > 0 else 0.0
    f1 = 2 * (pr / 100) * (rr / 100) / ((pr / 100) + (rr / 100)) if (pr + rr) > 0 else 0.0
    return {"gt_static": ns_gt, "gt_dynamic": nd_gt, "est_static": int((~est_dyn_all).sum()), "est_dynamic": int(est_dyn_all.sum()),
            "preserved_static": kept_s, "preserved_dynamic": kept_d, "PR": pr, "RR": rr, "F1": f1}


def evaluate_clouds(gt_xyzi, est_xyzi, voxelsize=0.2):
    gt = np.asarray(gt_xyzi, np.float32).reshape(-1, 4)
    est = np.asarray(est_xyzi, np.float32).reshape(-1, 4)
    return evaluate(gt[:, :3], labels(gt[:, 3]), est[:, :3], labels(est[:, 3]), voxelsize)
