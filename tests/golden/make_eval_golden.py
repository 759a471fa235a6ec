"""Generates tests/golden/eval_golden.npz by running the REFERENCE's own evaluator
(/root/reference/scripts/analysis_runner.py: labels(), evaluate()) on seeded clouds.  Only runs where the reference
checkout exists (this container); the fixture travels, the reference does not."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/scripts")
import analysis_runner as ar  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
rng = np.random.default_rng(7)
for case in range(4):
    n_gt, n_est = 6000 + 1000 * case, 5000 + 700 * case
    gt = rng.uniform(-20, 20, (n_gt, 3)).astype(np.float32)
    gt[:, 2] *= 0.1
    gt_lab = rng.choice([40, 50, 70, 252, 254, 259, 251, 260], n_gt).astype(np.uint32) | (rng.integers(0, 200, n_gt).astype(np.uint32) << 16)
    keep = rng.uniform(size=n_gt) < 0.8
    est = gt[keep][:n_est] + rng.normal(0, 0.03 + 0.03 * case, (min(n_est, int(keep.sum())), 3)).astype(np.float32)
    est_lab = gt_lab[keep][:n_est].copy()
    flip = rng.uniform(size=len(est_lab)) < 0.05
    est_lab[flip] = 40
    gi, ei = gt_lab.astype(np.float32), est_lab.astype(np.float32)   # intensity carries the label numerically
    r = ar.evaluate(gt, ar.labels(gi), est, ar.labels(ei), voxelsize=0.2)
    out["gt%d" % case] = np.concatenate([gt, gi[:, None]], 1)
    out["est%d" % case] = np.concatenate([est, ei[:, None]], 1)
    out["res%d" % case] = np.array([r[k] for k in ("gt_static", "gt_dynamic", "est_static", "est_dynamic", "preserved_static", "preserved_dynamic", "PR", "RR", "F1")], np.float64)
    print(case, r)
np.savez_compressed(os.path.join(HERE, "eval_golden.npz"), **out)
