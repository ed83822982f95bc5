"""G9: eval-metric golden values from the reference's own metrics.py (TEST INFRASTRUCTURE; build container only).
The median scaling of test.py:161-162 is applied with the same torch ops (test.py itself parses argv and loads datasets
at import time, so it cannot be imported)."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OMNI_REFERENCE_ROOT", "/root/reference")


def main():
    spec = importlib.util.spec_from_file_location("ref_metrics", os.path.join(REF, "metrics.py"))
    M = importlib.util.module_from_spec(spec); spec.loader.exec_module(M)
    rng = np.random.default_rng(9)
    gt = rng.uniform(0.1, 8.0, (2, 1, 64, 128)).astype(np.float32)
    pred = (gt * rng.uniform(0.7, 1.4, gt.shape) * 1.7).astype(np.float32)
    pred[0, 0, :2] = 0.0
    mask = (rng.random(gt.shape) < 0.8).astype(np.float32)
    p, g, m = torch.from_numpy(pred.copy()), torch.from_numpy(gt), torch.from_numpy(mask)
    scale = g[m > 0].median() / p[m > 0].median()              # test.py:161
    p *= scale                                                  # test.py:162
    vals = [M.abs_rel_error(p, g, m), M.sq_rel_error(p, g, m), M.lin_rms_sq_error(p, g, m), M.log_rms_sq_error(p, g, m),
            M.delta_inlier_ratio(p, g, m, 1), M.delta_inlier_ratio(p, g, m, 2), M.delta_inlier_ratio(p, g, m, 3)]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "G9_eval_metrics.npz"), pred=pred, gt=gt, mask=mask,
                        scaled=p.numpy(), metrics=np.array([float(v) for v in vals], np.float64), N=np.int64(m.sum().item()))
    print([float(v) for v in vals])


if __name__ == "__main__":
    main()
