"""CPU restatement of the reference's eval metrics — TEST INFRASTRUCTURE ONLY.
/root/reference/metrics.py:7-26 and /root/reference/test.py:151-176 (numpy, float32 like the reference)."""
import numpy as np


def compute_eval_metrics(pred, gt, mask):
    """returns (scaled_pred, [abs_rel, sq_rel, rms_sq_lin, rms_sq_log, d1, d2, d3], N)"""
    pred = pred.astype(np.float32).copy(); gt = gt.astype(np.float32); m = mask > 0
    def lower_median(v):                                   # torch.median: lower of the two middle elements
        s = np.sort(v.ravel())
        return s[(s.size - 1) // 2]
    scale = np.float32(lower_median(gt[m])) / np.float32(lower_median(pred[m]))          # test.py:161
    pred *= scale                                                                          # test.py:162
    p, g = pred[m].astype(np.float64), gt[m].astype(np.float64)
    out = [np.mean(np.abs(p - g) / g), np.mean((p - g) ** 2 / g), np.mean((p - g) ** 2)]   # metrics.py:7-17
    ml = m & (pred > 1e-7) & (gt > 1e-7)                                                   # metrics.py:21
    out.append(np.mean((np.log(pred[ml].astype(np.float64)) - np.log(gt[ml].astype(np.float64))) ** 2))
    r = np.maximum(p / g, g / p)
    out += [np.mean(r < 1.25 ** d) for d in (1, 2, 3)]                                     # metrics.py:24-26
    return pred, out, int(m.sum())
