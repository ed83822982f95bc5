"""GPU: on-device eval metrics (first "next" row) against the numpy restatement of test.py:151-176 /
metrics.py:7-26, and the masked median against torch.median semantics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_masked_median_matches_torch():
    from omnifusion_amd.eval import masked_median
    g = torch.Generator().manual_seed(3)
    for n, frac in ((1, 1.0), (2, 1.0), (1001, 0.5), (524288, 0.7), (4096, 0.0)):
        x = torch.randn(n, generator=g) * 5
        mask = (torch.rand(n, generator=g) < frac).float()
        got = masked_median(x.to(DEV), mask.to(DEV)).cpu()
        sel = x[mask > 0]
        if sel.numel() == 0:
            assert torch.isnan(got).all()
        else:
            assert got.item() == sel.median().item(), (n, frac)
    # duplicates and an even count: the LOWER middle element
    x = torch.tensor([3.0, 1.0, 2.0, 2.0, 5.0, 4.0]); m = torch.ones(6)
    assert masked_median(x.to(DEV), m.to(DEV)).item() == x.median().item() == 2.0


def test_depth_metrics_match_reference_formulas():
    from omnifusion_amd.eval import compute_eval_metrics, DepthMetrics, NAMES
    from oracle.metrics_ref import compute_eval_metrics as ref_metrics
    rng = np.random.default_rng(9)
    meters = DepthMetrics()
    tot = np.zeros(7); cnt = 0
    for b in range(3):
        gt = rng.uniform(0.1, 8.0, (2, 1, 64, 128)).astype(np.float32)
        pred = (gt * rng.uniform(0.7, 1.4, gt.shape) * 1.7).astype(np.float32)
        pred[0, 0, :2] = 0.0                                    # zero predictions: excluded from the log metric only
        mask = (rng.random(gt.shape) < 0.8).astype(np.float32)
        p_dev = torch.from_numpy(pred).to(DEV)
        m = meters.update(p_dev, torch.from_numpy(gt).to(DEV), torch.from_numpy(mask).to(DEV)).cpu().numpy()
        scaled, ref, N = ref_metrics(pred, gt, mask)
        np.testing.assert_allclose(p_dev.cpu().numpy(), scaled, rtol=1e-6)            # in-place median scaling (test.py:162)
        np.testing.assert_allclose(m[:7], ref, rtol=2e-5)
        assert int(m[7]) == N
        tot += np.array(ref) * N; cnt += N
    avg = meters.averages()
    np.testing.assert_allclose([avg[k] for k in NAMES], tot / cnt, rtol=2e-5)
    with pytest.raises(ValueError):
        compute_eval_metrics(torch.zeros(4), torch.zeros(4), torch.zeros(4))              # CPU tensors


def test_depth_metrics_golden():
    """G9: the HIP metrics against values computed by the reference's own metrics.py."""
    from omnifusion_amd.eval import compute_eval_metrics
    from _util import golden
    g = golden("G9_eval_metrics")
    p = torch.from_numpy(g["pred"].copy()).to(DEV)
    m = compute_eval_metrics(p, torch.from_numpy(g["gt"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(p.cpu().numpy(), g["scaled"], rtol=1e-6)
    np.testing.assert_allclose(m[:7], g["metrics"], rtol=2e-5)
    assert int(m[7]) == int(g["N"])
