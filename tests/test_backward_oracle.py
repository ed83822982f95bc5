"""CPU: pin the C restatement of the two operators' backward (oracle/omni_oracle.c, `*_bwd`) against gradients obtained by
differentiating the REFERENCE functions with autograd (oracle/gen_golden_bwd.py -> tests/golden/G10*, G11*), and check the
adjoint identity <J x, y> = <x, J^T y> against the forward restatement (size-independent property).

Tolerances: the operators are linear maps with non-negative weights summing to <= 1 per output, gradients here reach ~2.4
(equi2pers) / ~24 (pers2equi); |d| <= 2e-3 absolute with at most 1e-4 of the elements above 2e-4 — a sample whose fp32
coordinate lies within round-off of a pixel boundary moves its weight to the neighbouring tap (same caveat as the forward)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from _util import golden, rng_uniform, assert_close_outliers


@pytest.mark.parametrize("name", ["G10_e2p_bwd", "G10b_e2p_bwd_n6"])
def test_equi2pers_bwd_golden(name):
    g = golden(name)
    H, W, nrows, P, B, C = (int(v) for v in g["meta"])
    got = co.equi2pers_bwd(g["grad_pers"], (80, 80), nrows, (H, W))
    assert got.shape == g["grad_erp"].shape
    assert_close_outliers(got, g["grad_erp"], tol=2e-4, max_tol=2e-3, frac=1e-4, what=name)


@pytest.mark.parametrize("name", ["G11_p2e_bwd", "G11b_p2e_bwd_n6"])
def test_pers2equi_bwd_golden(name):
    g = golden(name)
    H, W, nrows, P, B, C = (int(v) for v in g["meta"])
    got = co.pers2equi_bwd(g["grad_erp"], (80, 80), nrows, (P, P))
    assert got.shape == g["grad_pers"].shape
    assert_close_outliers(got, g["grad_pers"], tol=2e-4, max_tol=2e-3, frac=1e-4, what=name)


@pytest.mark.parametrize("nrows,P,H,W", [(4, 16, 64, 128), (3, 9, 32, 64), (5, 12, 48, 96)])
def test_adjoint_identity(nrows, P, H, W):
    x = rng_uniform(5, (1, 2, H, W))
    fwd = co.equi2pers(x, (80, 80), nrows, (P, P))[0]
    y = rng_uniform(6, fwd.shape)
    lhs = float((fwd.astype(np.float64) * y).sum())
    rhs = float((x.astype(np.float64) * co.equi2pers_bwd(y, (80, 80), nrows, (H, W))).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    p = rng_uniform(7, fwd.shape)
    e = co.pers2equi(p, (80, 80), nrows, (P, P), (H, W))
    z = rng_uniform(8, e.shape)
    lhs = float((e.astype(np.float64) * z).sum())
    rhs = float((p.astype(np.float64) * co.pers2equi_bwd(z, (80, 80), nrows, (P, P))).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
