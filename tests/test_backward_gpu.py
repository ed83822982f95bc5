"""GPU: the HIP backward of equi2pers / pers2equi (omni_equi2pers_bwd, omni_pers2equi_bwd, reached through autograd on the
drop-in functions) against (i) gradients of the REFERENCE functions (goldens G10/G11), (ii) the C oracle on other sizes and
layouts, (iii) the adjoint identity <J x, y> = <x, J^T y> at BASELINE cfg 1 size.  fp32 atomics: the summation order is not
deterministic, tolerances as in tests/test_backward_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from _util import golden, rng_uniform, assert_close_outliers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers, equi2pers_patches
    from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
    from omnifusion_amd import _lib
    return equi2pers, equi2pers_patches, pers2equi, _lib


@pytest.mark.parametrize("name", ["G10_e2p_bwd", "G10b_e2p_bwd_n6"])
def test_equi2pers_backward_golden(name):
    equi2pers, _, _, _ = _ops()
    g = golden(name)
    H, W, nrows, P, B, C = (int(v) for v in g["meta"])
    erp = torch.rand((B, C, H, W), device=DEV, requires_grad=True)
    pers = equi2pers(erp, (80, 80), nrows, (P, P))[0]
    assert pers.requires_grad
    (pers * torch.from_numpy(g["grad_pers"]).to(DEV)).sum().backward()
    assert_close_outliers(erp.grad.cpu().numpy(), g["grad_erp"], tol=2e-4, max_tol=2e-3, frac=1e-4, what=name)


@pytest.mark.parametrize("name", ["G11_p2e_bwd", "G11b_p2e_bwd_n6"])
def test_pers2equi_backward_golden(name):
    _, _, pers2equi, _ = _ops()
    g = golden(name)
    H, W, nrows, P, B, C = (int(v) for v in g["meta"])
    N = g["grad_pers"].shape[-1]
    pers = torch.rand((B, C, P, P, N), device=DEV, requires_grad=True)
    erp = pers2equi(pers, (80, 80), nrows, (P, P), (H, W), "bwd")
    (erp * torch.from_numpy(g["grad_erp"]).to(DEV)).sum().backward()
    assert_close_outliers(pers.grad.cpu().numpy(), g["grad_pers"], tol=2e-4, max_tol=2e-3, frac=1e-4, what=name)


def test_backward_vs_oracle_planar_layout_and_rect():
    """other sizes (rectangular patches, nrows 3 and 5) and the planar [B,N,C,h,w] layout the model uses"""
    _, equi2pers_patches, pers2equi, L = _ops()
    for nrows, (ph, pw), (H, W) in ((3, (9, 14), (40, 96)), (5, (16, 16), (64, 128))):
        N = {3: 10, 4: 18, 5: 26, 6: 46}[nrows]
        gy = rng_uniform(21, (2, 2, ph, pw, N))
        want = co.equi2pers_bwd(gy, (80, 80), nrows, (H, W))
        for layout in (L.LAYOUT_BCHWN, L.LAYOUT_BNCHW):
            erp = torch.rand((2, 2, H, W), device=DEV, requires_grad=True)
            out = equi2pers_patches(erp, (80, 80), nrows, (ph, pw), layout=layout)
            gyt = torch.from_numpy(gy).to(DEV)
            if layout == L.LAYOUT_BNCHW:
                gyt = gyt.permute(0, 4, 1, 2, 3).contiguous()
            (out * gyt).sum().backward()
            assert_close_outliers(erp.grad.cpu().numpy(), want, tol=2e-4, max_tol=2e-3, frac=1e-4, what=f"e2p bwd {nrows} {layout}")
        ge = rng_uniform(22, (2, 2, H, W))
        want = co.pers2equi_bwd(ge, (80, 80), nrows, (ph, pw))
        for layout in (L.LAYOUT_BCHWN, L.LAYOUT_BNCHW):
            shape = (2, 2, ph, pw, N) if layout == L.LAYOUT_BCHWN else (2, N, 2, ph, pw)
            pers = torch.rand(shape, device=DEV, requires_grad=True)
            e = pers2equi(pers, (80, 80), nrows, (ph, pw), (H, W), None, layout=layout)
            (e * torch.from_numpy(ge).to(DEV)).sum().backward()
            got = pers.grad if layout == L.LAYOUT_BCHWN else pers.grad.permute(0, 2, 3, 4, 1)
            assert_close_outliers(got.cpu().numpy(), want, tol=2e-4, max_tol=2e-3, frac=1e-4, what=f"p2e bwd {nrows} {layout}")


def test_adjoint_identity_config1_size():
    """<J x, y> == <x, J^T y> for both operators at 512x1024 / 18 x 256^2 (size-independent property of a linear map)"""
    equi2pers, _, pers2equi, _ = _ops()
    x = torch.rand((1, 3, 512, 1024), device=DEV, requires_grad=True)
    fwd = equi2pers(x, (80, 80), 4, (256, 256))[0]
    y = torch.rand_like(fwd)
    (fwd * y).sum().backward()
    lhs = float((fwd.detach().double() * y.double()).sum()); rhs = float((x.detach().double() * x.grad.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * abs(lhs)
    p = torch.rand((1, 1, 256, 256, 18), device=DEV, requires_grad=True)
    e = pers2equi(p, (80, 80), 4, (256, 256), (512, 1024), "adj")
    z = torch.rand_like(e)
    (e * z).sum().backward()
    lhs = float((e.detach().double() * z.double()).sum()); rhs = float((p.detach().double() * p.grad.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * abs(lhs)


def test_pers2equi_backward_gather_equals_scatter():
    """the sparse-matrix gather (default, mode 0) and the patch-tile gather kernel (mode 2) against the round-1 scatter kernel (global atomics), at the benchmark size,
    at nrows = 6 (boxes that wrap around the +-pi seam and polar tiles that see a whole ERP row) and on ragged patch tiles"""
    _, _, _, L = _ops()
    import ctypes
    lib = L.load()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    for B, C, nrows, (ph, pw), (H, W) in ((8, 1, 4, (256, 256), (512, 1024)), (2, 2, 6, (128, 128), (512, 1024)), (1, 3, 3, (37, 50), (120, 250)),
                                          (2, 1, 5, (64, 64), (256, 512))):
        N = {3: 10, 4: 18, 5: 26, 6: 46}[nrows]
        ge = torch.rand((B, C, H, W), device=DEV)
        outs = []
        try:
            for simple, wide in ((0, 1), (1, 1), (2, 1), (0, 0)):       # (wide: through the plane-interleaved copy of the gradient | 4-byte gathers)
                L.set_option("p2e_bwd_simple", simple); L.set_option("bwd_wide", wide)
                gp = torch.full((B, N, C, ph, pw), float("nan"), device=DEV)
                rc = lib.omni_pers2equi_bwd(P_(ge), P_(gp), 0, B, C, ph, pw, H, W, nrows, ctypes.c_float(80), ctypes.c_float(80), L.LAYOUT_BNCHW, None)
                assert rc == 0, lib.omni_last_error()
                outs.append(gp)
        finally:
            L.set_option("p2e_bwd_simple", 0); L.set_option("bwd_wide", 1)
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(o).all()) for o in outs)
        assert torch.equal(outs[0], outs[3]), "the interleaved copy changes no bit"
        for k in (0, 2):
            d = (outs[k] - outs[1]).abs().max().item()
            assert d <= 1e-5 * max(1.0, outs[1].abs().max().item()), (nrows, ph, pw, k, d)


def test_equi2pers_backward_gather_equals_scatter():
    """the sparse-matrix gather (default, mode 4), the round-1 LDS-box kernel (global atomics) and the ERP-tile gather kernel against the plain scatter kernel,
    both layouts, the benchmark size, nrows = 6, ragged ERP tiles and odd x odd patches (NaN centre sample, quirk q4)"""
    _, _, _, L = _ops()
    import ctypes
    lib = L.load()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    for B, C, nrows, (ph, pw), (H, W) in ((8, 3, 4, (256, 256), (512, 1024)), (1, 1, 6, (64, 64), (512, 1024)), (2, 2, 3, (9, 13), (50, 99)),
                                          (2, 1, 5, (32, 32), (130, 260))):
        N = {3: 10, 4: 18, 5: 26, 6: 46}[nrows]
        for layout in (L.LAYOUT_BNCHW, L.LAYOUT_BCHWN):
            gp = torch.rand((B, N, C, ph, pw) if layout == L.LAYOUT_BNCHW else (B, C, ph, pw, N), device=DEV)
            outs = []
            try:
                for mode, wide in ((4, 1), (1, 1), (2, 1), (3, 1), (4, 0)):
                    L.set_option("e2p_bwd_simple", mode); L.set_option("bwd_wide", wide)
                    ge = torch.full((B, C, H, W), float("nan"), device=DEV)
                    rc = lib.omni_equi2pers_bwd(P_(gp), P_(ge), 0, B, C, H, W, ph, pw, nrows, ctypes.c_float(80), ctypes.c_float(80), layout, None)
                    assert rc == 0, lib.omni_last_error()
                    outs.append(ge)
            finally:
                L.set_option("e2p_bwd_simple", 0); L.set_option("bwd_wide", 1)
            torch.cuda.synchronize()
            assert torch.equal(outs[0], outs[4]), "the interleaved copy changes no bit"
            assert bool(torch.isfinite(outs[0]).all())
            scale = max(1.0, outs[1].abs().max().item())
            for k in (0, 2, 3):
                assert bool(torch.isfinite(outs[k]).all())
                d = (outs[k] - outs[1]).abs().max().item()
                assert d <= 2e-5 * scale, (nrows, ph, pw, layout, k, d)


def test_backward_is_deterministic():
    """the sparse-matrix gathers have no atomics and a summation order that is a constant of the geometry handle: two calls, same bits"""
    _, _, _, L = _ops()
    import ctypes
    lib = L.load()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    B, C, nrows, P, H, W, N = 2, 3, 4, 128, 256, 512, 18
    gp = torch.rand((B, N, C, P, P), device=DEV)
    ge = torch.rand((B, C, H, W), device=DEV)
    outs = []
    for _ in range(2):
        e = torch.empty((B, C, H, W), device=DEV); p = torch.empty((B, N, C, P, P), device=DEV)
        assert lib.omni_equi2pers_bwd(P_(gp), P_(e), 0, B, C, H, W, P, P, nrows, ctypes.c_float(80), ctypes.c_float(80), L.LAYOUT_BNCHW, None) == 0
        assert lib.omni_pers2equi_bwd(P_(ge), P_(p), 0, B, C, P, P, H, W, nrows, ctypes.c_float(80), ctypes.c_float(80), L.LAYOUT_BNCHW, None) == 0
        outs.append((e, p))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_backward_under_capture_keeps_its_scratch():
    """the backward gathers read a plane-interleaved copy of the gradient held in a per-(geometry, stream) scratch buffer: a batch size first
    seen while the stream is being captured is refused (the buffer would have to be allocated), and a larger eager batch later on the same
    stream does not free the buffer a captured graph replays from"""
    _, _, _, L = _ops()
    import ctypes
    lib = L.load()
    lib.omni_geometry_cache_clear()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    nrows, P, H, W, N = 4, 32, 64, 128, 18
    f80 = ctypes.c_float(80)
    s = torch.cuda.Stream()
    sp = ctypes.c_void_p(s.cuda_stream)
    def both(B, ge, gp, oe, op, stream):
        rc1 = lib.omni_equi2pers_bwd(P_(gp), P_(oe), 0, B, 1, H, W, P, P, nrows, f80, f80, L.LAYOUT_BNCHW, stream)
        rc2 = lib.omni_pers2equi_bwd(P_(ge), P_(op), 0, B, 1, P, P, H, W, nrows, f80, f80, L.LAYOUT_BNCHW, stream)
        return rc1, rc2
    mk = lambda B: (torch.rand((B, 1, H, W), device=DEV), torch.rand((B, N, 1, P, P), device=DEV),
                    torch.zeros((B, 1, H, W), device=DEV), torch.zeros((B, N, 1, P, P), device=DEV))
    t2 = mk(2)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        assert both(2, *t2, sp) == (0, 0)                              # warms tables and the stream's scratch
    s.synchronize()
    want = (t2[2].clone(), t2[3].clone())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        assert both(2, *t2, sp) == (0, 0)
    t8 = mk(8)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        rcs = both(8, *t8, sp)
    # (equi2pers^T needs 4x its warm-up scratch; pers2equi^T at this size still fits the buffer the stream already owns: both use the same one)
    assert rcs[0] != 0 and rcs[1] == 0 and b"captured" in lib.omni_last_error()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        assert both(8, *t8, sp) == (0, 0)                              # a larger scratch for the same stream; the graph's stays
    s.synchronize()
    t2[2].zero_(); t2[3].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(t2[2], want[0]) and torch.equal(t2[3], want[1])
    lib.omni_geometry_cache_clear()


def test_backward_every_plane_count():
    """the gather kernels are instantiated per plane-group size (4 / 8 / 12 / 16 / 24 planes per pass, the gradient's planes padded to a multiple
    of 4): every B * C from 1 to 26, both layouts, against the plain scatter kernels; the 16-byte and the 4-byte forms agree bit for bit"""
    _, _, _, L = _ops()
    import ctypes
    lib = L.load()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    nrows, P, H, W, N = 4, 16, 40, 80, 18
    f80 = ctypes.c_float(80)
    try:
        for planes in range(1, 27):
            B, C = (planes // 3, 3) if planes % 3 == 0 else (planes, 1)
            for layout in (L.LAYOUT_BNCHW, L.LAYOUT_BCHWN):
                gp = torch.rand((B, N, C, P, P) if layout == L.LAYOUT_BNCHW else (B, C, P, P, N), device=DEV)
                ge = torch.rand((B, C, H, W), device=DEV)
                res = []
                for wide, e2p_mode, p2e_mode in ((1, 0, 0), (0, 0, 0), (1, 1, 1)):
                    L.set_option("bwd_wide", wide); L.set_option("e2p_bwd_simple", e2p_mode); L.set_option("p2e_bwd_simple", p2e_mode)
                    oe = torch.full((B, C, H, W), float("nan"), device=DEV); op = torch.full_like(gp, float("nan"))
                    assert lib.omni_equi2pers_bwd(P_(gp), P_(oe), 0, B, C, H, W, P, P, nrows, f80, f80, layout, None) == 0, lib.omni_last_error()
                    assert lib.omni_pers2equi_bwd(P_(ge), P_(op), 0, B, C, P, P, H, W, nrows, f80, f80, layout, None) == 0, lib.omni_last_error()
                    res.append((oe, op))
                torch.cuda.synchronize()
                assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (planes, layout)
                for k in (0, 1):
                    assert bool(torch.isfinite(res[0][k]).all())
                    d = (res[0][k] - res[2][k]).abs().max().item()
                    assert d <= 2e-5 * max(1.0, res[2][k].abs().max().item()), (planes, layout, k, d)
    finally:
        L.set_option("bwd_wide", 1); L.set_option("e2p_bwd_simple", 0); L.set_option("p2e_bwd_simple", 0)


def test_backward_keeps_non_finite_gradients_local():
    """a NaN / Inf in the incoming gradient reaches exactly the outputs whose entries read it: the padding slots of the sparse tables (which
    read element 0) and the rows without entries must not spread it — compared with the plain scatter kernels"""
    _, _, _, L = _ops()
    import ctypes
    lib = L.load()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    B, C, nrows, P, H, W, N = 1, 1, 4, 32, 64, 128, 18
    f80 = ctypes.c_float(80)
    gp = torch.rand((B, N, C, P, P), device=DEV); ge = torch.rand((B, C, H, W), device=DEV)
    gp.view(-1)[0] = float("nan"); gp[0, 3, 0, 10, 11] = float("inf")       # element 0 is what an empty slot would read
    ge.view(-1)[0] = float("nan"); ge[0, 0, 30, 40] = float("inf")
    res = []
    try:
        for e2p_mode, p2e_mode, wide in ((0, 0, 1), (0, 0, 0), (1, 1, 1)):
            L.set_option("e2p_bwd_simple", e2p_mode); L.set_option("p2e_bwd_simple", p2e_mode); L.set_option("bwd_wide", wide)
            oe = torch.zeros((B, C, H, W), device=DEV); op = torch.zeros_like(gp)
            assert lib.omni_equi2pers_bwd(P_(gp), P_(oe), 0, B, C, H, W, P, P, nrows, f80, f80, L.LAYOUT_BNCHW, None) == 0
            assert lib.omni_pers2equi_bwd(P_(ge), P_(op), 0, B, C, P, P, H, W, nrows, f80, f80, L.LAYOUT_BNCHW, None) == 0
            res.append((oe, op))
    finally:
        L.set_option("e2p_bwd_simple", 0); L.set_option("p2e_bwd_simple", 0); L.set_option("bwd_wide", 1)
    torch.cuda.synchronize()
    for k in (0, 1):
        want = ~torch.isfinite(res[2][k])
        assert 0 < int(want.sum()) < want.numel() // 4
        for j in (0, 1):
            assert torch.equal(~torch.isfinite(res[j][k]), want), (k, j, int((~torch.isfinite(res[j][k])).sum()), int(want.sum()))


def test_adjoint_identity_random_geometries():
    """<J x, y> == <x, J^T y> for both operators on a dozen random geometries (odd sizes, every nrows preset, rectangular patches, ERP sizes that
    are no multiple of a slice): a size-independent property of the transposes as they are tabulated"""
    equi2pers, equi2pers_patches, pers2equi, L = _ops()
    rs = np.random.RandomState(7)
    for _ in range(12):
        nrows = int(rs.choice([3, 4, 5, 6]))
        N = {3: 10, 4: 18, 5: 26, 6: 46}[nrows]
        ph, pw = int(rs.randint(5, 41)), int(rs.randint(5, 41))
        H = int(rs.randint(24, 97)); W = int(rs.randint(2 * H - 10, 2 * H + 11))
        B, C = int(rs.randint(1, 4)), int(rs.randint(1, 4))
        layout = L.LAYOUT_BNCHW if rs.rand() < 0.5 else L.LAYOUT_BCHWN
        x = torch.rand((B, C, H, W), device=DEV, requires_grad=True)
        fwd = equi2pers_patches(x, (80, 80), nrows, (ph, pw), layout=layout)
        y = torch.rand_like(fwd)
        (fwd * y).sum().backward()
        lhs = float((fwd.detach().double() * y.double()).sum()); rhs = float((x.detach().double() * x.grad.double()).sum())
        assert abs(lhs - rhs) <= 5e-5 * abs(lhs), ("e2p", nrows, ph, pw, H, W, B, C, layout, lhs, rhs)
        shape = (B, C, ph, pw, N) if layout == L.LAYOUT_BCHWN else (B, N, C, ph, pw)
        p = torch.rand(shape, device=DEV, requires_grad=True)
        e = pers2equi(p, (80, 80), nrows, (ph, pw), (H, W), None, layout=layout)
        z = torch.rand_like(e)
        (e * z).sum().backward()
        lhs = float((e.detach().double() * z.double()).sum()); rhs = float((p.detach().double() * p.grad.double()).sum())
        assert abs(lhs - rhs) <= 5e-5 * abs(lhs), ("p2e", nrows, ph, pw, H, W, B, C, layout, lhs, rhs)


def test_backward_without_tables_falls_back():
    """a geometry whose sparse tables would exceed OMNI_BWD_TABLE_MB keeps the tile kernels of rounds 2-3: same results"""
    _, _, _, L = _ops()
    import ctypes
    lib = L.load()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    B, C, nrows, P, H, W, N = 2, 2, 4, 24, 48, 96, 18
    f80 = ctypes.c_float(80)
    gp = torch.rand((B, N, C, P, P), device=DEV); ge = torch.rand((B, C, H, W), device=DEV)
    res = []
    try:
        for mb in (1024, 0):
            lib.omni_geometry_cache_clear()                            # the budget is read when a geometry builds its tables
            L.set_option("bwd_table_mb", mb)
            oe = torch.full((B, C, H, W), float("nan"), device=DEV); op = torch.full_like(gp, float("nan"))
            assert lib.omni_equi2pers_bwd(P_(gp), P_(oe), 0, B, C, H, W, P, P, nrows, f80, f80, L.LAYOUT_BNCHW, None) == 0, lib.omni_last_error()
            assert lib.omni_pers2equi_bwd(P_(ge), P_(op), 0, B, C, P, P, H, W, nrows, f80, f80, L.LAYOUT_BNCHW, None) == 0, lib.omni_last_error()
            res.append((oe, op))
    finally:
        L.set_option("bwd_table_mb", 1024)
        lib.omni_geometry_cache_clear()
    torch.cuda.synchronize()
    for k in (0, 1):
        assert bool(torch.isfinite(res[1][k]).all())
        d = (res[0][k] - res[1][k]).abs().max().item()
        assert d <= 2e-5 * max(1.0, res[0][k].abs().max().item()), (k, d)
