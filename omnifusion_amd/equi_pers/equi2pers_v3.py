"""equi2pers — host-side mirror of /root/reference/equi_pers/equi2pers_v3.py:20.

Same name, arguments, return tuple, layouts and dtypes as the reference:

    pers, xyz, uv, center_p = equi2pers(erp_img, fov, nrows, patch_size)

    erp_img   Tensor[B,C,H,W] float32 (float16 also accepted) on an MI355X device
    fov       (fov_h, fov_w) in degrees, or a scalar            (reference :23)
    nrows     3 | 4 | 5 | 6  -> N = 10 | 18 | 26 | 46 patches    (reference :32-47)
    patch_size (h, w) or int                                    (reference :22)
    returns   pers[B,C,h,w,N] (N innermost, contiguous), xyz[N,3,h,w], uv[N,2,h,w] on
              erp_img.device and center_p[N,2] on the CPU      (reference :82,118,122)

All arithmetic runs in libomnifusion_hip.so (csrc/omni_equi2pers.hip) on the caller's
current stream.  Like the reference's (which autograd differentiates through F.grid_sample, :111), `pers` is
differentiable w.r.t. `erp_img` (float32): the backward is the HIP scatter kernel `omni_equi2pers_bwd`.
Errors: ValueError for bad nrows/shape/dtype/device (the reference raises
UnboundLocalError for an unsupported nrows), RuntimeError for HIP failures.
"""
import ctypes

import torch

from .. import _lib

_NPATCH = {3: 10, 4: 18, 5: 26, 6: 46}


def pair(t):
    return tuple(t) if isinstance(t, (tuple, list)) else (t, t)


def _check_input(t, name, ndim):
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name} must be a torch.Tensor")
    if t.dim() != ndim:
        raise ValueError(f"{name} must have {ndim} dimensions, got shape {tuple(t.shape)}")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on an MI355X device (got {t.device}); "
                         "this package has no CPU path")
    if t.requires_grad and torch.is_grad_enabled() and t.dtype != torch.float32:
        raise RuntimeError(f"{name} requires grad: the HIP backward is float32 only")


class _Equi2PersFn(torch.autograd.Function):
    """pers = equi2pers(erp) with the HIP backward (the operator is linear in erp: grad_erp = J^T grad_pers)."""

    @staticmethod
    def forward(ctx, erp_img, fov, nrows, patch_size, layout):
        ctx.cfg = (fov, nrows, patch_size, layout, tuple(erp_img.shape))
        with torch.no_grad():
            return equi2pers_patches(erp_img.detach(), fov, nrows, patch_size, layout)

    @staticmethod
    def backward(ctx, grad_pers):
        fov, nrows, patch_size, layout, (B, C, H, W) = ctx.cfg
        lib = _lib.load()
        ph, pw = (int(v) for v in pair(patch_size))
        fov_h, fov_w = (float(v) for v in pair(fov))
        g = grad_pers.contiguous().float()
        grad_erp = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.omni_equi2pers_bwd(_lib.ptr(g), _lib.ptr(grad_erp), _lib.F32, B, C, H, W, ph, pw, int(nrows),
                                        ctypes.c_float(fov_h), ctypes.c_float(fov_w), int(layout), _lib.stream_of(g))
        _lib.check(rc, "equi2pers backward")
        return grad_erp, None, None, None, None


def equi2pers_patches(erp_img, fov, nrows, patch_size, layout=_lib.LAYOUT_BCHWN):
    """Only the sampled patches, in the reference layout [B,C,h,w,N] (default) or in the
    patch-major planar layout [B,N,C,h,w] the model uses internally."""
    _check_input(erp_img, "erp_img", 4)
    if erp_img.requires_grad and torch.is_grad_enabled():
        return _Equi2PersFn.apply(erp_img, fov, nrows, patch_size, layout)
    lib = _lib.load()
    if nrows not in _NPATCH:
        raise ValueError(f"unsupported nrows={nrows!r}: presets are 3, 4, 5, 6")
    ph, pw = (int(v) for v in pair(patch_size))
    fov_h, fov_w = (float(v) for v in pair(fov))
    erp = erp_img.contiguous()
    B, C, H, W = erp.shape
    N = _NPATCH[nrows]
    shape = (B, C, ph, pw, N) if layout == _lib.LAYOUT_BCHWN else (B, N, C, ph, pw)
    pers = torch.empty(shape, dtype=erp.dtype, device=erp.device)
    with torch.cuda.device(erp.device):
        rc = lib.omni_equi2pers(_lib.ptr(erp), _lib.ptr(pers), _lib.dtype_code(erp), B, C, H, W, ph, pw,
                                int(nrows), ctypes.c_float(fov_h), ctypes.c_float(fov_w), int(layout),
                                _lib.stream_of(erp))
    _lib.check(rc, "equi2pers")
    return pers


def equi2pers_aux(device, fov, nrows, patch_size, want_xyz=True, want_uv=True):
    """xyz[N,3,h,w] / uv[N,2,h,w] (fp32, on `device`) and center_p[N,2] (CPU)."""
    lib = _lib.load()
    if nrows not in _NPATCH:
        raise ValueError(f"unsupported nrows={nrows!r}: presets are 3, 4, 5, 6")
    ph, pw = (int(v) for v in pair(patch_size))
    fov_h, fov_w = (float(v) for v in pair(fov))
    N = _NPATCH[nrows]
    device = torch.device(device)
    xyz = torch.empty((N, 3, ph, pw), dtype=torch.float32, device=device) if want_xyz else None
    uv = torch.empty((N, 2, ph, pw), dtype=torch.float32, device=device) if want_uv else None
    ref = xyz if xyz is not None else uv
    if ref is not None:
        with torch.cuda.device(device):
            rc = lib.omni_equi2pers_aux(_lib.ptr(xyz) if want_xyz else None, _lib.ptr(uv) if want_uv else None,
                                        ph, pw, int(nrows), ctypes.c_float(fov_h), ctypes.c_float(fov_w),
                                        _lib.stream_of(ref))
        _lib.check(rc, "equi2pers_aux")
    cp = (ctypes.c_float * (2 * N))()
    _lib.check(lib.omni_patch_centers(int(nrows), 0, cp), "patch_centers")
    center_p = torch.tensor(list(cp), dtype=torch.float32).reshape(N, 2)
    return xyz, uv, center_p


def equi2pers(erp_img, fov, nrows, patch_size):
    pers = equi2pers_patches(erp_img, fov, nrows, patch_size)
    xyz, uv, center_p = equi2pers_aux(erp_img.device, fov, nrows, patch_size)
    return pers, xyz, uv, center_p
