"""pers2equi — host-side mirror of /root/reference/equi_pers/pers2equi_v3.py:16.

    erp = pers2equi(pers_img, fov, nrows, patch_size, erp_size, layer_name)

    pers_img   Tensor[B,C,h,w,N] float32 (float16 accepted), N innermost like the reference
    erp_size   (H, W) or int
    layer_name accepted and ignored: in the reference it is only the file name of the
               ./grid/<layer_name>.pth table cache (:24-29); nothing is written to disk here.
    returns    Tensor[B,C,H,W] on pers_img.device

All arithmetic runs in libomnifusion_hip.so (csrc/omni_pers2equi.hip).  Differentiable w.r.t. `pers_img` (float32) like
the reference's indexing gathers (:174-196): the backward is the HIP scatter kernel `omni_pers2equi_bwd`.
"""
import ctypes

import torch

from .. import _lib
from .equi2pers_v3 import pair, _check_input, _NPATCH


def _args(fov, patch_size, erp_size):
    ph, pw = (int(v) for v in pair(patch_size))
    fov_h, fov_w = (float(v) for v in pair(fov))
    H, W = (int(v) for v in pair(erp_size))
    return ph, pw, fov_h, fov_w, H, W


def _patch_dims(t, layout, ph, pw, nrows):
    if nrows not in _NPATCH:
        raise ValueError(f"unsupported nrows={nrows!r}: presets are 3, 4, 5, 6")
    N = _NPATCH[nrows]
    if layout == _lib.LAYOUT_BCHWN:
        B, C, h, w, n = t.shape
    elif layout == _lib.LAYOUT_BNCHW:
        B, n, C, h, w = t.shape
    else:
        B, n, h, w, C = t.shape
    if (h, w) != (ph, pw):
        raise ValueError(f"patch_size {(ph, pw)} does not match the tensor's patch dims {(h, w)}")
    if n != N:
        raise ValueError(f"nrows={nrows} means {N} patches but the tensor holds {n}")
    return B, C


class _Pers2EquiFn(torch.autograd.Function):
    """erp = pers2equi(pers) with the HIP backward (linear in pers: grad_pers = J^T grad_erp)."""

    @staticmethod
    def forward(ctx, pers_img, fov, nrows, patch_size, erp_size, layout):
        ctx.cfg = (fov, nrows, patch_size, erp_size, layout, tuple(pers_img.shape))
        with torch.no_grad():
            return pers2equi(pers_img.detach(), fov, nrows, patch_size, erp_size, None, layout)

    @staticmethod
    def backward(ctx, grad_erp):
        fov, nrows, patch_size, erp_size, layout, shape = ctx.cfg
        lib = _lib.load()
        ph, pw, fov_h, fov_w, H, W = _args(fov, patch_size, erp_size)
        g = grad_erp.contiguous().float()
        B, C = g.shape[0], g.shape[1]
        grad_pers = torch.empty(shape, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.omni_pers2equi_bwd(_lib.ptr(g), _lib.ptr(grad_pers), _lib.F32, B, C, ph, pw, H, W, int(nrows),
                                        ctypes.c_float(fov_h), ctypes.c_float(fov_w), int(layout), _lib.stream_of(g))
        _lib.check(rc, "pers2equi backward")
        return grad_pers, None, None, None, None, None


VIA_PLANAR = True      # reference-layout inputs take conversion + planar kernel (False: the direct N-innermost kernel; same bits, 3x slower)
VIA_PLANAR_MIN = 2 << 20   # ... from 2 M elements up: below, the second launch costs more than the direct kernel's uncoalesced reads
                           # (8 x 18 x 256^2 fp32: 32.7 vs 94.6 us; 8 x 18 x 128^2: 25.8 vs 50.4; ONE 18 x 256^2 panorama: 20.2 vs 12.5)


def pers2equi(pers_img, fov, nrows, patch_size, erp_size, layer_name=None, layout=_lib.LAYOUT_BCHWN):
    _check_input(pers_img, "pers_img", 5)
    if pers_img.requires_grad and torch.is_grad_enabled():
        return _Pers2EquiFn.apply(pers_img, fov, nrows, patch_size, erp_size, layout)
    lib = _lib.load()
    ph, pw, fov_h, fov_w, H, W = _args(fov, patch_size, erp_size)
    B, C = _patch_dims(pers_img, layout, ph, pw, nrows)
    pers = pers_img.contiguous()
    erp = torch.empty((B, C, H, W), dtype=pers.dtype, device=pers.device)
    with torch.cuda.device(pers.device):
        if layout == _lib.LAYOUT_BCHWN and VIA_PLANAR and pers.numel() >= VIA_PLANAR_MIN:   # N-innermost input: one coalesced conversion pass, then the planar kernel
            N = pers.shape[-1]
            planar = torch.empty((B, N, C, ph, pw), dtype=pers.dtype, device=pers.device)
            _lib.check(lib.omni_patches_to_planar(_lib.ptr(pers), _lib.ptr(planar), _lib.dtype_code(pers), B, C, ph, pw, N, _lib.stream_of(pers)),
                       "patches_to_planar")
            pers, layout = planar, _lib.LAYOUT_BNCHW
        rc = lib.omni_pers2equi(_lib.ptr(pers), _lib.ptr(erp), _lib.dtype_code(pers), B, C, ph, pw, H, W,
                                int(nrows), ctypes.c_float(fov_h), ctypes.c_float(fov_w), int(layout),
                                _lib.stream_of(pers))
    _lib.check(rc, "pers2equi")
    return erp


def pers2equi_conf(pred_w, conf, fov, nrows, patch_size, erp_size, layout=_lib.LAYOUT_BCHWN):
    """Fused confidence blend of model/spherical_model.py:307-311:
    pers2equi(pred_w) / (pers2equi(conf) + 1e-8 * [pers2equi(conf) <= 1e-8]), one kernel."""
    _check_input(pred_w, "pred_w", 5)
    _check_input(conf, "conf", 5)
    if pred_w.shape != conf.shape or pred_w.dtype != conf.dtype:
        raise ValueError("pred_w and conf must have the same shape and dtype")
    lib = _lib.load()
    ph, pw, fov_h, fov_w, H, W = _args(fov, patch_size, erp_size)
    B, C = _patch_dims(pred_w, layout, ph, pw, nrows)
    if C != 1:
        raise ValueError("the confidence blend takes single-channel patch tensors")
    a, b = pred_w.contiguous(), conf.contiguous()
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = lib.omni_pers2equi_conf(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.dtype_code(a), B, ph, pw,
                                     H, W, int(nrows), ctypes.c_float(fov_h), ctypes.c_float(fov_w),
                                     int(layout), _lib.stream_of(a))
    _lib.check(rc, "pers2equi_conf")
    return out
