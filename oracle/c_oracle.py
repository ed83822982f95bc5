"""ctypes binding of oracle/libomni_oracle.so (TEST INFRASTRUCTURE ONLY).

Restates /root/reference/equi_pers/equi2pers_v3.py:20 and pers2equi_v3.py:16 on
the CPU; see omni_oracle.c for the line-by-line citations.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libomni_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "omni_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libomni_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.omni_oracle_patch_centers.restype = ctypes.c_int
        _lib.omni_oracle_equi2pers.restype = ctypes.c_int
        _lib.omni_oracle_pers2equi.restype = ctypes.c_int
        _lib.omni_oracle_pers2equi_tables.restype = ctypes.c_int
        _lib.omni_oracle_pers2equi_conf.restype = ctypes.c_int
        _lib.omni_oracle_equi2pers_bwd.restype = ctypes.c_int
        _lib.omni_oracle_pers2equi_bwd.restype = ctypes.c_int
    return _lib


def _pair(t):
    return tuple(t) if isinstance(t, (tuple, list)) else (t, t)


def _fp(a):
    return a.ctypes.data_as(_F) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(_I) if a is not None else None


def patch_centers(nrows, which=0):
    lam = np.zeros(64, np.float32); phi = np.zeros(64, np.float32); cp = np.zeros(128, np.float32)
    n = lib().omni_oracle_patch_centers(int(nrows), int(which), _fp(lam), _fp(phi), _fp(cp))
    if n < 0:
        raise ValueError(f"unsupported nrows={nrows}")
    return lam[:n].copy(), phi[:n].copy(), cp[:2 * n].reshape(n, 2).copy()


def equi2pers(erp, fov, nrows, patch_size, want_pers=True):
    """erp: float32 ndarray [B,C,H,W] -> (pers[B,C,h,w,N], xyz[N,3,h,w], uv[N,2,h,w], center_p[N,2])"""
    erp = np.ascontiguousarray(erp, np.float32)
    B, C, H, W = erp.shape
    ph, pw = _pair(patch_size)
    fh, fw = _pair(fov)
    N = patch_centers(nrows)[0].shape[0]
    pers = np.empty((B, C, ph, pw, N), np.float32) if want_pers else None
    xyz = np.empty((N, 3, ph, pw), np.float32)
    uv = np.empty((N, 2, ph, pw), np.float32)
    cp = np.empty((N, 2), np.float32)
    r = lib().omni_oracle_equi2pers(_fp(erp), B, C, H, W, ctypes.c_float(fh), ctypes.c_float(fw),
                                    int(nrows), ph, pw, _fp(pers), _fp(xyz), _fp(uv), _fp(cp))
    if r < 0:
        raise RuntimeError(f"omni_oracle_equi2pers failed: {r}")
    return pers, xyz, uv, cp


def pers2equi(pers, fov, nrows, patch_size, erp_size, want_cover=False):
    """pers: float32 ndarray [B,C,h,w,N] -> erp [B,C,H,W]"""
    pers = np.ascontiguousarray(pers, np.float32)
    B, C, ph, pw, N = pers.shape
    assert (ph, pw) == _pair(patch_size)
    fh, fw = _pair(fov)
    H, W = _pair(erp_size)
    if patch_centers(nrows, 1)[0].shape[0] != N:
        raise ValueError("patch count does not match nrows")
    out = np.empty((B, C, H, W), np.float32)
    cover = np.empty((H, W), np.int32) if want_cover else None
    r = lib().omni_oracle_pers2equi(_fp(pers), B, C, ph, pw, int(nrows), ctypes.c_float(fh),
                                    ctypes.c_float(fw), H, W, _fp(out), _ip(cover))
    if r < 0:
        raise RuntimeError(f"omni_oracle_pers2equi failed: {r}")
    return (out, cover) if want_cover else out


def pers2equi_tables(fov, nrows, patch_size, erp_size):
    ph, pw = _pair(patch_size); fh, fw = _pair(fov); H, W = _pair(erp_size)
    N = patch_centers(nrows, 1)[0].shape[0]
    x0, y0, x1, y1, mask = (np.empty((N, H, W), np.int32) for _ in range(5))
    wl = np.empty((N, H, W, 4), np.float32)
    r = lib().omni_oracle_pers2equi_tables(ph, pw, int(nrows), ctypes.c_float(fh), ctypes.c_float(fw),
                                           H, W, _ip(x0), _ip(y0), _ip(x1), _ip(y1), _ip(mask), _fp(wl))
    if r < 0:
        raise RuntimeError("omni_oracle_pers2equi_tables failed")
    return dict(x0=x0, y0=y0, x1=x1, y1=y1, mask=mask, w_list=wl)


def pers2equi_conf(pred_w, conf, fov, nrows, patch_size, erp_size):
    pred_w = np.ascontiguousarray(pred_w, np.float32); conf = np.ascontiguousarray(conf, np.float32)
    B, C, ph, pw, N = pred_w.shape
    assert C == 1 and conf.shape == pred_w.shape
    fh, fw = _pair(fov); H, W = _pair(erp_size)
    out = np.empty((B, 1, H, W), np.float32)
    r = lib().omni_oracle_pers2equi_conf(_fp(pred_w), _fp(conf), B, ph, pw, int(nrows),
                                         ctypes.c_float(fh), ctypes.c_float(fw), H, W, _fp(out))
    if r < 0:
        raise RuntimeError("omni_oracle_pers2equi_conf failed")
    return out


def equi2pers_bwd(grad_pers, fov, nrows, erp_size):
    """vector-Jacobian product of equi2pers w.r.t. the ERP image: grad_pers [B,C,h,w,N] -> grad_erp [B,C,H,W]"""
    g = np.ascontiguousarray(grad_pers, np.float32)
    B, C, ph, pw, N = g.shape
    fh, fw = _pair(fov)
    H, W = _pair(erp_size)
    out = np.empty((B, C, H, W), np.float32)
    r = lib().omni_oracle_equi2pers_bwd(_fp(g), B, C, H, W, ctypes.c_float(fh), ctypes.c_float(fw), int(nrows), ph, pw, _fp(out))
    if r != N:
        raise RuntimeError(f"omni_oracle_equi2pers_bwd failed: {r}")
    return out


def pers2equi_bwd(grad_erp, fov, nrows, patch_size):
    """vector-Jacobian product of pers2equi w.r.t. the patches: grad_erp [B,C,H,W] -> grad_pers [B,C,h,w,N]"""
    g = np.ascontiguousarray(grad_erp, np.float32)
    B, C, H, W = g.shape
    ph, pw = _pair(patch_size)
    fh, fw = _pair(fov)
    N = patch_centers(nrows, 1)[0].shape[0]
    out = np.empty((B, C, ph, pw, N), np.float32)
    r = lib().omni_oracle_pers2equi_bwd(_fp(g), B, C, ph, pw, int(nrows), ctypes.c_float(fh), ctypes.c_float(fw), H, W, _fp(out))
    if r != N:
        raise RuntimeError(f"omni_oracle_pers2equi_bwd failed: {r}")
    return out
