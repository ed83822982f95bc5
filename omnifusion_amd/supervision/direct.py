"""calculate_berhu_loss — host-side mirror of /root/reference/supervision/direct.py:3-18 (used at train_erp_depth.py:267).

    loss = calculate_berhu_loss(pred, gt, mask, weights)      # scalar tensor on pred.device, differentiable w.r.t. pred

Same name, arguments and value as the reference.  Everything numeric runs in libomnifusion_hip.so (csrc/omni_io.hip): a max
pass, a deterministic two-stage masked sum and — for backward — one element-wise gradient pass.  Unlike the reference
(`torch.max(abs_diff).item()`, a device->host synchronisation per step) the threshold c = max|gt - pred| / 5 stays on the device;
like there it is a constant of the backward pass.
"""
import ctypes

import torch

from .. import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _BerHu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask, weights):
        lib = _lib.load()
        B = pred.shape[0]
        per = pred.numel() // B
        ws = torch.empty(lib.omni_berhu_workspace_bytes(B) // 4 + 1, dtype=torch.int32, device=pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            _lib.check(lib.omni_berhu_loss_f32(_p(pred), _p(gt), _p(mask), _p(weights), B, ctypes.c_size_t(per), _p(ws), _p(loss),
                                               _lib.stream_of(pred)), "berhu_loss")
        ctx.save_for_backward(pred, gt, mask, weights, ws)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        pred, gt, mask, weights, ws = ctx.saved_tensors
        lib = _lib.load()
        B = pred.shape[0]
        per = pred.numel() // B
        g = grad_out.contiguous().to(torch.float32)
        grad = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            _lib.check(lib.omni_berhu_grad_f32(_p(pred), _p(gt), _p(mask), _p(weights), B, ctypes.c_size_t(per), _p(ws), _p(g), _p(grad),
                                               _lib.stream_of(pred)), "berhu_grad")
        return grad, None, None, None


def calculate_berhu_loss(pred, gt, mask, weights):
    for t, name in ((pred, "pred"), (gt, "gt"), (mask, "mask"), (weights, "weights")):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise ValueError(f"{name} must be a tensor on an MI355X device; there is no CPU path")
    if pred.shape != gt.shape or pred.numel() != mask.numel() or pred.numel() != weights.numel():
        raise ValueError("pred, gt, mask and weights must have the same number of elements")
    if pred.shape[0] < 1 or pred.numel() == 0:
        raise ValueError("empty batch")
    f = lambda t: t.contiguous().to(torch.float32)
    lib = _lib.load()
    lib.omni_berhu_workspace_bytes.restype = ctypes.c_size_t
    return _BerHu.apply(f(pred), f(gt).detach(), f(mask).detach(), f(weights).detach())
