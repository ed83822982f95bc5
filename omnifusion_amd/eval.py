"""On-device evaluation metrics — host-side mirror of /root/reference/test.py:121-186 and metrics.py:7-26.

    meters = DepthMetrics()
    meters.update(equi_outputs, depth, mask)        # = compute_eval_metrics(...) of test.py:151: median-scales
                                                    #   `equi_outputs` IN PLACE (test.py:162) and accumulates
    print(meters.averages())                        # {'abs_rel': ..., 'sq_rel': ..., 'rms_sq_lin': ..., 'rms_sq_log': ...,
                                                    #  'd1': ..., 'd2': ..., 'd3': ...}   (AverageMeter.avg, test.py:121-149)

Everything numeric (masked medians by radix select, the seven masked reductions) runs in libomnifusion_hip.so
(csrc/omni_eval.hip); nothing is copied to the host until `averages()` is called.
"""
import ctypes

import torch

from . import _lib

NAMES = ("abs_rel", "sq_rel", "rms_sq_lin", "rms_sq_log", "d1", "d2", "d3")


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def masked_median(x, mask):
    """x[mask > 0].median() as a 1-element device tensor (torch.median semantics: lower middle element)."""
    lib = _lib.load()
    x = x.contiguous(); mask = mask.contiguous().to(torch.float32)
    ws = torch.empty(260, dtype=torch.int32, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.omni_masked_median_f32(_p(x), _p(mask), ctypes.c_size_t(x.numel()), _p(ws), _p(out), _lib.stream_of(x)), "masked_median")
    return out


def compute_eval_metrics(output, gt, depth_mask):
    """test.py:151-176.  Scales `output` in place by median(gt)/median(output) over the mask and returns a device tensor
    [abs_rel, sq_rel, rms_sq_lin, rms_sq_log, d1, d2, d3, N, N_log]."""
    for t, name in ((output, "output"), (gt, "gt"), (depth_mask, "mask")):
        if not t.is_cuda:
            raise ValueError(f"{name} must live on an MI355X device; there is no CPU path")
    if not output.is_contiguous() or output.dtype != torch.float32:
        raise ValueError("output must be a contiguous float32 tensor (it is scaled in place, like test.py:162)")
    if output.shape != gt.shape or output.shape != depth_mask.shape:
        raise ValueError("output, gt and mask must have the same shape")
    lib = _lib.load()
    gt = gt.contiguous().to(torch.float32); mask = depth_mask.contiguous().to(torch.float32)
    num = masked_median(gt, mask)                       # gt_depth[depth_mask>0].median()
    den = masked_median(output, mask)                   # depth_pred[depth_mask>0].median()
    ws = torch.empty(2048 * 9, dtype=torch.float64, device=output.device)
    out = torch.empty(9, dtype=torch.float32, device=output.device)
    with torch.cuda.device(output.device):
        _lib.check(lib.omni_depth_metrics_f32(_p(output), _p(gt), _p(mask), _p(num), _p(den), ctypes.c_size_t(output.numel()),
                                              _p(ws), _p(out), _lib.stream_of(output)), "depth_metrics")
    return out


class DepthMetrics:
    """The seven AverageMeters of test.py:178-186, kept on the device (sum of val*N and of N, test.py:133-137)."""

    def __init__(self):
        self.sum = None
        self.count = None

    def update(self, output, gt, depth_mask):
        m = compute_eval_metrics(output, gt, depth_mask)
        n = m[7]
        if self.sum is None:
            self.sum = torch.zeros(7, dtype=torch.float64, device=m.device); self.count = torch.zeros((), dtype=torch.float64, device=m.device)
        self.sum += m[:7].double() * n.double()
        self.count += n.double()
        return m

    def averages_all_ranks(self):
        """averages over the meters of every rank of an image-sharded run (omnifusion_amd/dist.py): sums of val*N and of N are
        all-reduced once, at the end — not on the data path"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if self.sum is None:
                # a rank that evaluated nothing (fewer images than ranks) still enters the collective, with zeros — raising here would leave
                # the other ranks waiting in all_reduce until the RCCL timeout
                dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() and dist.get_backend() == "nccl" else torch.device("cpu")
                self.sum = torch.zeros(7, dtype=torch.float64, device=dev); self.count = torch.zeros((), dtype=torch.float64, device=dev)
            t = torch.cat([self.sum, self.count.reshape(1)])
            if dist.get_backend() != "nccl":
                t = t.cpu()
            dist.all_reduce(t)
            t = t.to(self.sum.device)
            return dict(zip(NAMES, (t[:7] / t[7]).cpu().tolist()))
        if self.sum is None:
            raise RuntimeError("no batch has been evaluated")
        return self.averages()

    def averages(self):
        if self.sum is None:
            return {k: float("nan") for k in NAMES}
        avg = (self.sum / self.count).cpu().tolist()
        return dict(zip(NAMES, avg))
