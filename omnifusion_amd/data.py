"""Host input pipeline on the device side of the PCIe link — replaces the per-pixel CPU work of
/root/reference/dataset_loader_stanford.py:42-109 (decoded frame -> cv2.resize INTER_AREA -> /255 -> CHW float32; 16-bit depth ->
metres -> validity mask) and the `rgb.cuda()` of test.py:196.

    rgb   = preprocess_rgb(frames_u8, (512, 1024))               # [B,Hs,Ws,3] uint8 BGR on the GPU -> [B,3,512,1024] float32
    d, m  = preprocess_depth(frames_u16, (512, 1024))            # [B,Hs,Ws] uint16 -> depth [B,1,H,W] float32 (masked), mask uint8

    feeder = DeviceFeeder(batches, (512, 1024))                  # iterable of host uint8 arrays [B,Hs,Ws,3] (decoded by the loader's workers)
    for rgb in feeder:                                           # rgb: device float32 [B,3,H,W], ready on the CURRENT stream
        depth = net(rgb)

What crosses PCIe is the DECODED uint8 frame (1.5 MB per 512x1024 panorama instead of the reference's 6.3 MB of float32: 4.3 GB/s
instead of 17 GB/s at 2 750 panoramas/s).  `DeviceFeeder` keeps `depth` (default 3) pinned staging buffers and device frame buffers
in rotation: the H2D copy of batch k+1 and its preprocess kernel run on a side stream while batch k is in the network; the consumer
stream only waits on an event.  Decoding PNG/EXR files stays with the loader's CPU workers (out of scope: no image codec is part of
the hot path); everything after the decoded frame is on the device.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _hw(size):
    return (int(size[0]), int(size[1])) if isinstance(size, (tuple, list)) else (int(size), int(size))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def preprocess_rgb(frames_u8, size, out=None):
    if not isinstance(frames_u8, torch.Tensor) or not frames_u8.is_cuda:
        raise ValueError("frames must be a uint8 tensor on an MI355X device; there is no CPU path")
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3:
        raise ValueError("expected decoded frames [B,Hs,Ws,3] uint8 (BGR, as cv2.imread returns them)")
    H, W = _hw(size)
    f = frames_u8.contiguous()
    B, Hs, Ws, _ = f.shape
    if out is None:
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=f.device)
    with torch.cuda.device(f.device):
        _lib.check(_lib.load().omni_preprocess_rgb_u8(_p(f), _p(out), B, Hs, Ws, H, W, _lib.stream_of(f)), "preprocess_rgb")
    return out


def preprocess_depth(frames_u16, size, min_depth=0.1, max_depth=8.0):
    if not isinstance(frames_u16, torch.Tensor) or not frames_u16.is_cuda:
        raise ValueError("frames must be a tensor on an MI355X device; there is no CPU path")
    if frames_u16.dtype not in (torch.uint16, torch.int16) or frames_u16.dim() != 3:
        raise ValueError("expected 16-bit depth frames [B,Hs,Ws] (uint16; int16 storage is reinterpreted)")
    H, W = _hw(size)
    f = frames_u16.contiguous()
    B, Hs, Ws = f.shape
    depth = torch.empty((B, 1, H, W), dtype=torch.float32, device=f.device)
    mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device=f.device)
    with torch.cuda.device(f.device):
        _lib.check(_lib.load().omni_preprocess_depth_u16(_p(f), _p(depth), _p(mask), B, Hs, Ws, H, W, ctypes.c_float(min_depth),
                                                        ctypes.c_float(max_depth), _lib.stream_of(f)), "preprocess_depth")
    return depth, mask


class DeviceFeeder:
    """Pinned, multi-buffered host -> device staging of decoded frames + the preprocess kernel on a copy stream."""

    def __init__(self, batches, size, device=None, depth=3):
        self.batches, self.size, self.depth = batches, _hw(size), max(2, int(depth))
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        self._slots = None

    def _alloc(self, shape):
        B, Hs, Ws, _ = shape
        H, W = self.size
        self._slots = [{"pin": torch.empty(shape, dtype=torch.uint8).pin_memory(),
                        "dev": torch.empty(shape, dtype=torch.uint8, device=self.device),
                        "out": torch.empty((B, 3, H, W), dtype=torch.float32, device=self.device),
                        "ready": torch.cuda.Event(), "free": None} for _ in range(self.depth)]
        self._last = None

    def _stage(self, slot, frames):
        s = self._slots[slot]
        if s["free"] is not None:
            s["free"].synchronize()                                    # the consumer is done with this slot's output
        if isinstance(frames, torch.Tensor) and frames.is_pinned():
            # a DataLoader(pin_memory=True) batch: already page-locked, copied to the device straight from where the worker put it
            if frames.dtype != torch.uint8 or tuple(frames.shape) != tuple(s["pin"].shape):
                raise ValueError("pinned batches must be uint8 [B,Hs,Ws,3] of one shape")
            src = frames
        else:
            a = np.ascontiguousarray(frames.numpy() if isinstance(frames, torch.Tensor) else frames)
            if a.dtype != np.uint8 or a.ndim != 4 or a.shape[3] != 3:
                raise ValueError("batches must yield uint8 arrays [B,Hs,Ws,3]")
            if tuple(s["pin"].shape) != a.shape:
                raise ValueError("all batches of one feeder must have the same shape")
            s["pin"].numpy()[...] = a                                  # host memcpy into pinned memory (the worker's hand-over)
            src = s["pin"]
        with torch.cuda.stream(self.side):
            s["dev"].copy_(src, non_blocking=True)                     # async H2D on the side stream
            preprocess_rgb(s["dev"], self.size, out=s["out"])
            s["ready"].record(self.side)

    def __iter__(self):
        it = iter(self.batches)
        pending = []
        k = 0
        for frames in it:
            if self._slots is None:
                self._alloc(tuple(frames.shape))
            self._stage(k % self.depth, frames)
            pending.append(k % self.depth)
            k += 1
            if len(pending) >= self.depth - 1:                         # keep depth-1 batches in flight ahead of the consumer
                yield self._hand_over(pending.pop(0))
        while pending:
            yield self._hand_over(pending.pop(0))

    def _hand_over(self, slot):
        s = self._slots[slot]
        cur = torch.cuda.current_stream(self.device)
        # the batch handed over LAST time has been consumed by everything enqueued on `cur` up to now: its slot becomes reusable
        # once that work is done (the event is waited for — on the host — only when the slot comes around again, depth-1 batches later)
        prev = getattr(self, "_last", None)
        if prev is not None:
            ev = torch.cuda.Event(); ev.record(cur); self._slots[prev]["free"] = ev
        self._last = slot
        cur.wait_event(s["ready"])                                     # no host synchronisation
        return s["out"]
