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
    elif (not isinstance(out, torch.Tensor) or out.dtype != torch.float32 or out.device != f.device or tuple(out.shape) != (B, 3, H, W)
          or not out.is_contiguous()):
        raise ValueError(f"out must be a contiguous float32 tensor [{B},3,{H},{W}] on {f.device} (the kernel writes through its raw pointer)")
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
    """Pinned, multi-buffered host -> device staging of decoded frames on a copy stream; /255 + HWC->CHW on the consumer's stream.

    The yielded tensor is a reusable float32 buffer: it is valid until the next batch is requested (stream order on the
    consumer's current stream guarantees that the previous forward has read it before it is overwritten) — clone it to keep it.
    Device footprint: `depth` uint8 frames + one float32 batch (the 256 MB Infinity Cache also holds the network's weights and
    activations: a float32 buffer per slot measurably slows the forward down).

    A consumer that reads the batch on ANOTHER stream (`spherical_fusion.pipelined`) asks for `out_buffers=2` and reports when the
    batch has been read: `feeder.done_with(rgb, pending.input_read)` (the event after equi2pers, the forward's only reader of
    the panoramas) — the buffer is rewritten only after that event."""

    def __init__(self, batches, size, device=None, depth=3, out_buffers=1):
        self.batches, self.size, self.depth = batches, _hw(size), max(2, int(depth))
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        self._slots = None
        self._outs, self._done, self._k, self._nout = None, None, 0, max(1, int(out_buffers))

    def done_with(self, rgb, event):
        """`rgb` (a tensor this feeder yielded) is read by work on another stream that `event` marks the end of"""
        for j, o in enumerate(self._outs):
            if o.data_ptr() == rgb.data_ptr():
                self._done[j] = event
                return
        raise ValueError("not a batch of this feeder")

    def _alloc(self, shape):
        B, Hs, Ws, _ = shape
        H, W = self.size
        self._slots = [{"pin": None, "dev": torch.empty(shape, dtype=torch.uint8, device=self.device),
                        "ready": torch.cuda.Event(), "free": None} for _ in range(self.depth)]
        self._outs = [torch.empty((B, 3, H, W), dtype=torch.float32, device=self.device) for _ in range(self._nout)]
        self._done = [None] * self._nout

    def _stage(self, slot, frames):
        s = self._slots[slot]
        if isinstance(frames, torch.Tensor) and frames.is_pinned():
            # a DataLoader(pin_memory=True) batch: already page-locked, copied to the device straight from where the worker put it
            if frames.dtype != torch.uint8 or frames.dim() != 4 or tuple(frames.shape[1:]) != tuple(s["dev"].shape[1:]) or frames.shape[0] > s["dev"].shape[0]:
                raise ValueError("pinned batches must be uint8 [B' <= B,Hs,Ws,3] of one frame size")
            src = frames
        else:
            a = np.ascontiguousarray(frames.numpy() if isinstance(frames, torch.Tensor) else frames)
            if a.dtype != np.uint8 or a.ndim != 4 or a.shape[3] != 3:
                raise ValueError("batches must yield uint8 arrays [B,Hs,Ws,3]")
            if tuple(s["dev"].shape[1:]) != a.shape[1:] or a.shape[0] > s["dev"].shape[0]:
                raise ValueError("all batches of one feeder must have the same frame size and at most the first batch's length")
            if s["pin"] is None:
                s["pin"] = torch.empty(tuple(s["dev"].shape), dtype=torch.uint8).pin_memory()
            if s["free"] is not None:
                s["free"].synchronize()                                # (the previous H2D out of this pinned buffer has been consumed)
            s["pin"].numpy()[:a.shape[0]] = a                          # host memcpy into pinned memory (the worker's hand-over)
            src = s["pin"][:a.shape[0]]
        if s["free"] is not None:
            # the consumer's preprocess has read this slot's device frame — waited for on the HOST (the event is `depth` batches old: it returns at
            # once): a stream-side wait in front of the asynchronous copy makes the runtime hold the copy back (round 5, tools/feed_ab.py: 3593-3893
            # panoramas/s, jittery, with 4 hardware queues and 2650 with 8, against 3904-3966 this way; the GPU-side cost of feeding is 1 %)
            s["free"].synchronize()
        with torch.cuda.stream(self.side):
            s["dev"][:src.shape[0]].copy_(src, non_blocking=True)      # async H2D on the side stream
            s["ready"].record(self.side)
            if src is frames and hasattr(self.batches, "recycle"):     # a producer with a buffer ring (png.PngBatches): the page-locked
                ev = torch.cuda.Event(); ev.record(self.side)          # batch may be decoded into again once this copy has run
                self.batches.recycle(frames, ev)
        s["src"] = src                                                 # keep the host buffer alive until the copy has been ordered
        s["n"] = int(src.shape[0])                                     # (a DataLoader with drop_last=False ends on a shorter batch)

    def __iter__(self):
        pending = []
        k = 0
        for frames in self.batches:
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
        cur.wait_event(s["ready"])                                     # no host synchronisation
        j = self._k % self._nout
        self._k += 1
        if self._done[j] is not None:
            cur.wait_event(self._done[j])                              # a consumer on another stream has finished with this buffer
            self._done[j] = None
        n = s.get("n", s["dev"].shape[0])
        out = self._outs[j][:n]
        preprocess_rgb(s["dev"][:n], self.size, out=out)               # 35 us at 8 x 512x1024; behind the previous forward in stream order
        ev = torch.cuda.Event(); ev.record(cur); s["free"] = ev        # the slot's device frame may be overwritten after this point
        return out
