"""PNG decoding for the host input pipeline — the two `cv2.imread` calls of /root/reference/dataset_loader_stanford.py (:85 RGB panorama,
:96 16-bit depth map) on a pool of host threads, straight into pinned memory (csrc/omni_png.hip, csrc/omni_inflate.h: chunk walk, checksums, inflate, scan-line reconstruction — no third-party code on the path).

    rgb_u8  = imread(path)                 # uint8 [H,W,3], B G R order          == cv2.imread(path)
    depth   = imread(path, unchanged=True) # uint16 [H,W] (or uint8)             == cv2.imread(path, -1) of a gray file
    frames  = decode_batch(list_of_paths_or_bytes, pinned=True)                  # uint8 [n,H,W,3]: what DeviceFeeder takes

No GPU is involved; the arrays feed `omnifusion_amd.data.DeviceFeeder` / `preprocess_rgb` / `preprocess_depth`.
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib


def effective_cpus():
    """CPUs this process may really use: the affinity mask, capped by the container's CPU-time quota (cgroup v2 `cpu.max` / v1 cfs quota) — a box
    can show 256 hardware threads and grant 16 CPUs; threads beyond that only take turns."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return n


def _bytes_of(src):
    if isinstance(src, (bytes, bytearray, memoryview)):
        return bytes(src)
    with open(src, "rb") as fh:
        return fh.read()


def png_info(src):
    """(height, width, bit_depth, colour_type) from the header"""
    data = _bytes_of(src)
    w, h, d, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.load().omni_png_info(data, ctypes.c_size_t(len(data)), ctypes.byref(w), ctypes.byref(h), ctypes.byref(d), ctypes.byref(c)), "png_info")
    return h.value, w.value, d.value, c.value


def imread(src, unchanged=False):
    """cv2.imread(src) (BGR uint8 [H,W,3]) or, unchanged=True, cv2.imread(src, -1) of a single-channel file (uint8 / uint16 [H,W])."""
    data = _bytes_of(src)
    H, W, depth, ctype = png_info(data)
    if unchanged:
        out = np.empty((H, W), np.uint16 if depth == 16 else np.uint8)
    else:
        out = np.empty((H, W, 3), np.uint8)
    _lib.check(_lib.load().omni_png_decode(data, ctypes.c_size_t(len(data)), out.ctypes.data_as(ctypes.c_void_p), H, W, 1 if unchanged else 0), "png_decode")
    return out


def decode_batch(srcs, unchanged=False, pinned=False, threads=0, out=None):
    """n PNG files of ONE size -> torch tensor uint8 [n,H,W,3] (or [n,H,W] uint8 / int16-stored uint16 when unchanged=True), decoded on
    `threads` host threads (0: all).  pinned=True allocates page-locked memory: `DeviceFeeder` then copies it to the device without staging."""
    datas = [_bytes_of(s) for s in srcs]
    n = len(datas)
    if n == 0:
        raise ValueError("decode_batch: no files")
    H, W, depth, ctype = png_info(datas[0])
    if unchanged:
        shape, dt = (n, H, W), (torch.int16 if depth == 16 else torch.uint8)       # (torch has no uint16 arithmetic: preprocess_depth reinterprets)
    else:
        shape, dt = (n, H, W, 3), torch.uint8
    if out is None:
        out = torch.empty(shape, dtype=dt, pin_memory=bool(pinned and torch.cuda.is_available()))
    elif tuple(out.shape) != shape or out.dtype != dt or out.is_cuda or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous host tensor {shape} of {dt}")
    per = out[0].numel() * out.element_size()
    ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(d), ctypes.c_void_p) for d in datas])
    sizes = (ctypes.c_size_t * n)(*[len(d) for d in datas])
    dsts = (ctypes.c_void_p * n)(*[out.data_ptr() + i * per for i in range(n)])
    _lib.check(_lib.load().omni_png_decode_batch(ptrs, sizes, dsts, n, H, W, 1 if unchanged else 0, int(threads)), "png_decode_batch")
    return out


class PngBatches:
    """Iterable of decoded batches for `DeviceFeeder`: `paths` (RGB panorama files of one size) in batches of `batch` frames, each decoded by
    the native thread pool into its own pinned buffer, AHEAD of the consumer on `workers` background threads (the decode of batches k+1 ..
    k+workers runs while batch k crosses PCIe and the network) — the role of the reference's 8 DataLoader workers (test.py:90-97) for the
    decode step.  A PNG is one deflate stream: an image occupies ONE thread, so a batch of 8 keeps 8 host threads busy whatever the pool's
    size (round 5 measured 1100 panoramas/s that way against 3800 consumed): several batches are decoded side by side, delivered in order.
    workers = 0: enough batches side by side to keep the CPUs this process is GRANTED busy (`effective_cpus`), 2 .. 16."""

    def __init__(self, paths, batch, threads=0, pinned=True, ring=0, workers=0):
        self.paths, self.batch, self.threads, self.pinned = list(paths), int(batch), int(threads), pinned
        ncpu = effective_cpus()
        self.workers = int(workers) if workers > 0 else max(2, min(16, ncpu // max(1, self.batch) + 1))   # enough batches in work to keep every granted CPU decoding
        self.ring = max(2, int(ring), self.workers + 1)            # decoded-but-unconsumed batches (= buffers in flight)
        self._pool, self._pool_lock = [], threading.Lock()         # buffers handed back by the consumer: (tensor, event or None)

    def __len__(self):
        return (len(self.paths) + self.batch - 1) // self.batch

    def recycle(self, buf, event=None):
        """The consumer is done with `buf` (a batch this iterable yielded) once `event` has completed (None: now): the producers decode a
        later batch into it instead of page-locking fresh memory (`DeviceFeeder` calls this with the event behind its H2D copy).  Without
        this call every batch gets a buffer of its own, as a DataLoader's would."""
        with self._pool_lock:
            if len(self._pool) < self.ring + 2:
                self._pool.append((buf, event))

    def _buffer(self, n):
        with self._pool_lock:
            for i, (buf, ev) in enumerate(self._pool):
                if buf.shape[0] == n and (ev is None or ev.query()):
                    del self._pool[i]
                    return buf
        return None

    def __iter__(self):
        nb = len(self)
        results, cv, stop = {}, threading.Condition(), threading.Event()
        slots = threading.Semaphore(self.ring)                     # bounds the batches decoded ahead; taken BEFORE an index, so the lowest are always in work
        state = {"next": 0}

        def worker():
            while not stop.is_set():
                if not slots.acquire(timeout=0.1):                  # never blocks past `stop`: an abandoned iterator must not strand a thread
                    continue
                with cv:
                    k = state["next"]
                    state["next"] += 1
                if k >= nb:
                    slots.release()
                    return
                chunk = self.paths[k * self.batch:(k + 1) * self.batch]
                try:
                    item = decode_batch(chunk, pinned=self.pinned, threads=self.threads, out=self._buffer(len(chunk)))
                except Exception as e:                               # surfaces in the consumer, at this batch's position
                    item = e
                with cv:
                    results[k] = item
                    cv.notify_all()
        pool = [threading.Thread(target=worker, daemon=True) for _ in range(min(self.workers, max(1, nb)))]
        for th in pool:
            th.start()
        try:
            for k in range(nb):
                with cv:
                    while k not in results:
                        cv.wait(0.5)
                    item = results.pop(k)
                slots.release()
                if isinstance(item, Exception):
                    raise item
                yield item
        finally:
            stop.set()
            for th in pool:
                th.join(timeout=5.0)
            results.clear()
