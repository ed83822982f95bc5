import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from omnifusion_amd import png
B = 8
base = bench.synthetic_photo(512, 1024, 900)
files = [bench.encode_png_bgr(np.roll(base, (131 * k, 517 * k), axis=(0, 1))) for k in range(16)]
for pinned in (False, True):
    for w in (1, 2, 4, 8):
        nb = 300
        pb = png.PngBatches([files[i % 16] for i in range(B * nb)], B, pinned=pinned, workers=w)
        t0, n = None, 0
        for buf in pb:
            pb.recycle(buf); n += 1
            if n == 40: t0 = time.perf_counter()
        print("pinned", pinned, "workers", w, "%.0f pano/s" % (B * (n - 40) / (time.perf_counter() - t0)), flush=True)
# raw C call from many Python threads without PngBatches
import threading, ctypes
def loop(k, out, cnt):
    for _ in range(cnt): png.decode_batch(files[:8], out=out, threads=0)
for nt in (1, 2, 4, 8):
    outs = [torch.empty((8, 512, 1024, 3), dtype=torch.uint8) for _ in range(nt)]
    ths = [threading.Thread(target=loop, args=(k, outs[k], 60)) for k in range(nt)]
    t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]
    print("raw decode_batch from", nt, "python threads: %.0f pano/s" % (nt * 60 * 8 / (time.perf_counter() - t0)), flush=True)
