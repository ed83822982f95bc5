"""PNG-fed rate against the number of batch decoders (png.PngBatches workers): the decode pool alone (steady state, 12 s budget split) and with the pipelined forward."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from omnifusion_amd import png
from omnifusion_amd.data import DeviceFeeder
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
B = 8
base = bench.synthetic_photo(512, 1024, 900)
files = [bench.encode_png_bgr(np.roll(base, (131 * k, 517 * k), axis=(0, 1))) for k in range(16)]
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
run = net.pipelined(3)
for w in (2, 4, 6, 8, 12, 16):
    nb = 400
    pb = png.PngBatches([files[i % 16] for i in range(B * nb)], B, pinned=True, workers=w)
    t0, n = None, 0
    for buf in pb:
        pb.recycle(buf); n += 1
        if n == 40: t0 = time.perf_counter()
    pool = B * (n - 40) / (time.perf_counter() - t0)
    feeder = DeviceFeeder(png.PngBatches([files[i % 16] for i in range(B * nb)], B, pinned=True, workers=w), (512, 1024), device="cuda", out_buffers=2)
    pend, nret, tf = collections.deque(), 0, None
    for frame in feeder:
        p = run(frame, confidence=True); feeder.done_with(frame, p.input_read); pend.append(p)
        if len(pend) > 3:
            pend.popleft().get(); nret += 1
            if nret == 40: torch.cuda.synchronize(); tf = time.perf_counter()
    while pend: pend.popleft().get()
    torch.cuda.synchronize()
    print("workers %2d (x %d threads): decode pool alone %6.0f panoramas/s, PNG-fed forward %6.0f panoramas/s" % (w, B, pool, B * (nb - 40 - 3) / (time.perf_counter() - tf)), flush=True)
