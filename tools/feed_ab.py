"""Where the host-fed rate (0.92 of the resident-input rate) goes: the pipelined forward (3 in flight, 8 panoramas) fed
  a) from ONE resident float32 batch (bench.py's `value`), b) from 2 rotating resident float32 batches, c) + prep_rgb_kernel from a resident uint8 frame,
  d) + the H2D copies on the side stream but the forward reading a resident batch, e) the full DeviceFeeder (bench.py's `host_fed`): tools/feed_ab.py"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.data import DeviceFeeder, preprocess_rgb
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
B, H, W, NB = 8, 512, 1024, 260
dev = torch.device("cuda:0")
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
run = net.pipelined(3)
g = torch.Generator().manual_seed(1)
host = [torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(4)]
dev_u8 = [h.to(dev) for h in host]
res = [preprocess_rgb(dev_u8[k], (H, W)).clone() for k in range(2)]
outs = [torch.empty((B, 3, H, W), device=dev) for _ in range(2)]
side = torch.cuda.Stream(device=dev)
stage = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(3)]


def loop(make, hook=None):
    pend, nret, tf = collections.deque(), 0, None
    for k in range(NB):
        x = make(k)
        p = run(x, confidence=True)
        if hook: hook(k, p)
        pend.append(p)
        if len(pend) > 3:
            pend.popleft().get(); nret += 1
            if nret == 40: torch.cuda.synchronize(); tf = time.perf_counter()
    while pend: pend.popleft().get()
    torch.cuda.synchronize()
    return B * (NB - 40 - 3) / (time.perf_counter() - tf)


done = [None, None]
def prep(k):
    j = k % 2
    if done[j] is not None: torch.cuda.current_stream().wait_event(done[j]); done[j] = None
    return preprocess_rgb(dev_u8[k % 4], (H, W), out=outs[j])
def prep_hook(k, p): done[k % 2] = p.input_read
def h2d_only(k):
    with torch.cuda.stream(side):
        stage[k % 3].copy_(host[k % 4], non_blocking=True)
    return res[0]
def feeder_loop():
    feeder = DeviceFeeder((host[k % 4] for k in range(NB)), (H, W), device=dev, out_buffers=2)
    pend, nret, tf = collections.deque(), 0, None
    for frame in feeder:
        p = run(frame, confidence=True); feeder.done_with(frame, p.input_read); pend.append(p)
        if len(pend) > 3:
            pend.popleft().get(); nret += 1
            if nret == 40: torch.cuda.synchronize(); tf = time.perf_counter()
    while pend: pend.popleft().get()
    torch.cuda.synchronize()
    return B * (NB - 40 - 3) / (time.perf_counter() - tf)


for rnd in range(3):
    a = loop(lambda k: res[0]); b = loop(lambda k: res[k % 2]); c = loop(prep, prep_hook); d = loop(h2d_only); e = feeder_loop()
    print(f"round {rnd}: a one resident batch {a:6.0f} | b two rotating {b:6.0f} | c + prep kernel {c:6.0f} | d H2D beside, resident input {d:6.0f} | e DeviceFeeder {e:6.0f} panoramas/s", flush=True)
