"""Investigation: tools/pk/libtaps_dump.so (pers2equi's tap arithmetic WITH packed fp32 instructions, every intermediate stored) on
stream 1 beside a real product convolution on stream 2.  Prints, per intermediate, how many values differ from a quiet run."""
import os, sys, ctypes, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model._engine import split_weights_f16x3
lib = L.load()
td = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("TAPS_LIB", "libtaps_dump2.so")))
NP = 18
class TapArgs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("kx", "ky", "half_h", "half_w", "fw", "fh")] + [("H", ctypes.c_int), ("W", ctypes.c_int), ("mask", ctypes.c_uint)] + \
               [(n, ctypes.c_float * NP) for n in ("sp", "cp", "sl0", "cl0")]
td.taps_dump2.argtypes = [TapArgs, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
H, W, P = 512, 1024, 128
lat = np.linspace(-np.pi / 2, np.pi / 2, H, dtype=np.float32).astype(np.float64); lon = np.linspace(-np.pi, np.pi, W, dtype=np.float32).astype(np.float64)
rt = torch.from_numpy(np.stack([np.sin(lat), np.cos(lat)], 1).astype(np.float32)).cuda(); ct = torch.from_numpy(np.stack([np.sin(lon), np.cos(lon)], 1).astype(np.float32)).cuda()
k = 1.0 / math.tan(math.radians(40.0))
cen = [(-67.5, 0), (-67.5, 120), (-67.5, -120)] + [(-22.5, 60 * q - 150) for q in range(6)] + [(22.5, 60 * q - 150) for q in range(6)] + [(67.5, 0), (67.5, 120), (67.5, -120)]
F = ctypes.c_float * NP
a = TapArgs(k, k, P / 2.0, P / 2.0, float(P), float(P), H, W, int(os.environ.get("MASK", "0x3ffff"), 0),
            F(*[math.sin(math.radians(c[0])) for c in cen]), F(*[math.cos(math.radians(c[0])) for c in cen]),
            F(*[math.sin(math.radians(c[1])) for c in cen]), F(*[math.cos(math.radians(c[1])) for c in cen]))
NF = 16
names = "cd sd cos_c rc nx ny X Y wa wb wc wd q0=sp*slat q1=cp*clat q2=q1*cd acc".split()
if os.environ.get("TAPS_CO"): names = "cd sd cos_c rc nx ny X Y wa wb wc wd wsum l1 acc n".split()      # (the layout fail_base.s was compiled with)
Pp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
CO = os.environ.get("TAPS_CO")                                  # a device code object (patched assembly): launched through the module API
if CO:
    hip = ctypes.CDLL("libamdhip64.so")
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), os.path.join(os.path.dirname(os.path.abspath(__file__)), CO).encode()) == 0
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"_Z17taps_dump2_kernel8TapArgs2PK15HIP_vector_typeIfLj2EES3_Pf") == 0
    class KArgs(ctypes.Structure):
        _fields_ = [("a", TapArgs), ("pad", ctypes.c_int), ("rt", ctypes.c_void_p), ("ct", ctypes.c_void_p), ("out", ctypes.c_void_p)]
    assert KArgs.rt.offset == 0x148, hex(KArgs.rt.offset)
def victim(out):
    if CO:
        ka = KArgs(a, 0, rt.data_ptr(), ct.data_ptr(), out.data_ptr()); sz = ctypes.c_size_t(ctypes.sizeof(ka))
        extra = (ctypes.c_void_p * 5)(1, ctypes.cast(ctypes.byref(ka), ctypes.c_void_p), 2, ctypes.cast(ctypes.byref(sz), ctypes.c_void_p), 3)
        rc = hip.hipModuleLaunchKernel(fn, (H // 4) * (W // 64), 1, 1, 256, 1, 1, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None, extra)
        assert rc == 0, rc
        return
    rc = td.taps_dump2(a, Pp(rt), Pp(ct), Pp(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); assert rc == 0
ref = torch.zeros((H * W * NP, NF), device="cuda"); victim(ref); torch.cuda.synchronize()
r2 = torch.zeros_like(ref); victim(r2); torch.cuda.synchronize(); assert torch.equal(ref.view(torch.int32), r2.view(torch.int32))
M = 144
def sh(t):
    o = torch.empty_like(t); lib.omni_sh_from_f32(Pp(t), Pp(o), ctypes.c_size_t(t.numel()), L.stream_of(t)); return o
def conv_noise(Hh, C1, Cout, k=3, pad=None):
    x = sh(torch.randn(M, Hh, Hh, C1, device="cuda")); K = k * k * C1
    w16 = split_weights_f16x3(torch.randn(Cout, K) / np.sqrt(K)).cuda(); b = torch.randn(Cout, device="cuda")
    pad = k // 2 if pad is None else pad
    Ho = Hh + 2 * pad - k + 1
    out = torch.empty(M, Ho, Ho, Cout, device="cuda"); ws = torch.empty(M * Ho * Ho * Cout, device="cuda")
    def run():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.omni_conv2d_sh_f16x3_ws(Pp(x), None, Pp(w16), Pp(b), None, Pp(out), 1, M, Hh, Hh, C1, 0, Cout, k, k, 1, pad, 1, 1, Pp(ws), ctypes.c_size_t(ws.numel() * 4), st) == 0
    return run
noises = {"none": lambda: None, "halo conv 32->32 @128": conv_noise(128, 32, 32), "conv_sh 128->128 @16": conv_noise(16, 128, 128),
          "conv_sh 128->128 @16 pad 0": conv_noise(16, 128, 128, pad=0)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
outs = [torch.zeros_like(ref) for _ in range(4)]
for name, nz in noises.items():
    bad = np.zeros(NF, dtype=np.int64); launches = 0; example = None
    for rep in range(6):
        with torch.cuda.stream(s2):
            for _ in range(40): nz()
        with torch.cuda.stream(s1):
            for o in outs: victim(o)
        torch.cuda.synchronize()
        for o in outs:
            d = (o.view(torch.int32) != ref.view(torch.int32))
            launches += 1
            if bool(d.any()):
                bad += d.sum(0).cpu().numpy()
                lanes = ((d.any(1).nonzero()[:, 0] // NP) % 64).cpu().numpy()
                print("      lanes of the differing (pixel, patch) pairs:", np.bincount(lanes, minlength=64), flush=True) if example is None else None
                if example is None:
                    idx = d.any(1).nonzero()[0, 0].item()
                    example = (idx, o[idx].cpu().numpy().copy(), ref[idx].cpu().numpy().copy())
    print("noise %-28s launches %3d  differing values per intermediate: %s" % (name, launches, " ".join("%s=%d" % (n, b) for n, b in zip(names, bad) if b)), flush=True)
    if example is not None:
        idx, got, want = example
        px = idx // NP
        print("   first differing (pixel, patch) %d: row %d, col %d, lane %d, patch %d" % (idx, px // W, px % W, px % 64, idx % NP))
        for n, g, w_ in zip(names, got, want):
            print("      %-6s got %-16.9g want %-16.9g %s" % (n, g, w_, "" if np.float32(g).view(np.int32) == np.float32(w_).view(np.int32) else "<--"))
