"""What the chip does during the timed region: shader clock and power (rocm-smi / amd-smi, sampled from a side thread) while forwards run — pipelined(3) at 8
panoramas, plain calls, one layer3 convolution back to back, the resample pair back to back — each for SECONDS (default 4).  If the clock sits well under 2.4 GHz
with the power at its cap, a step is bounded by the ENERGY of its instructions (fewer matrix instructions / LDS bytes / vector instructions per panorama help,
a better overlap of the same ones does not); at full clock with power to spare it is bounded by latency and occupancy."""
import os, sys, time, subprocess, threading, re, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from omnifusion_amd import _lib
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.model._engine import split_weights_f16x3
from omnifusion_amd.weights import make_state_dict
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
SECONDS = float(os.environ.get("SECONDS", "4"))
lib = _lib.load()


import glob
_HW = None


def _hwmon():
    """the amdgpu hwmon directory of THE GPU torch runs on (matched by PCI address: a box may show the hwmon files of the host's other GPUs too —
    card0 is not necessarily this one), or None.  power1_average / power1_input in microwatts, freq1_input = sclk in Hz."""
    global _HW
    if _HW is None:
        _HW = ""
        try:
            p = torch.cuda.get_device_properties(0)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            c = [d for d in sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")) if os.path.exists(d + "/freq1_input")]
            _HW = c[0] if c else ""
        except Exception:
            pass
    return _HW or None


def sample():
    """(sclk MHz, power W): sysfs hwmon where the container shows it (two file reads: does not disturb the thread that enqueues kernels), else rocm-smi's JSON"""
    hw = _hwmon()
    if hw:
        try:
            f = float(open(hw + "/freq1_input").read()) / 1e6
            pf = hw + ("/power1_average" if os.path.exists(hw + "/power1_average") else "/power1_input")
            return f, float(open(pf).read()) / 1e6
        except Exception:
            pass
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, timeout=5).stdout.decode()
        d = json.loads(out); c = d[sorted(d)[0]]
        sclk = next((float(re.search(r"(\d+)", v).group(1)) for k, v in c.items() if "sclk" in k.lower() and re.search(r"\d", str(v))), float("nan"))
        pw = next((float(re.search(r"([\d.]+)", str(v)).group(1)) for k, v in c.items() if "power" in k.lower() and re.search(r"\d", str(v))), float("nan"))
        return sclk, pw
    except Exception:
        return float("nan"), float("nan")


def under(name, fn):
    stop, got = threading.Event(), []
    def watcher():
        while not stop.is_set():
            got.append(sample()); time.sleep(0.25)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=watcher); th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < SECONDS:
        fn(); n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    stop.set(); th.join()
    g = np.array(got[2:] if len(got) > 4 else got)
    print(f"{name:58s} {dt / n * 1e3:9.3f} ms per call | sclk MHz median {np.nanmedian(g[:, 0]):6.0f} (min {np.nanmin(g[:, 0]):.0f}, max {np.nanmax(g[:, 0]):.0f}) | power W median {np.nanmedian(g[:, 1]):6.0f} (max {np.nanmax(g[:, 1]):.0f}) | {len(g)} samples", flush=True)


print("idle:", sample(), "via", _hwmon() or "rocm-smi", flush=True)
try:
    print(subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmaxpower", "--showclocks"], capture_output=True, timeout=5).stdout.decode()[-900:])
except Exception as e:
    print("rocm-smi:", e)
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
rgb = torch.rand((8, 3, 512, 1024), device="cuda")
run = net.pipelined(3); pend = []
def piped():
    pend.append(run(rgb, confidence=True))
    if len(pend) > 3: pend.pop(0).get()
under("pipelined(3), 8 panoramas per forward", piped)
while pend: pend.pop(0).get()
under("plain calls, 8 panoramas", lambda: net(rgb, confidence=True))
one = rgb[:1].contiguous()
under("plain calls, ONE panorama", lambda: net(one, confidence=True))
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def sh(t):
    o = torch.empty_like(t); lib.omni_sh_from_f32(P(t), P(o), ctypes.c_size_t(t.numel()), S()); return o
for name, (M, H, W, C, Co) in {"layer3 convolution (tile kernel, 144 CUs)": (144, 8, 8, 256, 256), "layer1 convolution (halo kernel)": (144, 32, 32, 64, 64)}.items():
    x, r = sh(torch.randn(M, H, W, C, device="cuda")), sh(torch.randn(M, H, W, Co, device="cuda"))
    w = split_weights_f16x3(torch.randn(Co, 9 * C) / np.sqrt(9 * C)).cuda(); b = torch.randn(Co, device="cuda"); o = torch.empty(M, H, W, Co, device="cuda")
    def conv():
        for _ in range(20):
            lib.omni_conv2d_sh_f16x3_ws(P(x), None, P(w), P(b), P(r), P(o), 1, M, H, W, C, 0, Co, 3, 3, 1, 1, 1, 1, None, ctypes.c_size_t(0), S())
    under(name + " x 20 back to back", conv)
d = torch.rand((8, 18, 1, 256, 256), device="cuda")
def pair():
    for _ in range(10):
        equi2pers_patches(rgb, (80, 80), 4, (256, 256), layout=_lib.LAYOUT_BNCHW); pers2equi(d, (80, 80), 4, (256, 256), (512, 1024), None, layout=_lib.LAYOUT_BNCHW)
under("resample pair x 10 back to back", pair)
