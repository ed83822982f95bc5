"""Per-block timeline of the two resample kernels (DEBUG build, bit 16 of OMNI_E2P_DBG / OMNI_P2E_DBG).

Every block stamps wall_clock64 (100 MHz, chip-wide) at entry, after its set-up and at exit (stores drained) together with
HW_ID / XCC_ID.  Prints: kernel span, when blocks start / end, how many are resident over time, per-block durations.
"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from omnifusion_amd import _lib as L
lib = L.load_debug()
B, N, P, H, W = int(os.environ.get("B", "8")), 18, int(os.environ.get("P", "256")), 512, 1024
dev = "cuda:0"
x = torch.rand((B, 3, H, W), device=dev); out = torch.empty((B, N, 3, P, P), device=dev)
pp = torch.rand((B, N, 1, P, P), device=dev); erp = torch.empty((B, 1, H, W), device=dev)
NBLK = 1 << 16
trace = torch.zeros((2 * NBLK, 4), dtype=torch.int64, device=dev)       # second half: phase stamps of the plane-major pers2equi kernel
lib.omni_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))


def e2p():
    rc = lib.omni_equi2pers(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), 0, B, 3, H, W, P, P, 4, ctypes.c_float(80), ctypes.c_float(80), 1, None)
    assert rc == 0, lib.omni_last_error()


def p2e():
    rc = lib.omni_pers2equi(ctypes.c_void_p(pp.data_ptr()), ctypes.c_void_p(erp.data_ptr()), 0, B, 1, P, P, H, W, 4, ctypes.c_float(80), ctypes.c_float(80), 1, None)
    assert rc == 0, lib.omni_last_error()


XB = int(os.environ.get("XBITS", "0"))       # extra ablation bits OR-ed into the debug word (see the kernels' OMNI_DBG uses)
BRIEF = os.environ.get("BRIEF", "0") == "1"
FLUSH = os.environ.get("FLUSH", "0") == "1"
flush = torch.empty((768 << 20) // 4, device=dev) if FLUSH else None


def report(name, fn, env):
    os.environ[env] = str(XB)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1) / 20 * 1e3
    os.environ[env] = str(16 | XB)
    fn(); torch.cuda.synchronize()
    trace.zero_(); torch.cuda.synchronize()
    if FLUSH:                                  # the traced launch reads its input from HBM: 768 MB written in between empty the 256-MB memory-side cache
        flush.fill_(1.0); torch.cuda.synchronize()
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    os.environ[env] = "0"
    tall = trace.cpu().numpy()
    ph = tall[NBLK:][tall[:NBLK, 2] != 0]
    t = tall[:NBLK][tall[:NBLK, 2] != 0]
    t0 = t[:, 0].min()
    st, su, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0     # us
    hw = t[:, 3] & 0xffffffff; xcc = (t[:, 3] >> 32) & 0xff; extra = t[:, 3] >> 40
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    cuid = xcc * 64 + se * 16 + sh * 8 + cu          # (an id, not necessarily dense)
    print(f"== {name} B={B} P={P} xbits={XB}: {plain:.1f} us per launch un-instrumented, {e0.elapsed_time(e1)*1e3:.1f} us instrumented; {len(t)} blocks, "
          f"span first start -> last end {en.max():.1f} us")
    q = lambda v: " ".join(f"{np.percentile(v, p):.1f}" for p in (0, 10, 50, 90, 100))
    print(f"   start  (min p10 p50 p90 max) us: {q(st)}")
    print(f"   end                           : {q(en)}")
    print(f"   set-up duration               : {q(su - st)}")
    print(f"   block duration                : {q(en - st)}")
    for k in sorted(set(extra.tolist())):
        m = extra == k
        print(f"   work class {k}: {m.sum()} blocks, duration p50 {np.percentile((en - st)[m], 50):.1f} us p90 {np.percentile((en - st)[m], 90):.1f}")
    if ph[:, 0].any():
        a1, a2, a3, a4 = (t[:, 1] - t[:, 0]) / 100.0, (ph[:, 0] - t[:, 1]) / 100.0, (ph[:, 1] - ph[:, 0]) / 100.0, (ph[:, 2] - ph[:, 1]) / 100.0
        a5 = (t[:, 2] - ph[:, 2]) / 100.0
        for k in sorted(set(extra.tolist())):
            m = extra == k
            print(f"   class {k} phases p50 us: loads {np.median(a1[m]):.2f} | DMA addresses + ring fill {np.median(a2[m]):.2f} | taps {np.median(a3[m]):.2f} | stages {np.median(a4[m]):.2f} | normalise + stores {np.median(a5[m]):.2f}")
    if BRIEF: return
    edges = np.arange(0, en.max() + 2, 2.0)
    res = [int(((st <= e) & (en > e)).sum()) for e in edges]
    print("   resident blocks every 2 us: " + " ".join(str(r) for r in res))
    ids, cnt = np.unique(cuid, return_counts=True)
    print(f"   distinct CU ids {len(ids)}; blocks per CU min {cnt.min()} p50 {int(np.median(cnt))} max {cnt.max()}")
    wsum = np.array([extra[cuid == i].sum() for i in ids]); lend = np.array([en[cuid == i].max() for i in ids])
    print(f"   per CU: sum of work classes min {wsum.min()} p50 {int(np.median(wsum))} max {wsum.max()}; last end min {lend.min():.1f} p50 {np.median(lend):.1f} max {lend.max():.1f} us; "
          f"correlation(work sum, last end) {np.corrcoef(wsum, lend)[0, 1]:.2f}")
    for xx in range(8):
        m = xcc == xx
        if m.any(): print(f"   XCD {xx}: {m.sum()} blocks, last end {en[m].max():.1f} us, median end {np.median(en[m]):.1f}")
    # second-wave blocks: those that start after the first block ended
    first_end = en.min()
    late = st > first_end
    print(f"   blocks started after the first one ended: {late.sum()} ({100.0*late.mean():.0f} %), their duration p50 {np.percentile((en-st)[late], 50) if late.any() else 0:.1f} us "
          f"vs early ones {np.percentile((en-st)[~late], 50):.1f} us")


report("equi2pers", e2p, "OMNI_E2P_DBG")
report("pers2equi", p2e, "OMNI_P2E_DBG")
