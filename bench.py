#!/usr/bin/env python
"""bench.py — headline benchmark of the `equi_pers` hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one full forward of the single-pass spherical_fusion model
(equi2pers -> ResNet-34 U-Net + transformer over the B*18 patches -> confidence pers2equi) over one
per-GPU batch of synthetic panoramas already resident in HBM.  Panoramas shard by image (no
data-path collective), so scaling is weak: every rank processes `--batch` panoramas
per step and `value` = all panoramas of all ranks / max-over-ranks time.

Workload notes (DESIGN.md "Measurement"):
  * the reference network only exists at patch size 128 (SURVEY.md finding 0.1: its token width is
    32*(P/32)^2 = 512 only for P = 128; P = 256 raises in the reference), so panoramas/s is measured
    at 512x1024 ERP, nrows = 4 (18 patches), P = 128;
  * the resample pair named by the metric (18 x 256^2 patches, equi2pers C=3 + pers2equi C=1) is
    timed in the same run and reported as `roofline_resample` (HBM-bound);
  * `roofline` is the dominant part of the step: the conv/GEMM network on the matrix cores.  In the default f16x3 mode
    every product block is THREE fp16 MFMAs on hi/lo half pairs (fp32-class accuracy), so the ceiling for ALGORITHMIC
    flops is the fp16 dense peak / 3 = 833 TFLOP/s; `achieved` is algorithmic (fp32-equivalent) TFLOP/s against that.
Prints ONE JSON line on rank 0.
"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3-6.9 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak (= the fp32 vector peak): the OMNI_NET_PRECISION=fp32 mode
MFMA_F16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md); the f16x3 mode issues 3 of them per product
ERP_H, ERP_W, NROWS, NPATCH, FOV = 512, 1024, 4, 18, (80.0, 80.0)
NET_GFLOP_PER_PANO = 71.3      # 2 x 35.66 GMAC at P=128, N=18 (probed with forward hooks)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=8, help="panoramas per GPU per step (8 = BASELINE cfg 4 shard)")
    ap.add_argument("--depth", type=int, default=3, help="forwards in flight per GPU (spherical_fusion.pipelined); 1 = one at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def _omp_threads(n):
    """thread count of BOTH OpenMP users of the CPU leg: torch's intra-op pool and the C oracle's `#pragma omp parallel for`."""
    import ctypes
    torch.set_num_threads(n)
    for name in ("libgomp.so.1", "libomp.so", "libomp.so.5"):
        try:
            ctypes.CDLL(name).omp_set_num_threads(n)
        except OSError:
            pass


def cpu_baseline():
    """The CPU oracle ('port' of the reference forward: oracle/model_ref.py on torch-CPU fp32 + the C
    restatement of equi2pers/pers2equi) timed on this host's cores on a bounded sample — with every core the box has
    (at most 64) AND with one thread (n in {1, all physical cores}, both reported)."""
    from oracle import c_oracle as co, model_ref
    from omnifusion_amd.weights import make_state_dict
    co.build()
    from omnifusion_amd.png import effective_cpus
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    granted = effective_cpus()                  # the container's CPU-time quota: the GPU boxes of this pool show 256 hardware threads and grant 16 CPUs
    sd = make_state_dict(42, NPATCH, False)
    rgb = torch.rand((1, 3, ERP_H, ERP_W), generator=torch.Generator().manual_seed(0))

    def run(nthreads, budget_s, nmax):
        _omp_threads(nthreads)
        model_ref.spherical_fusion_forward(sd, rgb[:, :, :64, :128], NROWS, 128, FOV, True)          # warm
        n, t0 = 0, time.perf_counter()
        while True:
            model_ref.spherical_fusion_forward(sd, rgb, NROWS, 128, FOV, True)
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= nmax:
                break
        return n, dt
    n, dt = run(threads, 10.0, 8)
    n1, dt1 = run(1, 8.0, 2)
    _omp_threads(threads)
    return {"value": n / dt, "unit": "panoramas/s", "cores": threads, "cpus_granted": granted, "kind": "port",
            "value_1thread": n1 / dt1, "s_per_panorama_1thread": dt1 / n1, "s_per_panorama": dt / n,
            "sample": f"{n} panorama(s) 512x1024 on {threads} threads ({granted} CPUs granted by the container's cgroup quota) + {n1} on 1 thread: single-pass model P=128 confidence=True, torch-CPU fp32 oracle "
                      f"+ C/OpenMP equi2pers/pers2equi (the reference's own dense tables are not re-read per call here: this port is faster than the reference's Python)"}


def kernel_us(fns, dev, reps=20):
    """Average launch duration (s) of each callable's kernel, HIP events on the launch stream: ONE event pair around `reps` back-to-back
    launches of the same kernel, queued behind a few milliseconds of device-side spinning so that the host is never the one being timed.
    This is the form that agrees with the dispatch time stamps of `rocprofv3 --kernel-trace --stats` (tools/evmethod.py, profiles/
    r03b_event_method.txt: 36.1 / 15.9 us against 36.6 / 16.0 us; an event pair around EVERY launch reads 41.2 / 20.4 us — each
    bracket adds ~4.5 us of command-processor time to a 16-us kernel)."""
    out = []
    for f in fns:
        for _ in range(3): f()
        torch.cuda.synchronize(dev)
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(int(12e6))                          # ~5 ms of device time: the queue fills while it spins
            e0.record()
            for _ in range(reps): f()
            e1.record()
            torch.cuda.synchronize(dev)
            ts.append(e0.elapsed_time(e1) / reps * 1e-3)
        out.append(float(np.mean(ts)))
    return out


def kernel_us_rotating(make_fn, inputs, dev, reps=20):
    """kernel_us() over ROTATING buffer sets: `inputs` are R distinct input tensors and the last R outputs are kept alive, so
    that consecutive launches touch R different input AND output buffers — R chosen by the caller so that the R sets together exceed the
    256-MB memory-side cache.  The non-rotating figure is what the operators see inside a forward (their input was just written by the
    previous kernel and sits in that cache); this one is the rate from HBM."""
    R = len(inputs)
    keep = collections.deque(maxlen=R)
    for k in range(2 * R):
        keep.append(make_fn(inputs[k % R]))
    torch.cuda.synchronize(dev)
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(12e6))
        e0.record()
        for k in range(reps):
            keep.append(make_fn(inputs[k % R]))
        e1.record()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1) / reps * 1e-3)
    keep.clear()
    return float(np.mean(ts))


def resample_pair(dev, B, H, W, nrows, P, dtype, reps=20, ref_layout=False):
    """The resample pair (equi2pers C = 3, pers2equi C = 1) at one BASELINE shape: seconds per kernel + algorithmic bytes.  Planar layout (what the
    model uses inside) or, ref_layout=True, the reference's own [B,C,ph,pw,N] (what the drop-in equi2pers() returns / pers2equi() takes)."""
    from omnifusion_amd import _lib
    from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
    from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
    N = {3: 10, 4: 18, 5: 26, 6: 46}[nrows]
    s = 2 if dtype == torch.float16 else 4
    x = torch.rand((B, 3, H, W), device=dev).to(dtype)
    d = (torch.rand((B, 1, P, P, N), device=dev) if ref_layout else torch.rand((B, N, 1, P, P), device=dev)).to(dtype)
    LAY = _lib.LAYOUT_BCHWN if ref_layout else _lib.LAYOUT_BNCHW
    t1, t2 = kernel_us([lambda: equi2pers_patches(x, FOV, nrows, (P, P), layout=LAY),
                        lambda: pers2equi(d, FOV, nrows, (P, P), (H, W), None, layout=LAY)], dev, reps)
    b1, b2 = B * 3 * (H * W + P * P * N) * s, B * (P * P * N + H * W) * s
    del x, d
    return {"equi2pers_us": t1 * 1e6, "pers2equi_us": t2 * 1e6, "bytes": b1 + b2, "GB/s": (b1 + b2) / (t1 + t2) / 1e9,
            "resample_pair_frac": (b1 + b2) / (t1 + t2) / 1e9 / HBM_PEAK_GBS}


def encode_png_bgr(frame_bgr, level=1):
    """A conforming 8-bit RGB PNG of one BGR frame (numpy only; scan-line filters Sub / Up / Average / Paeth by rows, as an adaptive encoder
    mixes them) — bench input for the decode pool, not product code."""
    import struct
    import zlib
    rgb = np.ascontiguousarray(frame_bgr[:, :, ::-1]).astype(np.int16)
    H, W, _ = rgb.shape
    cur = rgb.reshape(H, W * 3)
    prev = np.zeros_like(cur); prev[1:] = cur[:-1]
    out = np.empty((H, 1 + W * 3), np.uint8)
    for ft in (1, 2, 3, 4):                                                    # rows ft-1, ft+3, ...: Sub, Up, Average, Paeth
        c, u = cur[ft - 1::4], prev[ft - 1::4]
        left = np.zeros_like(c); left[:, 3:] = c[:, :-3]
        if ft == 1: pred = left
        elif ft == 2: pred = u
        elif ft == 3: pred = (left + u) >> 1
        else:
            ul = np.zeros_like(c); ul[:, 3:] = u[:, :-3]
            pa, pb, pc = np.abs(u - ul), np.abs(left - ul), np.abs(left + u - 2 * ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, u, ul))
        out[ft - 1::4, 0] = ft
        out[ft - 1::4, 1:] = ((c - pred) & 0xff).astype(np.uint8)
    chunk = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    z = zlib.compress(out.tobytes(), level)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b"")


def synthetic_photo(H, W, seed):
    """image-like uint8 BGR content (smooth structure + sensor-like noise; compresses ~2:1 like a photograph, unlike i.i.d. noise)"""
    r = np.random.default_rng(seed)
    low = r.random((H // 32 + 2, W // 32 + 2, 3)).astype(np.float32)
    big = np.kron(low, np.ones((32, 32, 1), np.float32))[:H + 32, :W + 32]
    k = 16
    c = np.cumsum(np.cumsum(big, 0), 1)                                        # box blur by summed-area table
    sm = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k])[:H, :W] / (k * k)
    return np.clip(sm * 255.0 + r.normal(0.0, 2.0, (H, W, 3)), 0, 255).astype(np.uint8)


def png_fed_rate(run, depth, dev, B, src_hw, nfiles, budget_s, pending):
    """PNG FILES -> omni_png_decode_batch on the host thread pool (png.PngBatches, one batch ahead) -> DeviceFeeder (pinned, async H2D) ->
    prep_rgb_kernel (INTER_AREA to 512x1024 when the files are larger, /255, CHW) -> the pipelined forward: panoramas/s with everything
    of dataset_loader_stanford.py:85-97 + test.py:90-97,196 inside the clock (VERDICT r4 #7c)."""
    from omnifusion_amd import png
    from omnifusion_amd.data import DeviceFeeder
    Hs, Ws = src_hw
    base = synthetic_photo(Hs, Ws, 900)
    files = [encode_png_bgr(np.roll(base, (131 * k, 517 * k), axis=(0, 1))) for k in range(nfiles)]     # distinct files, one synthesis
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    granted = png.effective_cpus()
    probe = png.PngBatches([files[i % nfiles] for i in range(B * 64)], B, threads=0, pinned=True)
    workers = probe.workers
    t0, nb = time.perf_counter(), 0                                           # the decode pool alone (no GPU work): the loader's ceiling
    for buf in probe:
        probe.recycle(buf)
        nb += 1
        if time.perf_counter() - t0 > min(2.0, budget_s / 3):
            break
    pool_rate = nb * B / (time.perf_counter() - t0)
    del probe
    per_s = pool_rate * 1.2                                                    # enough batches for ~budget_s at the slower of pool and GPU
    nbatch = max(12, int(min(per_s, 4000.0) * budget_s / B))
    paths = [files[i % nfiles] for i in range(nbatch * B)]
    feeder = DeviceFeeder(png.PngBatches(paths, B, threads=0, pinned=True), (ERP_H, ERP_W), device=dev, out_buffers=2 if depth > 1 else 1)
    nskip, nret, tf = min(4 * max(depth, 2), nbatch // 3), 0, None
    for frame_rgb in feeder:
        p_ = run(frame_rgb, confidence=True)
        if depth > 1:
            feeder.done_with(frame_rgb, p_.input_read)
        pending.append(p_)
        if len(pending) > depth:
            pending.popleft().get()
            nret += 1
            if nret == nskip:
                torch.cuda.synchronize()
                tf = time.perf_counter()
    while pending:
        pending.popleft().get()
    torch.cuda.synchronize()
    rate = B * (nbatch - nskip) / (time.perf_counter() - tf)
    raw_mb = Hs * Ws * 3 / 1e6
    return {"panoramas_per_s_per_gpu": rate, "file_size": [Hs, Ws], "file_MB": float(np.mean([len(f) for f in files])) / 1e6, "decoded_MB": raw_mb,
            "host_threads": ncpu, "cpus_granted": granted, "decode_workers": workers,                    # (a PNG is one deflate stream: one thread per image; the container's quota caps the pool)
            "decode_pool_alone_panoramas_per_s": pool_rate, "decode_MBps_per_granted_cpu": pool_rate * raw_mb / max(1, min(granted, workers * B)), "batches": nbatch}


def dominant_kernel(dev, B):
    """The dominant kernel of the step alone on the GPU: ONE layer3 convolution (3x3, 256 -> 256 channels, 8 x 8 images, B x 18 patches,
    residual + ReLU, split-half in and out: conv_sh_kernel<128,128,4,2,3,4,PP>, the form 11 of the 33 encoder convolutions and de_conv0_x run),
    HIP events around back-to-back launches.  Returns the per-launch figures the roofline check needs."""
    import ctypes
    from omnifusion_amd import _lib
    from omnifusion_amd.model._engine import split_weights_f16x3
    lib = _lib.load()
    P_ = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    S_ = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    M, H, W, C, Cout = B * NPATCH, 8, 8, 256, 256
    K = 9 * C

    def sh(t):
        o = torch.empty_like(t)
        assert lib.omni_sh_from_f32(P_(t), P_(o), ctypes.c_size_t(t.numel()), S_()) == 0
        return o
    x, res = sh(torch.randn(M, H, W, C, device=dev)), sh(torch.randn(M, H, W, Cout, device=dev))
    w16 = split_weights_f16x3(torch.randn(Cout, K) / np.sqrt(K)).to(dev)
    b = torch.randn(Cout, device=dev)
    out = torch.empty(M, H, W, Cout, device=dev)

    def run():
        rc = lib.omni_conv2d_sh_f16x3_ws(P_(x), None, P_(w16), P_(b), P_(res), P_(out), 1, M, H, W, C, 0, Cout, 3, 3, 1, 1, 1, 1, None, ctypes.c_size_t(0), S_())
        assert rc == 0, lib.omni_last_error()
    t = kernel_us([run], dev, 20)[0]
    gflop = 2.0 * M * H * W * Cout * K / 1e9
    blocks = ((M * H * W + 127) // 128) * (Cout // 128)
    return {"kernel": "conv_sh_kernel<128,128,4,2,3,4,PP> (one layer3 convolution: 3x3, 256->256, 8x8 images, %d patches, residual + ReLU)" % M,
            "gflop": gflop, "us_alone": t * 1e6, "blocks": blocks, "cus": 256,
            "achieved_algorithmic_TFLOPs": gflop / t / 1e3, "frac_algorithmic": gflop / t / 1e3 / MFMA_F16_PEAK_TFLOPS,
            "frac_issued": 3 * gflop / t / 1e3 / MFMA_F16_PEAK_TFLOPS,
            "frac_issued_on_the_cus_it_occupies": 3 * gflop / t / 1e3 / MFMA_F16_PEAK_TFLOPS * 256 / min(256, blocks),
            "note": "alone, the launch has %d blocks for 256 CUs (in the timed region other forwards' kernels run on the rest); frac_* against the 2.5-PF dense fp16 peak; "
                    "three fp16 matrix instructions are ISSUED per algorithmic product block (f16x3)" % blocks}


def profile_json(name):
    """profiles/<name> if it was measured on THIS build (its `build` = source_hash()), else (None, why) — per-kernel counters and time shares the
    bench line quotes but cannot measure itself (PMC passes, rocprofv3 --stats)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"profiles/{name} not present"
    with open(path) as fh:
        d = json.load(fh)
    from omnifusion_amd.build import source_hash
    if d.get("build") != source_hash():
        return None, f"profiles/{name} was measured on build {d.get('build')}, this is {source_hash()}"
    return d, ""


def pmc_traffic(B, name="resample_traffic.json"):
    """HBM bytes from the PMC counters — of the resample pair (one launch each; tools/pmc_traffic.sh -> profiles/resample_traffic.json) or of
    one forward of the network (tools/pmc_net.sh -> profiles/network_traffic.json) — if the file was measured on THIS build, else (None, why)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"profiles/{name} not present"
    with open(path) as fh:
        d = json.load(fh)
    from omnifusion_amd.build import source_hash
    if d.get("build") != source_hash():
        return None, f"profiles/{name} was measured on build {d.get('build')}, this is {source_hash()}"
    if d.get("B") != B:
        return None, f"profiles/{name} was measured at B={d.get('B')}"
    return d["traffic_bytes"], d.get("note", "")


def main():
    args = parse()
    from omnifusion_amd import dist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain process: become N ranks, one per GPU (the driver's own launch form:
        # python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)
        sys.exit(dist.respawn_under_launcher(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    # one rank per GPU over RCCL.  (OMNI_BENCH_DIST_BACKEND=gloo lets the multi-rank path be exercised with several ranks
    # sharing one GPU on a single-GPU box; the real launch is nccl = RCCL with LOCAL_RANK == device index.)
    backend = os.environ.get("OMNI_BENCH_DIST_BACKEND", "nccl")
    if dist.env_rank()[2] != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={dist.env_rank()[2]} ranks")
    rank, local, world, dev = dist.init(backend)
    local = dev.index

    from omnifusion_amd import _lib
    from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
    from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
    from omnifusion_amd.model.spherical_model import spherical_fusion
    from omnifusion_amd.weights import make_state_dict
    _lib.load()
    if world > 1:                                              # sharded runs reproduce single-GPU results bit for bit whatever the
        from omnifusion_amd.model._engine import Engine       # per-rank batch (SURVEY 8d parity gate): one split-K plan for all sizes
        Engine.latency_plan = False

    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    rgb = torch.rand((B, 3, ERP_H, ERP_W), generator=g).to(dev)               # synthetic RGB panoramas (BGR/255-like range)
    net = spherical_fusion(NROWS, NPATCH, (128, 128), FOV).cuda(local)
    net.load_state_dict(make_state_dict(42, NPATCH, False))                   # random-init weights of the reference architecture
    eng = net._eng
    LAY = _lib.LAYOUT_BNCHW

    # ---- the resample pair at the metric's patch size (18 x 256^2), same run, HIP events on the launch stream — measured BEFORE the
    # matrix-bound steps, in its own steady state (after 2500 panoramas/s of MFMA load the chip sits at its power-capped clock and the
    # same two kernels measure 5 % slower: round 2, 45.2 % in here vs 48 % in tools/kbench.py)
    P = 256
    depth_patches = torch.rand((B, NPATCH, 1, P, P), generator=g).to(dev)
    K = max(args.steps, 20)
    f_e2p = lambda: equi2pers_patches(rgb, FOV, NROWS, (P, P), layout=LAY)
    f_p2e = lambda: pers2equi(depth_patches, FOV, NROWS, (P, P), (ERP_H, ERP_W), None, layout=LAY)
    t_heat = time.perf_counter()
    while time.perf_counter() - t_heat < 0.2:
        f_e2p(); f_p2e()
        torch.cuda.synchronize()
    r_e2p, r_p2e = kernel_us([f_e2p, f_p2e], dev, K)
    # ... the same two kernels over rotating buffer sets whose total exceeds the 256-MB memory-side cache (3 x 164 MB, 6 x 55 MB)
    rot_in = [rgb] + [torch.rand_like(rgb) for _ in range(2)]
    rr_e2p = kernel_us_rotating(lambda x: equi2pers_patches(x, FOV, NROWS, (P, P), layout=LAY), rot_in, dev, K)
    rot_dp = [depth_patches] + [torch.rand_like(depth_patches) for _ in range(5)]
    rr_p2e = kernel_us_rotating(lambda d: pers2equi(d, FOV, NROWS, (P, P), (ERP_H, ERP_W), None, layout=LAY), rot_dp, dev, K)
    del rot_in, rot_dp
    # ... and as consecutive pipelined forwards run them: equi2pers of one batch beside pers2equi of another (two streams that really
    # run side by side, host wall clock over K pairs)
    from omnifusion_amd.model.spherical_model import _concurrent_streams
    sa, sb = _concurrent_streams(2, dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t_pair = time.perf_counter()
        for k in range(4 * K):
            with torch.cuda.stream(sa):
                f_e2p()
            with torch.cuda.stream(sb):
                f_p2e()
        torch.cuda.synchronize()
        t_pair = (time.perf_counter() - t_pair) / (4 * K)
    bytes_e2p = B * 3 * (ERP_H * ERP_W + P * P * NPATCH) * 4          # SURVEY 8d algorithmic bytes
    bytes_p2e = B * 1 * (P * P * NPATCH + ERP_H * ERP_W) * 4
    gbs_pair = (bytes_e2p + bytes_p2e) / (r_e2p + r_p2e) / 1e9
    # ... and at the other BASELINE shapes (rank 0 only: they are kernel figures, not whole-job ones)
    configs = {}
    if rank == 0:
        configs["b16"] = resample_pair(dev, 16, ERP_H, ERP_W, NROWS, 256, torch.float32)
        configs["reference_layout"] = resample_pair(dev, B, ERP_H, ERP_W, NROWS, 256, torch.float32, ref_layout=True)
        configs["reference_layout"]["note"] = "the same pair through the reference's own patch layout [B,C,256,256,18] (N innermost, equi2pers_v3.py:112-113): what the drop-in functions return / take; pers2equi = conversion to planar + the planar kernel"
        configs["cfg3"] = resample_pair(dev, 1, 1024, 2048, 6, 256, torch.float32)
        configs["cfg5"] = {"fp16": resample_pair(dev, 1, 2048, 4096, 6, 512, torch.float16),
                           "fp32": resample_pair(dev, 1, 2048, 4096, 6, 512, torch.float32),
                           "fp16_b4": resample_pair(dev, 4, 2048, 4096, 6, 512, torch.float16, reps=8)}
        torch.cuda.empty_cache()

    dominant = dominant_kernel(dev, B) if (rank == 0 and eng.precision == "f16x3") else None
    if dominant is not None:
        pm, why = profile_json("conv_pmc.json")                        # tools/pmc_conv.sh: mfma_busy of the same kernel form, hash-matched
        dominant["mfma_busy_from_profile"] = (pm or {}).get("mfma_busy", {}).get("conv_sh_kernel<128,128,4,2,3,4,PP>")
        dominant["mfma_busy_note"] = why or (pm or {}).get("note", "")
    shares, shares_note = profile_json("bench_kernel_shares.json")     # tools/prof_bench.sh: per-kernel-family shares of the timed region's GPU time
    # Bring the GPU out of its idle power state before anything is counted (sclk idles at ~366 MHz and takes tens of
    # milliseconds of load to ramp: a 5-step warm-up measured 1600-2200 panoramas/s on a box that then holds 2650).
    t_heat = time.perf_counter()
    while time.perf_counter() - t_heat < 0.3:
        net(rgb, confidence=True)
        torch.cuda.synchronize()
    # ---- stage breakdown: a few forwards one at a time, stage by stage, HIP events on the launch stream (not the timed region)
    nbr = max(5, min(args.steps, 20))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(nbr)]
    state = {"depth": None}

    def staged(k):                                                            # = spherical_fusion.forward, stage by stage
        e = ev[k] if k >= 0 else None
        if e: e[0].record()
        patches = equi2pers_patches(rgb, FOV, NROWS, (128, 128), layout=LAY)
        if e: e[1].record()
        a, c = net.network(patches, B, True)
        if e: e[2].record()
        state["depth"] = eng.blend(a, c, (ERP_H, ERP_W))
        if e: e[3].record()

    for k in range(-5, 0):
        staged(k)
    torch.cuda.synchronize()
    t_un = time.perf_counter()
    for k in range(nbr):
        staged(k)
    torch.cuda.synchronize()
    t_un = (time.perf_counter() - t_un) / nbr
    depth_seq = state["depth"].clone()
    sec = lambda i: float(np.mean([ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(nbr)])) * 1e-3
    t_e2p, t_net, t_p2e = sec(0), sec(1), sec(2)

    # ---- the timed region: W untimed steps, barrier + synchronize, EXACTLY K steps, barrier + synchronize, MAX over ranks
    # (omnifusion_amd/dist.py).  A step submits one complete forward of the batch; with --depth D > 1 up to D forwards are in
    # flight on D streams (spherical_fusion.pipelined: consecutive batches overlap, every kernel works on the whole batch) and a
    # step retires the forward submitted D steps earlier; the closing synchronize waits for all K.
    depth = max(1, args.depth)
    run = net.pipelined(depth)
    pending = collections.deque()

    def step():
        pending.append(run(rgb, confidence=True))
        if len(pending) > depth:
            state["depth"] = pending.popleft().get()

    dt = dist.timed_steps(step, args.steps, args.warmup, dev)
    while pending:
        state["depth"] = pending.popleft().get()
    torch.cuda.synchronize()
    # ... and a second region of at least 0.5 s (20 steps are 44 ms — too short to separate +-3 % of box noise); the contract's
    # K-step region above stays `value`, this one is reported beside it as `value_long`
    steps_long = max(args.steps, int(np.ceil(0.5 / (dt / args.steps))))
    dt_long = dist.timed_steps(step, steps_long, 0, dev)
    while pending:
        state["depth"] = pending.popleft().get()
    torch.cuda.synchronize()
    depth_map = state["depth"]
    assert depth_map.shape == (B, 1, ERP_H, ERP_W) and bool(torch.isfinite(depth_map).all())
    assert torch.equal(depth_map, depth_seq), "pipelined and one-at-a-time forwards must agree bit for bit"
    tflops = NET_GFLOP_PER_PANO * B * args.steps / dt / 1e3            # whole-GPU rate over the timed region (resample launches included)
    f16x3 = eng.precision == "f16x3"

    traffic, traffic_note = pmc_traffic(B)
    net_traffic, net_traffic_note = pmc_traffic(B, "network_traffic.json")

    # ---- the same step fed from HOST memory: decoded uint8 frames in pinned memory -> async H2D + /255 + CHW on a
    # side stream (omnifusion_amd/data.py), double-buffered against the forward.  Reported next to the resident-input `value`,
    # never as `value` (the PCIe-inclusive rate).
    from omnifusion_amd.data import DeviceFeeder
    host_frames = [torch.randint(0, 256, (B, ERP_H, ERP_W, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(4)]
    nfeed = max(200, 2 * args.steps)                                  # (20 batches in round 2: the first ones pay the start-up, 64 % measured)
    nskip = 4 * max(depth, 2)                                          # steady state: the clock starts when this many batches have been retired
    for _ in DeviceFeeder((host_frames[k % 4] for k in range(4)), (ERP_H, ERP_W), device=dev):
        pass
    torch.cuda.synchronize()
    feeder = DeviceFeeder((host_frames[k % 4] for k in range(nfeed + nskip)), (ERP_H, ERP_W), device=dev, out_buffers=2 if depth > 1 else 1)
    tf, nret = None, 0
    for frame_rgb in feeder:
        p_ = run(frame_rgb, confidence=True)
        if depth > 1:
            feeder.done_with(frame_rgb, p_.input_read)                 # the forward reads the batch on its own stream
        pending.append(p_)
        if len(pending) > depth:
            pending.popleft().get()
            nret += 1
            if nret == nskip:
                torch.cuda.synchronize()
                tf = time.perf_counter()
    while pending:
        pending.popleft().get()
    torch.cuda.synchronize()
    host_fed = B * nfeed / (time.perf_counter() - tf)
    png_fed = None
    if rank == 0 and os.environ.get("OMNI_BENCH_PNG", "1") != "0":
        png_fed = {"files_512x1024": png_fed_rate(run, depth, dev, B, (ERP_H, ERP_W), 16, 4.0, pending),
                   "files_2048x4096": png_fed_rate(run, depth, dev, B, (2048, 4096), 8, 4.0, pending),
                   "note": "PNG files (in memory: the page cache's role) -> omni_png_decode_batch on every host thread this rank may use, one batch ahead "
                           "(png.PngBatches; the decoder pool is sized to the CPUs the container GRANTS — cpus_granted — not to the hardware threads it shows) -> pinned buffer ring -> DeviceFeeder -> prep_rgb_kernel (INTER_AREA resize on the GPU for the 2048x4096 "
                           "files = Stanford2D3D's native panoramas, dataset_loader_stanford.py:92-97) -> pipelined forward; synthetic photo-like "
                           "content, filters Sub/Up/Average/Paeth by rows.  decode_pool_alone = the decode pool with no GPU work: the loader's ceiling"}
    # the link itself: one pinned 12.6-MB batch of frames, host -> device, back to back
    hbuf = torch.empty_like(host_frames[0], device=dev)
    for _ in range(3): hbuf.copy_(host_frames[0], non_blocking=True)
    torch.cuda.synchronize()
    th = time.perf_counter()
    for k in range(50): hbuf.copy_(host_frames[k % 4], non_blocking=True)
    torch.cuda.synchronize()
    h2d_gbps = 50 * host_frames[0].numel() / (time.perf_counter() - th) / 1e9

    # ---- BASELINE cfg 2 as written (ONE panorama per forward): latency-bound, reported next to the batched figure
    one = rgb[:1].contiguous()
    for _ in range(3):
        net(one, confidence=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        net(one, confidence=True)
    torch.cuda.synchronize()
    b1_ms = (time.perf_counter() - t1) / 10 * 1e3
    # ... the same forward replayed from ONE captured hipGraph (spherical_fusion.graphed): the launch sequence without the host's enqueue time
    run_g = net.graphed(one, confidence=True)
    for _ in range(3):
        run_g(one)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        run_g(one)
    torch.cuda.synchronize()
    b1_graph_ms = (time.perf_counter() - t1) / 20 * 1e3
    assert torch.equal(run_g(one), net(one, confidence=True)), "graph replay and eager single-panorama forwards must agree bit for bit"
    del run_g
    # ... and as a stream of single-panorama requests: 4 forwards in flight, one captured hipGraph per slot (at one panorama per
    # forward the host cannot enqueue ~135 launches as fast as several streams execute them)
    run1 = net.pipelined(4, graphs=True)
    q = collections.deque()

    def requests(n):
        for _ in range(n):
            q.append(run1(one, confidence=True))
            if len(q) > 4:
                q.popleft().get()
        while q:
            last = q.popleft().get()
        return last
    requests(12)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    last1 = requests(100)
    torch.cuda.synchronize()
    b1_stream_ms = (time.perf_counter() - t1) / 100 * 1e3
    assert torch.equal(last1, net(one, confidence=True)), "graph-replayed and plain single-panorama forwards must agree bit for bit"

    # ---- BASELINE cfg 3 as written: 1024x2048 ERP, nrows 6 (46 patches), 2-iteration spherical_fusion_iterative, one panorama
    cfg3_ms = None
    if rank == 0:
        from omnifusion_amd.model.spherical_model_iterative import spherical_fusion as spherical_fusion_iterative
        net3 = spherical_fusion_iterative(6, 46, (128, 128), FOV).cuda(local)
        net3.load_state_dict(make_state_dict(42, 46, True))
        x3 = torch.rand((1, 3, 1024, 2048), generator=g).to(dev)
        for _ in range(3):
            o3 = net3(x3, iter=2, confidence=False)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(10):
            o3 = net3(x3, iter=2, confidence=False)
        torch.cuda.synchronize()
        cfg3_ms = (time.perf_counter() - t3) / 10 * 1e3
        assert len(o3) == 2 and bool(torch.isfinite(o3[-1]).all())
        configs["cfg3"]["ms_per_forward"] = cfg3_ms
        configs["cfg3"]["note"] = "2-iteration iterative model at patch size 128 (SURVEY 0.1), one 1024x2048 panorama per forward; resample pair at 46 x 256^2, B = 1"
        del net3, x3, o3

    out = {
        "metric": "panoramas/sec at 512x1024 ERP, N=18 256^2 patches; equi2pers+pers2equi GB/s vs HBM peak",
        "value": world * B * args.steps / dt, "unit": "panoramas/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "value_long": world * B * steps_long / dt_long, "steps_long": steps_long, "ms_per_step_long": dt_long / steps_long * 1e3,
        "vs_baseline": None, "dtype": "f16x3 (fp16 hi/lo pairs, fp32 accumulate: fp32-class)" if f16x3 else "f32", "data": "synthetic",
        "config": {"workload": f"cfg2/cfg4 shard: {B} panoramas/GPU/step, 512x1024 ERP, fov 80, nrows 4 (18 patches); full "
                               "single-pass spherical_fusion forward (confidence=True) at patch size 128 — the only size the "
                               "reference network exists at (SURVEY 0.1); random-init weights (seed 42); inputs resident in HBM; "
                               "resample pair at 18x256^2 reported in roofline_resample",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"image-sharded x{world}",
                   "execution": (f"{depth} forwards in flight per GPU on {depth} streams (spherical_fusion.pipelined)" if depth > 1 else
                                 "one forward at a time" + (f", two half-batch lanes" if B >= 2 * net.LANES else ""))},
        "pipelining": {"depth": depth, "value_one_at_a_time": world * B / t_un, "ms_per_forward_one_at_a_time": t_un * 1e3,
                       "note": "value_one_at_a_time: the same forwards issued one after the other (two half-batch lanes inside each), "
                               "this rank's rate x ranks; outputs of the two modes are compared bit for bit in this run"},
        "stage_ms": {"equi2pers_P128": t_e2p * 1e3, "network": t_net * 1e3, "pers2equi_conf_P128": t_p2e * 1e3,
                     "note": "one forward at a time (stage events on the launch stream); in the timed region forwards overlap"},
        "host_fed": {"panoramas_per_s_per_gpu": host_fed, "frac_of_resident": host_fed / (B * args.steps / dt), "batches": nfeed,
                     "h2d_GBps": h2d_gbps, "h2d_GBps_needed": host_fed * ERP_H * ERP_W * 3 / 1e9, "png_fed": png_fed,
                     "note": "inputs arrive as decoded uint8 BGR frames in pinned host memory (1.5 MB per panorama over PCIe), H2D + /255 + "
                             "HWC->CHW on a side stream, triple-buffered (omnifusion_amd/data.py DeviceFeeder); steady state: timed after the first batches have been retired; h2d_GBps = pinned uint8 batches copied back to back on this box"},
        "batch1": {"ms_per_forward": b1_ms, "panoramas_per_s": 1e3 / b1_ms, "graph_ms": b1_graph_ms,
                   "note": "BASELINE cfg 2 literally: one 512x1024 panorama per forward on this GPU (latency of ~135 dependent launches)",
                   "stream_of_requests": {"ms_per_forward": b1_stream_ms, "panoramas_per_s": 1e3 / b1_stream_ms,
                                          "note": "the same single-panorama forwards, 4 in flight on 4 streams, one hipGraph replay each "
                                                  "(spherical_fusion.pipelined(4, graphs=True)); outputs compared bit for bit"}},
        # headline fraction = fp16 MFMA flops ISSUED / the fp16 dense peak (2500 TFLOP/s, MI355X_MICROARCH.md).  The f16x3 scheme issues three
        # fp16 MFMAs per product block for fp32-class accuracy (1e-3 abs on depth needs it: profiles/r02b_precision_map.txt), so the ALGORITHMIC
        # (fp32-equivalent) rate is a third of that: `frac_algorithmic`; against the exact-fp32 MFMA peak (157 TFLOP/s) it is `x_fp32_mfma_peak`.
        "roofline": {"bound": "mfma",
                     "kernel": ("network section (conv_sh_kernel / conv3x3_halo_sh_kernel dominant" if f16x3 else
                                "network section (conv_igemm_f32_kernel<...> dominant") + "; includes the stem/pool/upsample/LN/"
                               "attention/heads launches)",
                     # achieved = ALGORITHMIC flops (71.3 GFLOP per panorama) / time; `frac` = that against the dense fp16 MFMA peak
                     "achieved": tflops, "peak": MFMA_F16_PEAK_TFLOPS if f16x3 else MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tflops / (MFMA_F16_PEAK_TFLOPS if f16x3 else MFMA_F32_PEAK_TFLOPS),
                     "frac_algorithmic": tflops / MFMA_F16_PEAK_TFLOPS,
                     "achieved_issued": (3 * tflops if f16x3 else tflops), "frac_issued": (3 * tflops if f16x3 else tflops) / MFMA_F16_PEAK_TFLOPS,
                     "x_fp32_mfma_peak": tflops / MFMA_F32_PEAK_TFLOPS,
                     "traffic": net_traffic, "traffic_note": net_traffic_note,
                     "dominant": dominant,
                     "kernel_time_shares": (shares or {}).get("shares"), "kernel_time_shares_note": shares_note or (shares or {}).get("note", ""),
                     "flops_per_step": NET_GFLOP_PER_PANO * B * 1e9,
                     "achieved_network_section_alone": NET_GFLOP_PER_PANO * B / t_net / 1e3,
                     "note": ("`achieved` / `frac` = ALGORITHMIC flops (71.3 GFLOP per panorama x panoramas / time over the whole timed region, all launches of the "
                              "steps) against the fp16 dense peak; `achieved_issued` / `frac_issued` = fp16 MFMA flops ISSUED: the f16x3 scheme issues three fp16 MFMAs "
                              "per product block for fp32-class accuracy (rounds 1-4 printed the issued figure as `frac`).  With every CU issuing MFMAs the chip sustains "
                              "1.5-1.75 GHz, not 2.4 (tools/dbg_mfma.py: 18-22 ns per 32x32x16 MFMA per SIMD chip-wide vs 13.5 ns on one CU): the "
                              "reachable ceiling is ~0.7 of `peak` issued, ~0.23 algorithmic") if f16x3 else
                             "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)"},
        "roofline_resample": {"bound": "hbm", "kernel": "e2p_box_kernel<float,2,false> + p2e_lds_kernel<float,8,false,2> at 18x256^2, B=%d (planar layout)" % B,
                              # `achieved` / `frac`: the launches over ROTATING buffer sets (> 256 MB in total): nothing is found in the memory-side cache — the rate
                              # from HBM.  `frac_cache_warm`: the same kernels re-launched on one buffer set (input resident in that cache, as inside a
                              # forward where the previous kernel has just written it) — the figure rounds 1-4 printed as `frac`.
                              "achieved": (bytes_e2p + bytes_p2e) / (rr_e2p + rr_p2e) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": (bytes_e2p + bytes_p2e) / (rr_e2p + rr_p2e) / 1e9 / HBM_PEAK_GBS,
                              "achieved_cache_warm": gbs_pair, "frac_cache_warm": gbs_pair / HBM_PEAK_GBS,
                              # HBM bytes per launch pair from rocprofv3 PMC (separate --pmc passes; FETCH_SIZE doubled as
                              # MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as reported), written by tools/pmc_traffic.sh
                              # together with the hash of the sources it was measured on: null when that is not THIS build
                              "traffic": traffic, "traffic_note": traffic_note,
                              "two_streams": {"us_per_pair": t_pair * 1e6, "GB/s": (bytes_e2p + bytes_p2e) / t_pair / 1e9,
                                              "frac": (bytes_e2p + bytes_p2e) / t_pair / 1e9 / HBM_PEAK_GBS,
                                              "note": "the two operators on two streams (as consecutive pipelined forwards run them), one buffer set; "
                                                      "`achieved` / `frac` above are the strict figures: one launch after the other, HIP events per kernel, rotating buffers"},
                              "rotating_buffers": {"equi2pers_us": rr_e2p * 1e6, "pers2equi_us": rr_p2e * 1e6,
                                                   "note": "3 (equi2pers) / 6 (pers2equi) rotating input+output buffer sets: the `achieved` / `frac` above"},
                              "cache_warm": {"equi2pers_us": r_e2p * 1e6, "pers2equi_us": r_p2e * 1e6, "note": "one buffer set re-launched: `frac_cache_warm`"},
                              "equi2pers": {"us": r_e2p * 1e6, "bytes": bytes_e2p, "GB/s": bytes_e2p / r_e2p / 1e9},
                              "pers2equi": {"us": r_p2e * 1e6, "bytes": bytes_p2e, "GB/s": bytes_p2e / r_p2e / 1e9},
                              "method": "one HIP event pair around 20+ back-to-back launches of the kernel, queued behind device-side spinning (agrees with rocprofv3 --kernel-trace: profiles/r03b_event_method.txt); mean of 3 such runs; measured before the matrix-bound steps"},
        "configs": configs,
    }
    if rank == 0:
        out["cpu_baseline"] = None if args.no_cpu_baseline or world > 1 else cpu_baseline()
        print(json.dumps(out))
    dist.finalize()


if __name__ == "__main__":
    main()
