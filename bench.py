#!/usr/bin/env python
"""bench.py — headline benchmark of the `equi_pers` hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one pass of the hot path over one per-GPU batch of synthetic
panoramas that are already resident in HBM.  Panoramas shard by image (no data-path collective,
SURVEY.md §8e), so scaling is weak: every rank processes `--batch` panoramas per step.
Prints ONE JSON line on rank 0 (contract: see the task statement / DESIGN.md §Measurement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (≈6.3 TB/s achievable)
ERP_H, ERP_W, NROWS, NPATCH, FOV = 512, 1024, 4, 18, (80.0, 80.0)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="panoramas per GPU per step (8 = BASELINE cfg 4 shard)")
    ap.add_argument("--patch", type=int, default=256, help="resample patch size (BASELINE: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(patch):
    """The C oracle ('port' of the reference algorithm) timed on this host's cores on a
    bounded sample: ONE panorama through equi2pers (C=3) + pers2equi (C=1)."""
    from oracle import c_oracle as co
    co.build()
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    rng = np.random.default_rng(0)
    erp = rng.random((1, 3, ERP_H, ERP_W), dtype=np.float32)
    pin = rng.random((1, 1, patch, patch, NPATCH), dtype=np.float32)
    co.equi2pers(erp[:, :, :64, :128], FOV, NROWS, (16, 16))           # load + warm
    n, t0 = 0, time.perf_counter()
    while True:
        co.equi2pers(erp, FOV, NROWS, (patch, patch))
        co.pers2equi(pin, FOV, NROWS, (patch, patch), (ERP_H, ERP_W))
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 20:
            break
    return {"value": n / dt, "unit": "panoramas/s", "cores": cores, "kind": "port",
            "sample": f"{n} panorama(s) 512x1024 -> 18x{patch}^2 (equi2pers C=3 + pers2equi C=1), "
                      f"C oracle with OpenMP over {cores} threads"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    from omnifusion_amd import _lib
    from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
    from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
    _lib.load()

    B, P = args.batch, args.patch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    erp = torch.rand((B, 3, ERP_H, ERP_W), generator=g).to(dev)               # synthetic RGB panoramas
    depth_patches = torch.rand((B, NPATCH, 1, P, P), generator=g).to(dev)     # synthetic per-patch depth
    LAY = _lib.LAYOUT_BNCHW

    def step():
        p = equi2pers_patches(erp, FOV, NROWS, (P, P), layout=LAY)
        e = pers2equi(depth_patches, FOV, NROWS, (P, P), (ERP_H, ERP_W), None, layout=LAY)
        return p, e

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        p = equi2pers_patches(erp, FOV, NROWS, (P, P), layout=LAY)
        ev[k][1].record()
        e = pers2equi(depth_patches, FOV, NROWS, (P, P), (ERP_H, ERP_W), None, layout=LAY)
        ev[k][2].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())

    t_e2p = float(np.mean([ev[k][0].elapsed_time(ev[k][1]) for k in range(args.steps)])) * 1e-3
    t_p2e = float(np.mean([ev[k][1].elapsed_time(ev[k][2]) for k in range(args.steps)])) * 1e-3
    s = 4
    bytes_e2p = B * 3 * (ERP_H * ERP_W + P * P * NPATCH) * s          # SURVEY §8d algorithmic bytes
    bytes_p2e = B * 1 * (P * P * NPATCH + ERP_H * ERP_W) * s
    gbs_pair = (bytes_e2p + bytes_p2e) / (t_e2p + t_p2e) / 1e9
    out = {
        "metric": "panoramas/sec at 512x1024 ERP, N=18 256^2 patches; equi2pers+pers2equi GB/s vs HBM peak",
        "value": world * B * args.steps / dt, "unit": "panoramas/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"resample pair only (network not in this step yet): {B} panoramas/GPU/step, "
                               f"512x1024 ERP fov 80 nrows 4 -> 18x{P}^2 patches (equi2pers C=3) and back "
                               f"(pers2equi C=1), patch-major layout, inputs resident in HBM",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"image-sharded x{world}"},
        "roofline": {"bound": "hbm", "kernel": "e2p_planar_kernel<float,4> + p2e_kernel<float,8,false>",
                     "achieved": gbs_pair, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_pair / HBM_PEAK_GBS,
                     "traffic": None,
                     "equi2pers": {"us": t_e2p * 1e6, "bytes": bytes_e2p, "GB/s": bytes_e2p / t_e2p / 1e9},
                     "pers2equi": {"us": t_p2e * 1e6, "bytes": bytes_p2e, "GB/s": bytes_p2e / t_p2e / 1e9}},
    }
    if rank == 0:
        out["cpu_baseline"] = None if args.no_cpu_baseline or world > 1 else cpu_baseline(P)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
