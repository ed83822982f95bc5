#!/usr/bin/env python
"""Evaluation harness — the loop of /root/reference/test.py:193-203,244-258 on the MI355X path.

    python tools/eval.py [--checkpoint ckpt.pth] [--iterative --iter 2] [--batches 8 --batch 4] [--ply-every 20 --out results/]
                         [--from-host]  [--gpus N]

    network = spherical_fusion(...); network.load_state_dict(ckpt); network.cuda(); network.eval()          test.py:104-111
    for rgb, depth, mask in loader:                                                                         test.py:195
        equi_outputs = network(rgb[, iter])[-1]                                                             test.py:198-199
        compute_eval_metrics(equi_outputs, depth, mask)        # median scaling + 7 metrics, on the device  test.py:203
        every --ply-every batches: point cloud of item 0 -> PLY                                             test.py:210-240
    print the seven averages                                                                                test.py:244-258

There is no dataset in this image (Stanford2D3D is not redistributable and there is no network): without --data the loader yields
Stanford2D3D-SHAPED synthetic frames — uint8 BGR 512x1024 (or larger with --src-scale) and 16-bit depth — through the same device
pipeline a real loader would feed (omnifusion_amd/data.py: pinned H2D + INTER_AREA + /255 on a side stream); weights are the
deterministic random-init generator unless --checkpoint names a reference state_dict.  With --gpus N the batches are sharded by
image over N ranks (omnifusion_amd/dist.py); every rank gathers the batch before metering it (ONE median per batch, as test.py:161).
"""
import argparse
import collections
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def synthetic_batches(n, B, H, W, scale, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        rgb = rng.integers(0, 256, (B, H * scale, W * scale, 3), dtype=np.uint8)
        depth = rng.integers(60, 4090, (B, H * scale, W * scale), dtype=np.uint16)      # 0.12 .. 7.99 m
        yield rgb, depth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--iterative", action="store_true"); ap.add_argument("--iter", type=int, default=2)
    ap.add_argument("--nrows", type=int, default=4); ap.add_argument("--patchsize", type=int, default=128); ap.add_argument("--fov", type=float, default=80.0)
    ap.add_argument("--batches", type=int, default=8); ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=512); ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--src-scale", type=int, default=1, help="decoded frames are this many times larger than the network input (INTER_AREA on the device)")
    ap.add_argument("--ply-every", type=int, default=0); ap.add_argument("--out", default="results")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--depth", type=int, default=3, help="forwards in flight (spherical_fusion.pipelined); 1 = the loop of test.py as written")
    args = ap.parse_args()
    from omnifusion_amd import dist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(dist.respawn_under_launcher(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    assert torch.cuda.is_available(), "tools/eval.py needs an MI355X (no CPU fallback exists)"
    rank, local, world, dev = dist.init(os.environ.get("OMNI_BENCH_DIST_BACKEND", "nccl"))

    from omnifusion_amd.data import DeviceFeeder, preprocess_depth
    from omnifusion_amd.eval import DepthMetrics
    from omnifusion_amd.ply import write_ply_pointcloud
    from omnifusion_amd.weights import make_state_dict
    N = {3: 10, 4: 18, 5: 26, 6: 46}[args.nrows]
    if args.iterative:
        from omnifusion_amd.model.spherical_model_iterative import spherical_fusion
    else:
        from omnifusion_amd.model.spherical_model import spherical_fusion
    network = spherical_fusion(args.nrows, N, (args.patchsize, args.patchsize), (args.fov, args.fov))        # test.py:104
    if world > 1:                                              # a shard of ONE panorama must give the bits of the unsharded run
        from omnifusion_amd.model._engine import Engine
        Engine.latency_plan = False
    sd = torch.load(args.checkpoint, map_location="cpu") if (args.checkpoint and rank == 0) else (make_state_dict(42, N, args.iterative) if rank == 0 else None)
    sd = dist.broadcast_state_dict(sd, src=0)                                                                # one reader, RCCL broadcast
    network.load_state_dict(sd)
    network.cuda(dev.index); network.eval()                                                                  # test.py:110-111,194

    lo, hi = dist.shard(args.batch, rank, world)                                                             # image sharding of every batch
    frames = [(r[lo:hi], d[lo:hi]) for r, d in synthetic_batches(args.batches, args.batch, args.height, args.width, args.src_scale, 1234)]
    meters = DepthMetrics()
    os.makedirs(args.out, exist_ok=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    depth = max(1, args.depth)
    feeder = DeviceFeeder((r for r, _ in frames), (args.height, args.width), device=dev, out_buffers=2 if depth > 1 else 1)
    run = network.pipelined(depth)

    def finish(batch_idx, rgb, gt, pending):                                                                 # the part of the loop after the forward
        depth_gt, mask = gt
        out = pending.get()
        out = out[-1] if args.iterative else out
        if args.ply_every and batch_idx % args.ply_every == 0 and rank == 0:
            write_ply_pointcloud(os.path.join(args.out, f"test_pred_{batch_idx}"), out, rgb)                 # test.py:233-238 (before the in-place scaling)
        if world > 1:
            # test.py:161 takes ONE median scaling factor over the whole batch (batch_size = 2): a per-shard median gives other averages than
            # the single-GPU run.  The shards' depth maps, ground truth and masks are gathered (one all_gather of equal blocks each: teardown
            # traffic of a batch, 2 MB per panorama) and every rank meters the WHOLE batch — the averages then need no reduction at all.
            out, depth_gt, mask = (dist.gather_batch(t_.contiguous(), args.batch) for t_ in (out, depth_gt, mask))
        meters.update(out, depth_gt, mask.to(torch.float32))                                                 # test.py:203

    inflight = collections.deque()
    for batch_idx, rgb in enumerate(feeder):
        d16 = torch.from_numpy(frames[batch_idx][1].view(np.int16)).to(dev, non_blocking=True)
        gt = preprocess_depth(d16, (args.height, args.width))                                                # loader :76-80,99-109
        pending = run(rgb, iter=args.iter) if args.iterative else run(rgb)                                   # test.py:198-199, `depth` batches in flight
        feeder.done_with(rgb, pending.input_read) if depth > 1 else None
        want_ply = args.ply_every and batch_idx % args.ply_every == 0 and rank == 0
        inflight.append((batch_idx, rgb.clone() if (want_ply and depth > 1) else rgb, gt, pending))        # (the feeder recycles rgb once equi2pers has read it)
        if len(inflight) >= depth:
            finish(*inflight.popleft())
    while inflight:
        finish(*inflight.popleft())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    avg = meters.averages()                                                                                  # (sharded runs metered whole batches on every rank, see finish())
    if rank == 0:
        print('  Avg. Abs. Rel. Error: {:.4f}\n  Avg. Sq. Rel. Error: {:.4f}\n  Avg. Lin. RMS Error: {:.4f}\n  Avg. Log RMS Error: {:.4f}\n'
              '  Inlier D1: {:.4f}\n  Inlier D2: {:.4f}\n  Inlier D3: {:.4f}\n'.format(
                  avg["abs_rel"], avg["sq_rel"], math.sqrt(avg["rms_sq_lin"]), math.sqrt(avg["rms_sq_log"]), avg["d1"], avg["d2"], avg["d3"]))
        print(f"{args.batches * args.batch} panoramas in {dt:.3f} s ({args.batches * args.batch / dt:.1f} panoramas/s incl. host staging, {world} rank(s))")
        if network.overflowed():
            print("WARNING: activations left the fp16 range of the f16x3 format (saturated): rerun with OMNI_NET_PRECISION=fp32")
    dist.finalize()


if __name__ == "__main__":
    main()
