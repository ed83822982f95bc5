"""GPU parity of the host-facing ends (csrc/omni_io.hip through the Python mirrors) against the reference's own outputs (G12, G13)
and the oracle restatement (oracle/io_ref.py)."""
import os

import numpy as np
import pytest
import torch

from _util import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_berhu_loss_and_gradient_golden():
    from omnifusion_amd.supervision.direct import calculate_berhu_loss
    g = golden("G12_berhu")
    pred = torch.from_numpy(g["pred"]).to(DEV).requires_grad_(True)
    loss = calculate_berhu_loss(pred, torch.from_numpy(g["gt"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV), torch.from_numpy(g["weights"]).to(DEV))
    assert loss.shape == () and abs(loss.item() - float(g["loss"])) <= 2e-6
    (3.0 * loss).backward()
    assert np.abs(pred.grad.cpu().numpy() - 3.0 * g["grad"]).max() <= 1e-7
    # a larger, batch-8 case against the oracle (two-stage sums are deterministic: same bits on repeat)
    from oracle import io_ref
    rng = np.random.default_rng(5)
    gt = rng.uniform(0, 8, (8, 1, 512, 1024)).astype(np.float32); pr = (gt + rng.normal(0, 1, gt.shape)).astype(np.float32)
    mk = (rng.random(gt.shape) < 0.6).astype(np.float32); wt = rng.uniform(0.5, 2, gt.shape).astype(np.float32)
    a = calculate_berhu_loss(*(torch.from_numpy(v).to(DEV) for v in (pr, gt, mk, wt)))
    b = calculate_berhu_loss(*(torch.from_numpy(v).to(DEV) for v in (pr, gt, mk, wt)))
    ref, _ = io_ref.berhu_loss(pr, gt, mk, wt)
    assert a.item() == b.item() and abs(a.item() - float(ref)) <= 1e-5 * abs(float(ref))


def test_pointcloud_and_ply_golden(tmp_path):
    from omnifusion_amd.ply import depth_to_pointcloud, write_ply_pointcloud
    g = golden("G13_pointcloud")
    d, c = torch.from_numpy(g["depth"]).to(DEV), torch.from_numpy(g["rgb"]).to(DEV)
    pts, col = depth_to_pointcloud(d, c)
    assert pts.shape == (2, 512, 3) and np.abs(pts.cpu().numpy() - g["pts"]).max() <= 2e-6 and np.array_equal(col.cpu().numpy(), g["col"])
    f = write_ply_pointcloud(str(tmp_path / "pred_0"), d, c)
    mine, ref = open(f, "rb").read(), g["ply0"].tobytes()
    hm, bm = mine.split(b"end_header\n"); hr, br = ref.split(b"end_header\n")
    assert hm == hr                                                          # the header ply.write_ply writes, byte for byte
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("blue", "u1"), ("green", "u1"), ("red", "u1")])
    a, b = np.frombuffer(bm, dt), np.frombuffer(br, dt)
    assert a.shape == b.shape and all(np.array_equal(a[k], b[k]) for k in ("blue", "green", "red"))
    assert max(np.abs(a[k] - b[k]).max() for k in ("x", "y", "z")) <= 2e-6


@pytest.mark.parametrize("src,dst", [((64, 128), (64, 128)), ((128, 256), (64, 128)), ((256, 512), (64, 128)), ((90, 200), (32, 64))])
def test_preprocess_vs_restatement_INTER_AREA_UNPINNED(src, dst):
    """HIP prep kernels against oracle/io_ref.py — whose INTER_AREA is a restatement of OpenCV's published algorithm that NO cv2 output pins
    (cv2 is absent from this image).  /255, HWC->CHW, depth scaling and the (0.1, 8] mask follow dataset_loader_stanford.py:54,76-109 exactly;
    the resize is 'equal to the restatement', not 'equal to the reference' (keep out of green parity claims)."""
    from omnifusion_amd.data import preprocess_rgb, preprocess_depth
    from oracle import io_ref
    rng = np.random.default_rng(7)
    fr = rng.integers(0, 256, (2,) + src + (3,), dtype=np.uint8)
    got = preprocess_rgb(torch.from_numpy(fr).to(DEV), dst).cpu().numpy()
    ref = io_ref.preprocess_rgb(fr, *dst)
    d = np.abs(got - ref)
    # one uint8 step (1/255) where the float average sits within round-off of x.5 (fractional scales only)
    assert d.max() <= (0.0 if src[0] % dst[0] == 0 else 1.0 / 255 + 1e-6) and (d > 1e-6).mean() <= 1e-3
    dz = rng.integers(0, 6000, (2,) + src, dtype=np.uint16)
    gd, gm = preprocess_depth(torch.from_numpy(dz.view(np.int16)).to(DEV), dst)
    rd, rm = io_ref.preprocess_depth(dz, *dst)
    flip = gm.cpu().numpy() != rm
    assert flip.mean() <= 1e-3 and np.abs(gd.cpu().numpy() - rd)[~flip].max() <= 1e-4


def test_device_feeder_delivers_every_batch_in_order():
    from omnifusion_amd.data import DeviceFeeder
    from oracle import io_ref
    rng = np.random.default_rng(9)
    batches = [rng.integers(0, 256, (2, 64, 128, 3), dtype=np.uint8) for _ in range(7)]
    outs = []
    for rgb in DeviceFeeder(batches, (32, 64), depth=3):
        assert rgb.is_cuda and rgb.shape == (2, 3, 32, 64)
        outs.append(((rgb * 2.0).sum(), rgb.clone()))                        # consume on the current stream (the buffer is reused: clone to keep)
    assert len(outs) == 7
    for (_, got), fr in zip(outs, batches):
        assert np.array_equal(got.cpu().numpy(), io_ref.preprocess_rgb(fr, 32, 64))


def test_device_feeder_with_a_consumer_on_other_streams():
    """`out_buffers=2` + `done_with(rgb, pending.input_read)`: pipelined forwards read the staging buffers on their own streams —
    every batch must reach its forward unmodified (same depth as a plain call on a private copy)"""
    from omnifusion_amd.data import DeviceFeeder, preprocess_rgb
    from omnifusion_amd.model.spherical_model import spherical_fusion
    from omnifusion_amd.weights import make_state_dict
    net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda()
    net.load_state_dict(make_state_dict(42, 18, False))
    rng = np.random.default_rng(10)
    batches = [torch.from_numpy(rng.integers(0, 256, (2, 128, 256, 3), dtype=np.uint8)).pin_memory() for _ in range(9)]
    want = [net(preprocess_rgb(b.cuda(), (128, 256))).clone() for b in batches]
    run = net.pipelined(3)
    feeder = DeviceFeeder(batches, (128, 256), depth=3, out_buffers=2)
    pend = []
    for rgb in feeder:
        p = run(rgb)
        feeder.done_with(rgb, p.input_read)
        pend.append(p)
    got = [p.get() for p in pend]
    torch.cuda.synchronize()
    diffs = [float((g - w).abs().max()) for g, w in zip(got, want)]
    assert len(got) == 9 and all(torch.equal(g, w) for g, w in zip(got, want)), f"max |d| per batch: {diffs}"


@pytest.mark.parametrize("extra", [[], ["--iterative", "--iter", "2"], ["--depth", "1", "--src-scale", "2"]])
def test_eval_harness_runs_end_to_end(tmp_path, extra):
    """tools/eval.py (the loop of test.py:193-258: loader -> forward -> on-device metrics -> PLY) on synthetic Stanford2D3D-shaped frames:
    finishes, prints the seven averages and writes a point cloud."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "eval.py"), "--batches", "3", "--batch", "2", "--height", "128", "--width", "256",
           "--ply-every", "2", "--out", str(tmp_path)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Avg. Abs. Rel. Error" in r.stdout and "panoramas/s" in r.stdout, r.stdout[-1000:]
    assert any(f.endswith(".ply") for _, _, fs in os.walk(tmp_path) for f in fs)


def test_sharded_eval_gives_the_unsharded_averages(tmp_path):
    """test.py:161 scales by ONE median over the batch; a per-shard median gave a sharded run other averages than the
    single-GPU one although its depth maps were the same bits.  tools/eval.py now gathers every batch before metering it: the printed
    averages of a 2-rank run (two gloo ranks sharing this GPU, the hook test_bench_gpu.py uses) equal the 1-rank run's, digit for digit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "tools", "eval.py"), "--batches", "2", "--batch", "4", "--height", "128", "--width", "256",
            "--ply-every", "0", "--out", str(tmp_path)]
    def averages(cmd, env):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return [ln.strip() for ln in r.stdout.splitlines() if ln.strip().startswith(("Avg.", "Inlier"))]
    one = averages(base, dict(os.environ))
    two = averages(base + ["--gpus", "2"], dict(os.environ, OMNI_BENCH_DIST_BACKEND="gloo"))
    assert len(one) >= 7 and one == two, (one, two)


def test_png_files_through_the_feeder(tmp_path):
    """SURVEY 8f rank 2 end to end: PNG files -> native decode on host threads into pinned memory (omnifusion_amd/png.py) -> DeviceFeeder
    (async H2D) -> /255 + HWC->CHW on the device == the loader's arithmetic on the source frames (dataset_loader_stanford.py:54,85,92-97)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_png import encode_png
    from omnifusion_amd import png
    from omnifusion_amd.data import DeviceFeeder
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (64, 128, 3), dtype=np.uint8) for _ in range(10)]          # file order R,G,B
    paths = []
    for k, f in enumerate(frames):
        pth = tmp_path / f"pano_{k}.png"
        pth.write_bytes(encode_png(f, 2, 8, idat=2))
        paths.append(str(pth))
    got = []
    for rgb in DeviceFeeder(png.PngBatches(paths, 4, threads=2), (64, 128), device=DEV):
        got.append(rgb.clone())
    got = torch.cat(got).cpu().numpy()
    want = np.stack([f[:, :, ::-1].astype(np.float32).transpose(2, 0, 1) / 255 for f in frames])   # cv2.imread -> /255 -> CHW
    assert got.shape == (10, 3, 64, 128) and np.abs(got - want).max() <= 1e-7
