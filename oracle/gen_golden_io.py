"""G12 (BerHu loss + autograd) and G13 (point cloud + the PLY bytes) from the reference's own code — supervision/direct.py,
util.py:159-174 and ply.py — run in the build container (TEST INFRASTRUCTURE; nothing of the reference travels)."""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OMNI_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def main():
    from oracle import io_ref
    direct = _load("ref_direct", "supervision/direct.py")
    ply = _load("ref_ply", "ply.py")
    rng = np.random.default_rng(12)
    # ---- G12: BerHu
    gt = rng.uniform(0.0, 8.0, (3, 1, 32, 64)).astype(np.float32)
    pred = (gt + rng.normal(0, 0.8, gt.shape)).astype(np.float32)
    pred[1, 0, 3, 5] = gt[1, 0, 3, 5]                                         # an exact hit (sign(0) = 0)
    mask = (rng.random(gt.shape) < 0.7).astype(np.float32)
    wts = rng.uniform(0.5, 1.5, gt.shape).astype(np.float32)
    p = torch.from_numpy(pred.copy()).requires_grad_(True)
    loss = direct.calculate_berhu_loss(p, torch.from_numpy(gt), torch.from_numpy(mask), torch.from_numpy(wts))
    loss.backward()
    rl, rg = io_ref.berhu_loss(pred, gt, mask, wts)
    assert abs(float(loss) - float(rl)) < 1e-6 and np.abs(p.grad.numpy() - rg).max() < 1e-7, (float(loss), float(rl))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "G12_berhu.npz"), pred=pred, gt=gt, mask=mask, weights=wts,
                        loss=np.float32(loss.item()), grad=p.grad.numpy())
    # ---- G13: point cloud.  util.py imports the whole model zoo at module level; its two pure-numpy helpers are exec'd alone
    src = open(os.path.join(REF, "util.py")).read()
    start = src.index("def coords2uv"); end = src.index("def xyz2uv")
    ns = {"np": np}
    exec(compile(src[start:end], "util_helpers", "exec"), ns)                  # executed here only, never stored
    h, w, B = 16, 32, 2
    depth = rng.uniform(0.1, 8.0, (B, 1, h, w)).astype(np.float32)
    rgb = rng.random((B, 3, h, w)).astype(np.float32)
    coords = np.stack(np.meshgrid(range(w), range(h)), -1).reshape(-1, 2); coords += 1          # test.py:211-214
    xyz = ns["uv2xyz"](ns["coords2uv"](coords, w, h))
    pts = (torch.from_numpy(xyz).unsqueeze(0).repeat(B, 1, 1) * torch.from_numpy(depth).reshape(B, w * h, 1)).numpy()   # :217-219
    col = np.reshape(rgb.transpose(0, 2, 3, 1) * 255, (B, -1, 3)).astype(np.uint8)                                   # :229,236
    rp, rc = io_ref.pointcloud(depth, rgb)
    assert np.abs(rp - pts).max() < 1e-6 and np.array_equal(rc, col)
    with tempfile.TemporaryDirectory() as td:
        ply.write_ply(os.path.join(td, "g"), [pts[0], col[0]], ['x', 'y', 'z', 'blue', 'green', 'red'])              # :238
        raw = np.frombuffer(open(os.path.join(td, "g.ply"), "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "G13_pointcloud.npz"), depth=depth, rgb=rgb, pts=pts, col=col, ply0=raw)
    print("G12 loss", float(loss), "| G13 ply bytes", raw.size)


if __name__ == "__main__":
    main()
