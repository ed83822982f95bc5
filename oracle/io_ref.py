"""TEST INFRASTRUCTURE ONLY — numpy restatement of the host-side ends of the hot path.

  preprocess_rgb / preprocess_depth   /root/reference/dataset_loader_stanford.py:54,76-80,92-109 (cv2.resize INTER_AREA, /255, masks)
  berhu_loss (+ gradient)             /root/reference/supervision/direct.py:3-18
  pointcloud                          /root/reference/test.py:210-240, util.py:159-174

Pinned against the reference itself by oracle/gen_golden_io.py (G12 BerHu incl. autograd, G13 point cloud + the bytes
ply.write_ply produces).  cv2 is an un-vendored third-party dependency (requirements.txt pins no version; the image has
none): `inter_area` restates OpenCV's published algorithm (modules/imgproc/src/resize.cpp: computeResizeAreaTab, and the
integer-scale fast path ResizeAreaFast) — PARITY UNPINNED for that one function.
"""
import numpy as np


def _area_tab(dsize, ssize):
    """OpenCV computeResizeAreaTab: per destination index the (source index, weight) list."""
    scale = ssize / dsize
    tab = []
    for d in range(dsize):
        fs1 = d * scale; fs2 = fs1 + scale
        cell = min(scale, ssize - fs1)
        s1 = int(np.ceil(fs1)); s2 = int(np.floor(fs2))
        s2 = min(s2, ssize - 1); s1 = min(s1, s2)
        e = []
        if s1 - fs1 > 1e-3:
            e.append((s1 - 1, (s1 - fs1) / cell))
        for s in range(s1, s2):
            e.append((s, 1.0 / cell))
        if fs2 - s2 > 1e-3:
            e.append((s2, min(min(fs2 - s2, 1.0), cell) / cell))
        tab.append(e)
    return tab


def inter_area(img, H, W):
    """cv2.resize(img, (W, H), interpolation=cv2.INTER_AREA) for a down-scale; img [Hs,Ws(,C)] uint8 / uint16 / float32.
    Integer dtypes are averaged in float32 and rounded back (saturate_cast: nearest-even; the 2x2 fast path rounds half up)."""
    src = np.asarray(img)
    x = src.astype(np.float32)
    if x.ndim == 2:
        x = x[..., None]
    Hs, Ws, C = x.shape
    ty, tx = _area_tab(H, Hs), _area_tab(W, Ws)
    tmp = np.zeros((Hs, W, C), np.float32)
    for d, e in enumerate(tx):
        for s, w in e:
            tmp[:, d] += x[:, s] * np.float32(w)
    out = np.zeros((H, W, C), np.float32)
    for d, e in enumerate(ty):
        for s, w in e:
            out[d] += tmp[s] * np.float32(w)
    if src.dtype in (np.uint8, np.uint16):
        if Hs == 2 * H and Ws == 2 * W:
            out = np.floor(out + 0.5)
        else:
            out = np.rint(out)
        out = np.clip(out, 0, np.iinfo(src.dtype).max).astype(src.dtype)
    return out[..., 0] if src.ndim == 2 else out


def preprocess_rgb(frames_u8, H, W):
    """[B,Hs,Ws,3] uint8 BGR -> [B,3,H,W] float32 (dataset_loader_stanford.py:54,85,92-97)"""
    out = []
    for f in frames_u8:
        r = f if f.shape[:2] == (H, W) else inter_area(f, H, W)
        out.append((r.astype(np.float32) / np.float32(255)).transpose(2, 0, 1))
    return np.stack(out)


def preprocess_depth(frames_u16, H, W, min_depth=0.1, max_depth=8.0):
    """[B,Hs,Ws] uint16 -> depth [B,1,H,W] float32 (masked), mask [B,1,H,W] uint8 (dataset_loader_stanford.py:76-80,99-109)"""
    ds, ms = [], []
    for f in frames_u16:
        d = f.astype(np.float32)
        if d.shape != (H, W):
            d = inter_area(d, H, W)
        d = d / np.float32(65535) * np.float32(128)
        m = ((d <= max_depth) & (d > min_depth)).astype(np.uint8)
        ds.append((d * m)[None]); ms.append(m[None])
    return np.stack(ds), np.stack(ms)


def berhu_loss(pred, gt, mask, weights):
    """supervision/direct.py:3-18 in float32 numpy; returns (loss, dloss/dpred)."""
    pred = np.asarray(pred, np.float32); gt = np.asarray(gt, np.float32)
    bs = pred.shape[0]
    diff = gt - pred
    ad = np.abs(diff)
    c = np.float32(ad.max() / 5)                                           # :7 (.item() -> a constant)
    leq = ad <= c
    l2 = (diff ** 2 + c ** 2) / (2 * c)
    loss = np.where(leq, ad, l2).reshape(bs, -1)
    m = np.asarray(mask, np.float32).reshape(bs, -1); w = np.asarray(weights, np.float32).reshape(bs, -1)
    count = m.sum(1, keepdims=True)
    val = np.mean((loss * m * w).sum(1, keepdims=True, dtype=np.float64) / count)
    dl = np.where(leq, np.sign(diff), diff / c).reshape(bs, -1)
    grad = -(1.0 / bs) * (m * w / count) * dl
    return np.float32(val), grad.reshape(pred.shape).astype(np.float32)


def pointcloud(depth, rgb):
    """test.py:210-240 for a batch: depth [B,1,H,W], rgb [B,3,H,W] -> xyz [B,H*W,3] float32, colours [B,H*W,3] uint8"""
    B, _, h, w = depth.shape
    coords = np.stack(np.meshgrid(range(w), range(h)), -1).reshape(-1, 2)
    coords = coords + 1
    uv = np.zeros_like(coords, dtype=np.float32)                           # util.py:159-165
    uv[..., 0] = (coords[..., 0] - (w / 2 + 0.5)) / w * 2 * np.pi
    uv[..., 1] = -(coords[..., 1] - (h / 2 + 0.5)) / h * np.pi
    xyz = np.zeros((uv.shape[0], 3), np.float32)                           # util.py:168-173
    xyz[:, 0] = np.cos(uv[:, 1]) * np.sin(uv[:, 0])
    xyz[:, 1] = np.cos(uv[:, 1]) * np.cos(uv[:, 0])
    xyz[:, 2] = np.sin(uv[:, 1])
    pts = xyz[None] * depth.reshape(B, w * h, 1).astype(np.float32)        # test.py:218-219
    col = (rgb.transpose(0, 2, 3, 1).reshape(B, -1, 3) * 255).astype(np.uint8)   # test.py:229,236
    return pts, col
