"""Footprint statistics for LDS-staged equi2pers tiles: for every (patch, TH x TW sample tile) the bounding box of the bilinear
taps on the ERP (16-byte aligned columns, seam-aware).  Uses the CPU oracle's rays.  Design aid only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle as co

def coords(H, W, nrows, P):
    _, xyz, _, _ = co.equi2pers(np.zeros((1, 1, H, W), np.float32), (80, 80), nrows, (P, P), want_pers=False)
    lon = np.arctan2(xyz[:, 0], xyz[:, 1]); lat = np.arcsin(np.clip(xyz[:, 2], -1, 1))
    ix = (lon / np.pi + 1) / 2 * (W - 1); iy = (lat / (np.pi / 2) + 1) / 2 * (H - 1)
    return np.clip(ix, 0, W - 1), np.clip(iy, 0, H - 1)

def stats(H, W, nrows, P, TH, TW, cap_kb, esize=4):
    ix, iy = coords(H, W, nrows, P)
    N = ix.shape[0]
    x0 = np.floor(ix).astype(int); y0 = np.floor(iy).astype(int); y1 = np.minimum(y0 + 1, H - 1)
    epc = 16 // esize
    sizes = []
    for n in range(N):
        for h in range(0, P, TH):
            for w in range(0, P, TW):
                xs = x0[n, h:h+TH, w:w+TW]; ya = y0[n, h:h+TH, w:w+TW].min(); yb = y1[n, h:h+TH, w:w+TW].max()
                d = xs - xs[0, 0]; d = np.where(d >= W // 2, d - W, d); d = np.where(d < -W // 2, d + W, d)
                xa = (xs[0, 0] + d.min()); xb = xs[0, 0] + d.max() + 1
                xa4 = xa // epc * epc
                bw = ((xb - xa4) // epc + 1) * epc
                bw = min(bw, W)
                sizes.append(bw * (yb - ya + 1) * esize)
    sizes = np.array(sizes)
    cap = cap_kb * 1024
    fits = sizes <= cap
    print(f"{H}x{W} nrows={nrows} P={P} tile {TH}x{TW} esize {esize}: tiles={len(sizes)} mean box {sizes[fits].mean()/1024:.2f} KB (fitting) "
          f"p90 {np.percentile(sizes,90)/1024:.1f} p99 {np.percentile(sizes,99)/1024:.1f} max {sizes.max()/1024:.1f} KB | > {cap_kb} KB: {100*(~fits).mean():.1f}% "
          f"| staged/input (fitting tiles) = {sizes[fits].sum()/(H*W*esize):.2f}")

if __name__ == "__main__":
    H, W, nrows, P = (int(v) for v in sys.argv[1:5])
    for th, tw in [(32, 32), (16, 32), (8, 32), (4, 32), (8, 64), (4, 64), (16, 16)]:
        for cap in (4, 8):
            stats(H, W, nrows, P, th, tw, cap)
