"""Footprint statistics for an LDS-staged pers2equi: for every ERP tile (TH x TW) and covering patch, the bounding box of the
bilinear taps inside the patch (16-byte aligned columns).  Uses the CPU oracle's tap tables.  Design aid only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle as co

def stats(H, W, nrows, P, TH, TW, verbose=False):
    t = co.pers2equi_tables((80, 80), nrows, (P, P), (H, W))
    N = t["x0"].shape[0]
    w = t["w_list"] * t["mask"][..., None]
    used = (w > 1e-5).any(-1)                      # (pixel, patch) pairs that contribute
    tot_box = 0; nbox = 0; mx = 0; tot_fp = 0; sizes = []
    maxc = 0
    for i0 in range(0, H, TH):
        for j0 in range(0, W, TW):
            c = 0
            for n in range(N):
                u = used[n, i0:i0+TH, j0:j0+TW]
                if not u.any(): continue
                c += 1
                x0 = t["x0"][n, i0:i0+TH, j0:j0+TW][u]; x1 = t["x1"][n, i0:i0+TH, j0:j0+TW][u]
                y0 = t["y0"][n, i0:i0+TH, j0:j0+TW][u]; y1 = t["y1"][n, i0:i0+TH, j0:j0+TW][u]
                xa = (x0.min() // 4) * 4; xb = (x1.max() // 4) * 4 + 4
                bw = xb - xa; bh = y1.max() - y0.min() + 1
                a = bw * bh
                tot_box += a; nbox += 1; mx = max(mx, a); sizes.append(a)
                tot_fp += u.sum()
            maxc = max(maxc, c)
    sizes = np.array(sizes)
    print(f"{H}x{W} nrows={nrows} P={P} tile {TH}x{TW}: tiles={H//TH*(W//TW)} boxes={nbox} ({nbox/(H//TH*(W//TW)):.2f}/tile, max {maxc}) "
          f"mean box {sizes.mean()*4/1024:.1f} KB p99 {np.percentile(sizes,99)*4/1024:.1f} max {mx*4/1024:.1f} KB | staged/patch-data = {tot_box/(N*P*P):.2f} "
          f"| pairs/patchpx = {tot_fp/(N*P*P):.2f}")

if __name__ == "__main__":
    H, W, nrows, P = (int(v) for v in sys.argv[1:5])
    for th, tw in [(16, 64), (32, 32), (8, 64), (16, 32), (32, 64), (8, 128), (16,128)]:
        stats(H, W, nrows, P, th, tw)
