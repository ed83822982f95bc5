"""Offline analysis: ERP footprint (bounding box of bilinear taps) of equi2pers tiles."""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import c_oracle as co

def grid(nrows, P, H, W, fov=80.0):
    lam, phi, _ = co.patch_centers(nrows)
    s = np.linspace(0, 1, P)
    x = (2 * s - 1) * np.pi * fov / 360.0; y = (2 * s - 1) * (np.pi / 2) * fov / 180.0
    X, Y = np.meshgrid(x, y)           # [h, w]
    out = []
    for n in range(len(lam)):
        sp, cp = np.sin(phi[n]), np.cos(phi[n])
        inv = 1 / np.sqrt(1 + X * X + Y * Y)
        lat = np.arcsin(np.clip((sp + Y * cp) * inv, -1, 1))
        lon = lam[n] + np.arctan2(X, cp - Y * sp)
        u = lon / np.pi; u = np.where(u > 1, u - 2, u); u = np.where(u < -1, u + 2, u)
        v = lat / (np.pi / 2)
        ix = np.clip((u + 1) * (W - 1) / 2, 0, W - 1); iy = np.clip((v + 1) * (H - 1) / 2, 0, H - 1)
        out.append((np.floor(ix).astype(int), np.floor(iy).astype(int)))
    return out

def analyse(nrows, P, H, W, th, tw, budget):
    g = grid(nrows, P, H, W)
    areas = []; wrapped_areas = []
    for (x0, y0) in g:
        for h in range(0, P, th):
            for w in range(0, P, tw):
                xs = x0[h:h+th, w:w+tw]; ys = y0[h:h+th, w:w+tw]
                rows = ys.max() + 1 - ys.min() + 1
                cols = xs.max() + 1 - xs.min() + 1
                # wrapped: smallest circular interval containing all xs
                u = np.unique(xs.ravel()); gaps = np.diff(np.concatenate([u, [u[0] + W]]))
                colsw = W - gaps.max() + 1 + 1
                areas.append(rows * (cols + 3)); wrapped_areas.append(rows * (min(cols, colsw) + 3))
    a = np.array(areas); aw = np.array(wrapped_areas)
    ns = th * tw
    fit = aw <= budget
    print(f"nrows={nrows} P={P} {H}x{W} tile {th}x{tw}: tiles={len(a)} median box={np.median(aw):.0f} floats "
          f"({np.median(aw)/ns:.2f}/sample) mean(fit)={aw[fit].mean()/ns:.2f}/sample  fit<= {budget}: {fit.mean()*100:.1f}%  "
          f"(no-wrap fit {100*(a<=budget).mean():.1f}%)  p90={np.percentile(aw,90):.0f} max={aw.max()}")

for cfg in ((4, 256, 512, 1024), (6, 256, 1024, 2048), (4, 128, 512, 1024), (6, 512, 2048, 4096)):
    for (th, tw) in ((4, 256), (8, 128), (16, 64), (32, 32), (16, 128), (32, 64)):
        if tw > cfg[1]: continue
        analyse(*cfg, th, tw, 8192)
