"""pile-up regime: how many adjacent-row candidates a per-cell y box would let the search skip"""
import copy, sys
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import bench, sand_crate_amd as sc
n = 1048576; T = int(sys.argv[1]) if len(sys.argv) > 1 else 450
wc, d = bench.world_for(n)
p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
s.particles = p; s.particle_velocities = v
s.run(T); s.synchronize()
pos, vel, ids, _ = s.engine.download()
x, y = pos[:, 0], pos[:, 1]
cx = np.floor(x / d).astype(np.int64); cy = np.floor(y / d).astype(np.int64)
cx -= cx.min() - 2; cy -= cy.min() - 2
ncol = cx.max() + 3; nrow = cy.max() + 3
key = cy * ncol + cx
cnt = np.bincount(key, minlength=nrow * ncol)
ymin = np.full(nrow * ncol, np.inf); ymax = np.full(nrow * ncol, -np.inf)
np.minimum.at(ymin, key, y); np.maximum.at(ymax, key, y)
xmin = np.full(nrow * ncol, np.inf); xmax = np.full(nrow * ncol, -np.inf)
np.minimum.at(xmin, key, x); np.maximum.at(xmax, key, x)
big = cnt > 96
print("particles in big cells:", cnt[big].sum(), "cells:", big.sum())
ext = (ymax - ymin)[big] / d
w = cnt[big]
for q in (0.1, 0.25, 0.5, 0.75, 0.9):
    o = np.argsort(ext); cw = np.cumsum(w[o]) / w.sum()
    print(f"  y extent of big cells, particle-weighted quantile {q}: {ext[o][np.searchsorted(cw, q)]:.3f} d")
ext = (xmax - xmin)[big] / d
for q in (0.1, 0.5, 0.9):
    o = np.argsort(ext); cw = np.cumsum(w[o]) / w.sum()
    print(f"  x extent of big cells, particle-weighted quantile {q}: {ext[o][np.searchsorted(cw, q)]:.3f} d")
tot = {+1: 0, -1: 0}; keep = {+1: 0, -1: 0}; keep_xy = {+1: 0, -1: 0}
for dr in (+1, -1):
    for dc in (-1, 0, 1):
        k2 = key + dr * ncol + dc
        c2 = cnt[k2]
        gap_y = np.maximum(np.maximum(ymin[k2] - y, y - ymax[k2]), 0.0)
        gap_x = np.maximum(np.maximum(xmin[k2] - x, x - xmax[k2]), 0.0)
        ok = (c2 > 0) & (gap_y <= d)
        okxy = (c2 > 0) & (gap_y * gap_y + gap_x * gap_x <= d * d)
        tot[dr] += c2.sum(); keep[dr] += c2[ok].sum(); keep_xy[dr] += c2[okxy].sum()
for dr in (+1, -1):
    print(f"row {dr:+d}: candidates (3 cells) {tot[dr]:.3e}, after the cell's y box {keep[dr]:.3e} ({keep[dr]/tot[dr]:.2f}), after its (x, y) box {keep_xy[dr]:.3e} ({keep_xy[dr]/tot[dr]:.2f})")
# where are the big cells
rows = np.unique(np.flatnonzero(big) // ncol, return_counts=True)
print("rows holding big cells (row: cells):", dict(zip(rows[0].tolist()[:12], rows[1].tolist()[:12])), "... of", nrow, "rows")
cols = np.unique(np.flatnonzero(big) % ncol, return_counts=True)
print("cols holding big cells (col: cells):", dict(zip(cols[0].tolist()[:6], cols[1].tolist()[:6])), "...", dict(zip(cols[0].tolist()[-6:], cols[1].tolist()[-6:])), "of", ncol)
sizes = np.sort(cnt[big])[::-1]
print("largest cells:", sizes[:24].tolist())
for lim in (96, 256, 512, 1024, 2048, 3072, 4096, 8192):
    m = cnt > lim
    print(f"cells > {lim}: {m.sum()} holding {cnt[m].sum()} particles")
