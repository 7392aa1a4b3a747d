"""Pile-up regime: the size distribution of the buckets k_sort_big is given (cells above 96 particles) and its tasks.
   python scripts/pile_buckets.py [particles] [tick]"""
import copy, sys
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import bench, sand_crate_amd as sc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
T = int(sys.argv[2]) if len(sys.argv) > 2 else 450
wc, d = bench.world_for(n)
p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
s.particles = p; s.particle_velocities = v
s.run(T - 1); s.synchronize()
pos0, _, _, ids0 = s.engine.download()
s.run(1); s.synchronize()
pos, vel, _, ids = s.engine.download()
cx = np.floor(pos[:, 0] / d).astype(np.int64); cy = np.floor(pos[:, 1] / d).astype(np.int64)
key = (cy - cy.min()) * (cx.max() - cx.min() + 1) + (cx - cx.min())
cnt = np.bincount(key)
big = cnt[cnt > 96]
print(f"tick {T}: {len(big)} buckets above 96, {big.sum()} particles in them; tasks of 2048: {np.sum((big + 2047) // 2048)}")
edges = [96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 4096, 8192, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (big > lo) & (big <= hi)
    print(f"   {lo:5d} < size <= {hi:10d}: {m.sum():5d} buckets, {big[m].sum():8d} particles")

# how many returning atomics the scatter (and the fused count of the force kernel) puts on one address: waves of 64
# particles in storage order, one atomic per distinct cell of a wave (the grouping of a scrambled wave; runs otherwise)
# storage order = the sorted order of the tick that just ran; the cells are those of the positions it left
# (by row, x, id of the positions the tick started from; the wall fix moves a few of them first -- close enough for a count)
sorted_ids = ids0[np.lexsort((ids0, pos0[:, 0], np.floor(pos0[:, 1] / d)))]
sorted_ids = sorted_ids[np.isin(sorted_ids, ids)]
key_by_id = np.empty(ids.max() + 1, dtype=np.int64); key_by_id[ids] = key
key = key_by_id[sorted_ids]
w = (len(key) + 63) // 64
pad = np.full(w * 64, -1, dtype=np.int64); pad[:len(key)] = key
rows = np.sort(pad.reshape(w, 64), axis=1)
distinct = (np.diff(rows, axis=1) != 0).sum(axis=1) + 1
print(f"waves {w}: distinct cells per wave mean {distinct.mean():.1f}, p90 {np.percentile(distinct, 90):.0f}, max {distinct.max()}; waves with more than 12 cells: {(distinct > 12).sum()}")
hot = np.argsort(cnt)[-5:]
for k in hot[::-1]:
    touching = (rows == k).any(axis=1).sum()
    print(f"   cell of {cnt[k]} particles: {touching} waves hold one of them (= atomics on its counter per kernel)")
per_cell = np.bincount(np.unique(np.stack([np.repeat(np.arange(w), 64), pad.reshape(-1)], 1)[pad.reshape(-1) >= 0], axis=0)[:, 1])
print(f"   atomics per counter: max {per_cell.max()}, counters with more than 100: {(per_cell > 100).sum()}, more than 300: {(per_cell > 300).sum()}")
