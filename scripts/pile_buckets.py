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
s.run(T); s.synchronize()
pos, vel, ids, _ = s.engine.download()
cx = np.floor(pos[:, 0] / d).astype(np.int64); cy = np.floor(pos[:, 1] / d).astype(np.int64)
key = (cy - cy.min()) * (cx.max() - cx.min() + 1) + (cx - cx.min())
cnt = np.bincount(key)
big = cnt[cnt > 96]
print(f"tick {T}: {len(big)} buckets above 96, {big.sum()} particles in them; tasks of 2048: {np.sum((big + 2047) // 2048)}")
edges = [96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 4096, 8192, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (big > lo) & (big <= hi)
    print(f"   {lo:5d} < size <= {hi:10d}: {m.sum():5d} buckets, {big[m].sum():8d} particles")
