"""Cell occupancy of the contract workload along a long run (how many particles sit in cells of hundreds / thousands).
   python scripts/pile_stats.py [particles] [tick,tick,...]"""
import copy, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import bench, sand_crate_amd as sc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
ticks = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100, 200, 300, 500]
wc, d = bench.world_for(n)
p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
s.particles = p; s.particle_velocities = v
done = 0
for T in ticks:
    s.run(T - done); s.synchronize(); done = T
    pos, vel, ids, _ = s.engine.download()
    cx = np.floor(pos[:, 0] / d).astype(np.int64); cy = np.floor(pos[:, 1] / d).astype(np.int64)
    key = (cy - cy.min()) * (cx.max() - cx.min() + 1) + (cx - cx.min())
    cnt = np.bincount(key)
    occ = cnt[cnt > 0]
    print(f"tick {T}: live {len(pos)}  d={d:.5f}  occupied cells {len(occ)}  mean {occ.mean():.2f}  max {occ.max()}  "
          f"cells>24: {np.sum(occ > 24)}  >96: {np.sum(occ > 96)}  >1000: {np.sum(occ > 1000)}  particles in cells>96: {occ[occ > 96].sum()}", flush=True)
    print("   occupancy percentiles (per particle):", np.percentile(np.repeat(occ, occ), [10, 50, 90, 99, 99.9]).round(0))
    print("   y range", pos[:, 1].min(), pos[:, 1].max(), " x range", pos[:, 0].min(), pos[:, 0].max(), " |v| mean", np.linalg.norm(vel, axis=1).mean())
    # exact duplicates of x within a cell (ties)
    top = np.argsort(cnt)[-3:]
    for k in top:
        m = key == k
        xs = pos[m, 0]
        print(f"   cell {k}: {m.sum()} particles, distinct x {len(np.unique(xs))}, distinct (x,y) {len(np.unique(pos[m], axis=0))}")
