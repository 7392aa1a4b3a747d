"""Where the big cells of the pile-up regime are and how their particles are spread inside them.
   python scripts/pile_geom.py [tick]"""
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
cx = np.floor(pos[:, 0] / d).astype(np.int64); cy = np.floor(pos[:, 1] / d).astype(np.int64)
ncol = cx.max() - cx.min() + 3
key = (cy - cy.min()) * ncol + (cx - cx.min())
cnt = np.bincount(key)
big = np.flatnonzero(cnt > 1000)
print("cells > 1000:", len(big), " rows of big cells:", np.unique(big // ncol)[:20], "... cols:", np.unique(big % ncol)[:20])
for k in big[np.argsort(cnt[big])[-12:]]:
    m = key == k
    q = pos[m]
    fy = (q[:, 1] / d) % 1.0; fx = (q[:, 0] / d) % 1.0
    print(f"cell row {k // ncol + cy.min()} col {k % ncol + cx.min()}: {m.sum()} particles  x in cell [{fx.min():.3f},{fx.max():.3f}] 5-95% [{np.percentile(fx,5):.3f},{np.percentile(fx,95):.3f}]  y in cell [{fy.min():.3f},{fy.max():.3f}] 5-95% [{np.percentile(fy,5):.3f},{np.percentile(fy,95):.3f}]  distinct x {len(np.unique(q[:,0]))} distinct y {len(np.unique(q[:,1]))}")
