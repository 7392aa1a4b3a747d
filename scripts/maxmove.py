"""How many cells the fastest particles of the contract workload cross per tick (the halo overlap's band margin).
   python scripts/maxmove.py"""
import copy, sys
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import bench, sand_crate_amd as sc
n = 524288
wc, d = bench.world_for(n)
p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
s.particles = p; s.particle_velocities = v
prev = None
for t in range(45):
    s.run(1); s.synchronize()
    pos, vel, _, ids = s.engine.download()
    order = np.argsort(ids); pos = pos[order]; ids_s = ids[order]
    if prev is not None and len(prev[0]) == len(pos) and np.array_equal(prev[1], ids_s):
        dc = np.abs(np.floor(pos / d) - np.floor(prev[0] / d))
        if t > 28:
            k = int(np.argmax(dc[:, 1]))
            print(f"tick {t+1}: max column move {int(dc[:,0].max())}, max row move {int(dc[:,1].max())}; particles moving >= 2 rows: {(dc[:,1] >= 2).sum()}; the fastest in y: from {prev[0][k]} to {pos[k]} (d = {d:.5f})")
    prev = (pos, ids_s)
