import sys; sys.path.insert(0, ".")
import numpy as np, bench, sand_crate_amd as sc, copy
n = 262144
wc, d = bench.world_for(n); p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n+1024); s.particles = p; s.particle_velocities = v
for T in (5, 40, 80, 95, 105, 150):
    s.run(T - s.tick); P = s.particles; V = s.particle_velocities
    cx = np.floor(P[:,0]/d).astype(int); cy = np.floor(P[:,1]/d).astype(int)
    key = cy*100000+cx; u, c = np.unique(key, return_counts=True)
    sp = np.linalg.norm(V, axis=1)
    print(f"tick {T}: P={len(P)} cell occ mean {c.mean():.2f} max {c.max()} p99 {np.percentile(c,99):.0f} cells>=20: {(c>=20).sum()}  |v| mean {sp.mean():.3f} max {sp.max():.2f}  v*dt/d max {sp.max()*s.dt/d:.2f}  pressure mean {s.particles_pressure.mean():.2f}")
    big = u[c>=20][:5]
    for k in big: print("   dense cell row,col", k//100000, k%100000)
    xs = P[:,0]; print("   x==r ties:", (np.abs(xs - d/2) < 1e-12).sum(), " x near right wall:", (np.abs(xs-(1-d/2))<1e-12).sum(), "y near bottom", (np.abs(P[:,1]-(1-d/2))<1e-12).sum(), "y top", (np.abs(P[:,1]-d/2)<1e-12).sum())
