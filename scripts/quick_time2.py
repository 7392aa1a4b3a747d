"""Wall time per tick only (no kernel events): python scripts/quick_time2.py <tag> [particles] [ticks]"""
import sys, time, copy; sys.path.insert(0, ".")
import numpy as np, bench, sand_crate_amd as sc
tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
wc, d = bench.world_for(n); p, v = bench.synthetic_state(n)
res = []
for rep in range(5):
    s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024); s.particles = p; s.particle_velocities = v
    s.run(5); s.synchronize()
    t0 = time.perf_counter(); s.run(k); s.synchronize(); res.append((time.perf_counter() - t0) / k)
    state = s.engine.download()
print(f"{tag:30s} tick {1e6 * np.median(res):7.1f} us (min {1e6 * min(res):.1f} max {1e6 * max(res):.1f})  checksum {float(state[0].sum()):.12f} {float(state[1].sum()):.12f}", flush=True)
