import sys, time, copy; sys.path.insert(0, ".")
import numpy as np, bench, sand_crate_amd as sc
tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
wc, d = bench.world_for(n); p, v = bench.synthetic_state(n)
def mk():
    s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n+1024); s.particles = p; s.particle_velocities = v; return s
s = mk(); s.run(5); s.synchronize()
t0 = time.perf_counter(); s.run(60); s.synchronize(); el = (time.perf_counter() - t0) / 60
s = mk(); s.run(5); s.synchronize(); e = s.engine; e.reset_timing(); e.enable_timing(True); s.run(60); s.synchronize(); e.enable_timing(False)
tm = {k: round(1000*ms/c, 1) for k, (ms, c) in e.timing().items() if c}
print(f"{tag:60s} tick {el*1e6:7.1f} us  A {tm.get('neighbors_density')}  B {tm.get('force_integrate')}  AB {tm.get('neighbors_density_force')}  sort {tm.get('wall_bin')}+{tm.get('cell_scan')}+{tm.get('scatter')}+{tm.get('reorder')}", flush=True)
