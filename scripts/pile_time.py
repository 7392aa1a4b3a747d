"""Wall time per tick of the contract workload in its pile-up regime (ticks 400-499 of a 500-tick run, no events) and
the kernels' HIP-event times over ticks 500-549.   python scripts/pile_time.py [tag] [particles]"""
import copy, sys, time
sys.path.insert(0, ".")
import numpy as np, bench, sand_crate_amd as sc
tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
wc, d = bench.world_for(n); p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024); s.particles = p; s.particle_velocities = v
s.run(400); s.synchronize()
t0 = time.perf_counter(); s.run(100); s.synchronize(); wall = (time.perf_counter() - t0) / 100
e = s.engine
e.reset_timing(); e.enable_timing(True); s.run(50); s.synchronize(); e.enable_timing(False)
tm = {k: round(1000 * ms / 50, 1) for k, (ms, c) in e.timing().items() if c}
print(f"{tag:40s} ticks 400-499: {1e6 * wall:7.1f} us/tick   kernels per tick (ticks 500-549, events): {tm}", flush=True)
