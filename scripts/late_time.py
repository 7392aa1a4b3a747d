"""The contract workload beyond the uniform regime: per-kernel times over tick windows.  python scripts/late_time.py [particles]"""
import copy, sys, time
sys.path.insert(0, ".")
import numpy as np, bench, sand_crate_amd as sc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
wc, d = bench.world_for(n); p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024); s.particles = p; s.particle_velocities = v
e = s.engine
from sand_crate_amd._native import NativeError
tick = 0
for span in (5, 20, 20, 20, 20, 20, 20, 20, 20):
    e.reset_timing(); e.enable_timing(True)
    t0 = time.perf_counter()
    s.run(span)
    try:
        s.synchronize()
    except NativeError as err:
        print("flag:", err)
    wall = (time.perf_counter() - t0) / span
    e.enable_timing(False)
    tm = {k: round(1000 * ms / span, 1) for k, (ms, c) in e.timing().items() if c}
    print(f"ticks {tick:3d}..{tick + span - 1:3d}: wall(with events) {1e6 * wall:6.1f} us/tick  per tick: {tm}", flush=True)
    tick += span
