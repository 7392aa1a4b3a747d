"""Does a long spin-up change kernel times (DVFS)?"""
import sys, time, json; sys.path.insert(0, ".")
import numpy as np, bench, sand_crate_amd as sc
n = 262144
wc, d = bench.world_for(n); p, v = bench.synthetic_state(n)
def mk():
    import copy
    s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n+1024); s.particles = p; s.particle_velocities = v; return s
def measure(tag, spin):
    if spin:
        t = mk(); t0 = time.time()
        while time.time() - t0 < spin: t.run(20); t.synchronize()
    s = mk(); s.run(5); s.synchronize()
    t0 = time.perf_counter(); s.run(100); s.synchronize(); el = time.perf_counter() - t0
    s = mk(); s.run(5); s.synchronize(); e = s.engine; e.reset_timing(); e.enable_timing(True); s.run(100); s.synchronize(); e.enable_timing(False)
    tm = {k: round(1000*ms/c, 1) for k, (ms, c) in e.timing().items() if c}
    print(tag, "ms/step %.4f" % (el*10), tm, flush=True)
measure("cold", 0); measure("spin2s", 2.0); measure("again", 0)
