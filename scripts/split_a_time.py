"""Pass A split into its two halves at full size: the search alone (ENUM launch) and the pair math alone (DENS launch
reading the lists back), via the host-noise path of the C ABI with a zero noise block."""
import copy, sys
sys.path.insert(0, ".")
import numpy as np, bench, sand_crate_amd as sc
from sand_crate_amd import _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
wc, d = bench.world_for(n); p, v = bench.synthetic_state(n)
c = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024); c.particles = p; c.particle_velocities = v
c.run(10); c.synchronize()
p, v, _, _ = c.engine.download()
h = sc.Crate(copy.deepcopy(wc), noise="host-sync", capacity=n + 1024)
e = h.engine
acc = {}
for rep in range(6):
    e.upload(p, v)
    for b in h.rigid_bodies: b.apply_velocity(h.dt)
    h._send_tick_inputs()
    e.synchronize(); e.reset_timing(); e.enable_timing(True)
    e.step_begin()
    st = e.step_stats()
    e.set_noise_host(np.full((st.neighbor_slots, 2), 0.5))
    e.step_finish(); e.synchronize(); e.enable_timing(False)
    for k, (ms, cnt) in e.timing().items():
        if cnt and rep: acc.setdefault(k, []).append(1000 * ms / cnt)
print({k: round(float(np.median(x)), 1) for k, x in acc.items()}, "mean C", st.neighbor_slots / st.particles)
