import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import sand_crate_amd as sc
from test_gpu_parity import synthetic, wave_world
from oracle.scene import OracleCrate
from oracle.tick import counter_noise_key, counter_noise_u01, tick_core
from oracle.world import World
n=20000; noise="counter"
p, v, d = synthetic(n, seed=n, margin=0.0, vel=30.0)
wc = wave_world(sc, d, 0.1); wc.coefficients["max_particles"] = n
crate = sc.Crate(wc, noise=noise, noise_seed=77)
crate.particles = p; crate.particle_velocities = v
orc = OracleCrate(World(wc.rigid_bodies, [], dict(wc.coefficients)))
ids = np.arange(n); op, ov = p.copy(), v.copy()
for t in range(3):
    crate.physics_tick()
    for b in orc.rigid_bodies: b.advance(orc.coef["dt"])
    eta = counter_noise_u01(ids, counter_noise_key(77, t))
    out = tick_core(op, ov, orc.segments, orc.body_states(), orc.coef, eta_u01=eta)
    gp, gv = crate.particles, crate.particle_velocities
    print("tick", t, "count", len(gp), len(out["particles"]))
    if len(gp) != len(out["particles"]): break
    err = np.abs(gv - out["velocities"]).max(1)
    bad = np.flatnonzero(err > 1e-9)
    print(" bad", len(bad), "pressure err", np.abs(crate.particles_pressure-out["pressure"]).max())
    for i in bad[:8]:
        print("  i", i, "gpu v", gv[i], "orc v", out["velocities"][i], "V", out["wall_count"][i], "ccd", out["ccd_factor"][i],
              "C", out["neighbor_counts"][i], "pos", op[i], "v_bounce", out["v_after_bounce"][i], "v_visc", out["v_after_viscosity"][i])
    op, ov = out["particles"], out["velocities"]
    if len(bad): break
