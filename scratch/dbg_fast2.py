import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import sand_crate_amd as sc
from sand_crate_amd import _native as N
from test_gpu_parity import synthetic, wave_world
from oracle.scene import OracleCrate
from oracle.tick import counter_noise_key, counter_noise_u01, tick_core
from oracle.world import World
n=20000
p, v, d = synthetic(n, seed=n, margin=0.0, vel=30.0)
wc = wave_world(sc, d, 0.1); wc.coefficients["max_particles"] = n
crate = sc.Crate(wc, noise="counter", noise_seed=77)
crate.particles = p; crate.particle_velocities = v
orc = OracleCrate(World(wc.rigid_bodies, [], dict(wc.coefficients)))
ids = np.arange(n)
crate.physics_tick()
for b in orc.rigid_bodies: b.advance(orc.coef["dt"])
out = tick_core(p, v, orc.segments, orc.body_states(), orc.coef, eta_u01=counter_noise_u01(ids, counter_noise_key(77, 0)))
op, ov = out["particles"], out["velocities"]
print("tick0 err", np.abs(crate.particles-op).max())
# tick 1 by hand on the crate's engine (carried, cell-sorted storage)
for b in crate.rigid_bodies: b.apply_velocity(crate.dt)
for b in orc.rigid_bodies: b.advance(orc.coef["dt"])
crate._send_tick_inputs()
eng = crate.engine
eng.step_begin()
st = eng.step_stats(); print("stats", st)
gid, cnt, nb, fx = eng.download_neighbors()
inv = np.argsort(gid)
out1 = tick_core(op, ov, orc.segments, orc.body_states(), orc.coef, eta_u01=counter_noise_u01(ids, counter_noise_key(77, 1)))
print("fixed equal", np.array_equal(fx[inv], out1["fixed_positions"]), "counts equal", np.array_equal(cnt[inv], out1["neighbor_counts"]),
      "table equal", np.array_equal(nb[inv], out1["neighbor_table"]))
badrows = np.flatnonzero((nb[inv] != out1["neighbor_table"]).any(1))
print("bad rows", len(badrows), badrows[:10])
for i in badrows[:4]:
    print(i, "pos", out1["fixed_positions"][i], "gpu", nb[inv][i][:cnt[inv][i]], "orc", out1["neighbor_table"][i][:out1["neighbor_counts"][i]])
fdiff = np.flatnonzero((fx[inv] != out1["fixed_positions"]).any(1))
print("fixed diff", len(fdiff), fdiff[:5])
for i in fdiff[:4]:
    print(i, "in", op[i], "gpu", fx[inv][i], "orc", out1["fixed_positions"][i], "V", out1["wall_count"][i])
