"""Where a tick-by-tick `physics_tick()` of config/wave_machine.yaml spends its time: host wall per tick, kernel time
per tick (HIP events), and the Python share (cProfile).   python scripts/drop_in_profile.py [mode]"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import sand_crate_amd as sc
mode = sys.argv[1] if len(sys.argv) > 1 else "host"
crate = sc.Crate(sc.load_config("config/wave_machine.yaml").world_config, noise=mode)
for _ in range(50):
    crate.physics_tick()
crate.synchronize()
e = crate.engine
e.reset_timing(); e.enable_timing(True)
t0 = time.perf_counter()
for _ in range(200):
    crate.physics_tick()
crate.synchronize()
wall = (time.perf_counter() - t0) / 200
e.enable_timing(False)
tm = e.timing()
print(f"{mode}: wall {1e6 * wall:.0f} us/tick; kernels {sum(ms for ms, c in tm.values()) * 1000 / 200:.0f} us/tick:",
      {k: round(1000 * ms / 200, 1) for k, (ms, c) in tm.items() if c})
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    crate.physics_tick()
crate.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
