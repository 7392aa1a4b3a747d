"""Where physics_tick() spends the host's time on config/wave_machine.yaml (cProfile over 300 ticks), and the GPU's
(kernel events).   python scripts/drop_in_profile2.py"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import sand_crate_amd as sc
crate = sc.Crate(sc.load_config("config/wave_machine.yaml").world_config, noise="host")
for _ in range(50):
    crate.physics_tick()
crate.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    crate.physics_tick()
pr.disable()
crate.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(22)
t0 = time.perf_counter()
for _ in range(300):
    crate.physics_tick()
t1 = time.perf_counter()
crate.synchronize()
t2 = time.perf_counter()
print(f"300 ticks: {1e6 * (t1 - t0) / 300:.1f} us per tick until the calls returned, {1e6 * (t2 - t0) / 300:.1f} us per tick synchronised")
eng = crate.engine
eng.reset_timing(); eng.enable_timing(True)
for _ in range(100):
    crate.physics_tick()
crate.synchronize()
tm = {k: (round(1000 * ms / c, 1), c) for k, (ms, c) in eng.timing().items() if c}
print("kernel events, us per launch and launches over 100 ticks:", tm)
print("sum per tick: %.1f us" % sum(1000 * ms / 100 for ms, c in eng.timing().values()))
