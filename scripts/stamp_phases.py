"""Where a wave of pass A / pass B spends its life (diagnostic build with -DSC_STAMPS): median clock ticks between
the phase stamps of sc_tiled.h, the last tick of a 20-tick run of the contract workload (that tick has no look-ahead:
pass B runs without its fused wall pass -- scripts/stamp_phases_b.py shows both).   python scripts/stamp_phases.py [particles]"""
import copy, ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import bench, sand_crate_amd as sc
from sand_crate_amd import _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
wc, d = bench.world_for(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
p, v = bench.synthetic_state(n)
s.particles = p; s.particle_velocities = v
s.run(20); s.synchronize()
lib = N.load()
buf = np.zeros((6, 1 << 16, 24), dtype=np.int64)
lib.sc_debug_stamps.restype = C.c_int
lib.sc_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sc_debug_stamps(s.engine._ctx, buf.ctypes.data_as(C.c_void_p)) == 0
waves = (n + 63) // 64
names = {0: ["start", "cell+buckets loaded", "tile staged", "scan same-row right", "scan next row", "scan same-row left",
             "scan previous row", "pair math", "lists out"],
         1: ["start", "bounds, lane, table loaded", "tile staged", "pair loop", "velocities staged", "viscosity+finish",
             "next tick's wall pass", "stores"]}
for k, label in ((0, "pass A"), (1, "pass B")):
    st = buf[k if k == 0 else (1 if (buf[1, :waves, 7] > buf[3, :waves, 7]).mean() > 0.5 else 3), :waves, :len(names[k])]
    ok = (st > 0).all(axis=1)
    st = st[ok]
    dt = np.diff(st, axis=1)
    life = st[:, -1] - st[:, 0]
    print(f"{label}: {ok.sum()} waves, median wave life {np.median(life):.0f} ticks; kernel span {(st[:, -1].max() - st[:, 0].min()):.0f} ticks")
    for j, nm in enumerate(names[k][1:]):
        print(f"    {nm:32s} median {np.median(dt[:, j]):8.0f}   mean {dt[:, j].mean():8.0f}   p95 {np.percentile(dt[:, j], 95):8.0f}")

# (the occupancy of the chip over a kernel: scripts/timeline.py on a -DSC_TIMELINE build; pass B with its look-ahead epilogue:
# scripts/stamp_phases_b.py)
