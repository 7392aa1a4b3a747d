"""Where a wave of pass A / pass B spends its life (diagnostic build with -DSC_STAMPS): median clock ticks between
the phase stamps of sc_tiled.h, one frozen tick of the contract workload.   python scripts/stamp_phases.py [particles]"""
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
buf = np.zeros((2, 1 << 16, 16), dtype=np.int64)
lib.sc_debug_stamps.restype = C.c_int
lib.sc_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sc_debug_stamps(s.engine._ctx, buf.ctypes.data_as(C.c_void_p)) == 0
waves = (n + 63) // 64
names = {0: ["start", "cell+buckets loaded", "tile staged", "scan same-row right", "scan next row", "scan same-row left",
             "scan previous row", "pair math", "lists out"],
         1: ["start", "bounds, lane, table loaded", "tile staged", "pair loop", "velocities staged", "viscosity+finish",
             "next tick's wall pass", "stores"]}
for k, label in ((0, "pass A"), (1, "pass B")):
    st = buf[k, :waves, :len(names[k])]
    ok = (st > 0).all(axis=1)
    st = st[ok]
    dt = np.diff(st, axis=1)
    life = st[:, -1] - st[:, 0]
    print(f"{label}: {ok.sum()} waves, median wave life {np.median(life):.0f} ticks; kernel span {(st[:, -1].max() - st[:, 0].min()):.0f} ticks")
    for j, nm in enumerate(names[k][1:]):
        print(f"    {nm:32s} median {np.median(dt[:, j]):8.0f}   mean {dt[:, j].mean():8.0f}   p95 {np.percentile(dt[:, j], 95):8.0f}")

# ---- timeline over the whole chip on the 100 MHz clock (slots 14 / 15): how many workgroups run at a time?
for k, label, wpb in ((0, "pass A", 4), (1, "pass B", 4)):
    st = buf[k, :waves, :].astype(np.float64)
    blocks = waves // wpb
    start = st[:blocks * wpb, 14].reshape(blocks, wpb).min(axis=1) * 0.01  # us
    end = st[:blocks * wpb, 15].reshape(blocks, wpb).max(axis=1) * 0.01
    ok = (start > 0) & (end > 0)
    start, end = start[ok], end[ok]
    t0 = start.min()
    start, end = start - t0, end - t0
    dur = end - start
    span = end.max()
    prof = [int(((start <= f * span) & (end > f * span)).sum()) for f in np.linspace(0.05, 0.95, 10)]
    print(f"{label}: {len(dur)} workgroups over {span:.1f} us; workgroup duration median {np.median(dur):.1f} p95 {np.percentile(dur, 95):.1f} "
          f"max {dur.max():.1f} us; first starts spread over {np.percentile(start, 25):.1f} us (25 % started); last start at {start.max():.1f} us")
    print(f"    workgroups running at 5 %, 15 %, ... 95 % of the span: {prof}  (slots: {256 * (6 if k == 0 else 4)})")
    late = np.argsort(end)[-5:]
    print(f"    the five last to finish: started at {np.round(start[late], 1)} ran {np.round(dur[late], 1)} us")
    ss = np.sort(start)
    print("    start time of the k-th workgroup (us):", {k: round(float(ss[k - 1]), 2) for k in (64, 256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096) if k <= len(ss)})
