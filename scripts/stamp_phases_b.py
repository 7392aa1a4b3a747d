"""Where a wave of pass B spends its life, with and without the look-ahead epilogue (diagnostic build with -DSC_STAMPS):
the last tick of a run has no look-ahead, the tick before it has; their stamps go to different buffers (sc_tiled.h:
SC_STAMP_B).   python scripts/stamp_phases_b.py [particles]"""
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
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s.run(ticks); s.synchronize()
lib = N.load()
buf = np.zeros((6, 1 << 16, 24), dtype=np.int64)
lib.sc_debug_stamps.restype = C.c_int
lib.sc_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sc_debug_stamps(s.engine._ctx, buf.ctypes.data_as(C.c_void_p)) == 0
waves = (n + 63) // 64
order = [0, 1, 2, 3, 4, 5, 16, 17, 18, 19, 6, 7]
names = ["bounds, lane, table loaded", "tile staged", "pair loop", "velocities staged", "viscosity+finish",
         "look-ahead: near segments", "look-ahead: wall_and_cell", "look-ahead: cell stores", "look-ahead: count_cells", "halo pack", "stores"]
for k in (1, 3):
    st = buf[k, :waves]
    fused = (st[:, 16] > 0).mean() > 0.5
    cols = order if fused else [0, 1, 2, 3, 4, 5, 6, 7]
    nm = names if fused else names[:5] + ["look-ahead (none)", "stores"]
    ok = (st[:, cols] > 0).all(axis=1)
    # a slot the launch never wrote still holds an earlier launch's stamp: a wave counts only when its stamps ascend (and
    # a buffer in which most waves do not is not a launch of this form at all -- r03's third block of negative medians)
    ok &= (np.diff(st[:, cols], axis=1) >= 0).all(axis=1)
    if ok.sum() < 0.5 * len(st):
        print(f"pass B buffer {k}: {ok.sum()} of {len(st)} waves have a complete ascending set of stamps -- not a launch of this form, skipped")
        continue
    sel = st[ok][:, cols]
    dt = np.diff(sel, axis=1)
    life = sel[:, -1] - sel[:, 0]
    print(f"pass B, {'with' if fused else 'without'} the look-ahead epilogue: {ok.sum()} waves, median wave life {np.median(life):.0f} cycles")
    for j, label in enumerate(nm):
        print(f"    {label:32s} median {np.median(dt[:, j]):8.0f}   mean {dt[:, j].mean():8.0f}   p95 {np.percentile(dt[:, j], 95):8.0f}")
