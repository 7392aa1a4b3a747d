"""Where a wave of the scatter and of K4 (reorder) spends its life (diagnostic build with -DSC_STAMPS), at a tick of the
uniform regime or deep in the pile-up.   python scripts/sort_stamps.py [particles] [ticks before the stamped one]"""
import copy, ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import bench, sand_crate_amd as sc
from sand_crate_amd import _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wc, d = bench.world_for(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
p, v = bench.synthetic_state(n)
s.particles = p; s.particle_velocities = v
s.run(ticks); s.synchronize()
lib = N.load()
buf = np.zeros((6, 1 << 16, 24), dtype=np.int64)
lib.sc_debug_stamps.restype = C.c_int
lib.sc_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sc_debug_stamps(s.engine._ctx, buf.ctypes.data_as(C.c_void_p)) == 0
waves = (n + 63) // 64
def show(k, title, cols, names):
    st = buf[k, :waves]
    ok = (st[:, cols] > 0).all(axis=1)
    if not ok.any():
        return print(f"{title}: no such waves")
    sel = st[ok][:, cols]
    dt = np.diff(sel, axis=1)
    life = sel[:, -1] - sel[:, 0]
    print(f"{title}: {ok.sum()} waves with every stamp, wave life median {np.median(life):.0f} mean {life.mean():.0f} p95 {np.percentile(life, 95):.0f} cycles")
    slow = life >= np.percentile(life, 99)  # the waves the kernel ends with
    for j, label in enumerate(names):
        print(f"    {label:44s} median {np.median(dt[:, j]):8.0f}   mean {dt[:, j].mean():8.0f}   p95 {np.percentile(dt[:, j], 95):8.0f}   p99 {np.percentile(dt[:, j], 99):8.0f}   slowest 1 % of waves: mean {dt[slow, j].mean():8.0f}")
    idx = np.flatnonzero(ok)[slow]
    print(f"    the slowest 1 % of waves: mean life {life[slow].mean():.0f} cycles; their wave numbers (of {waves}): " + " ".join(str(i) for i in idx[:: max(1, len(idx) // 24)]))
# only the grouping variant of the scatter (pile-up regime: the workgroup's cell table) takes stamp 2: report the two kinds apart
st = buf[4, :waves]
scr = st[:, 2] > st[:, 1]
print(f"scatter: {scr.sum()} of {waves} waves went through the workgroup's cell table")
keep = buf[4, :waves].copy()
buf[4, :waves][~scr] = 0
show(4, "scatter, cell table", [0, 1, 2, 3, 4], ["cell, x, id, live count loaded", "grouped by cell in the workgroup's LDS table", "bucket starts + returning atomics (one per cell and workgroup)", "key and cell stored"])
buf[4, :waves] = keep
buf[4, :waves][scr] = 0
show(4, "scatter, waves in runs", [0, 1, 3, 4], ["cell, x, id, live count loaded", "runs, bucket starts + returning atomics", "key and cell stored"])
show(5, "reorder", [0, 1, 2, 3, 4, 5, 6], ["own key, window swept, gathers back", "ranked inside the window", "searches in a sorted bucket's other chunks", "small bucket outside the window", "big unsorted buckets, workgroup-wide", "stores"])
