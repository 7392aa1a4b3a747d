"""The sort kernels (scan, scatter, k_sort_big, reorder) of one tick deep in the pile-up regime, wave by wave, from a
-DSC_TIMELINE build: span, wave lives, and how long the kernel runs on its last few waves.
   python scripts/pile_timeline.py [particles] [ticks before the stamped one]"""
import copy, ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import bench, sand_crate_amd as sc
from sand_crate_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 470
wc, d = bench.world_for(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
p, v = bench.synthetic_state(n)
s.particles = p; s.particle_velocities = v
s.run(ticks); s.synchronize()
lib = N.load()
buf = np.zeros((8, 1 << 16, 4), dtype=np.int64)
lib.sc_debug_timeline.restype = C.c_int
lib.sc_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sc_debug_timeline(s.engine._ctx, buf.ctypes.data_as(C.c_void_p)) == 0
def span(k):
    st = buf[k][buf[k][:, 0] > 0]
    return (st[:, 0].min(), st[:, 1].max()) if len(st) else (0, 0)
scan = 2 if span(2)[0] > span(7)[0] else 7
passb = 1 if span(1)[0] > span(6)[0] else 6
order = ((scan, "scan"), (3, "scatter"), (5, "sort_big"), (4, "reorder"), (0, "pass A"), (passb, "pass B"))
tick0 = span(scan)[0]
prev_end = None
for k, label in order:
    st = buf[k][buf[k][:, 0] > 0]
    if not len(st):
        continue
    a, b = (st[:, 0].min() - tick0) * 0.01, (st[:, 1].max() - tick0) * 0.01
    start, end = (st[:, 0] - st[:, 0].min()) * 0.01, (st[:, 1] - st[:, 0].min()) * 0.01
    life = end - start
    es = np.sort(end)
    gap = "" if prev_end is None else f" gap {a - prev_end:5.2f}"
    print(f"{label:9s} {len(st):6d} waves, starts {a:7.2f} ends {b:7.2f} (span {b - a:6.2f} us){gap}; wave life median {np.median(life):6.2f} "
          f"p99 {np.percentile(life, 99):6.2f} max {life.max():6.2f}; last start {start.max():6.2f}; "
          f"ends: 50% {es[len(es) // 2]:6.2f} 90% {es[int(len(es) * .9)]:6.2f} 99% {es[int(len(es) * .99)]:6.2f} us")
    prev_end = b
if len(sys.argv) > 3:  # keep the raw stamps of the two passes (scripts/tile_order_sim.py)
    np.savez_compressed(sys.argv[3], a=buf[0][: (n + 63) // 64], b=buf[passb][: (n + 63) // 64])
