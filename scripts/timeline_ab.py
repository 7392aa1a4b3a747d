"""Pass A's and pass B's waves inside the ONE launch of k_pass_ab, from a -DSC_TIMELINE build: when each role's waves start
and end, and how many waves of each role a CU holds over the launch.   python scripts/timeline_ab.py [particles]"""
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
buf = np.zeros((8, 1 << 16, 4), dtype=np.int64)
lib.sc_debug_timeline.restype = C.c_int
lib.sc_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sc_debug_timeline(s.engine._ctx, buf.ctypes.data_as(C.c_void_p)) == 0
tick = s.tick  # ticks done; the last one (tick - 1) ran as two launches, the one before (tick - 2) as one
par = (tick - 2) & 1
A = buf[5 if par else 0]; B = buf[6 if par else 1]  # (both kernels: A by parity in slots 0 / 5, B in 1 / 6)
A = A[A[:, 0] > 0]; B = B[B[:, 0] > 0]
t0 = min(A[:, 0].min(), B[:, 0].min())
f = lambda x: (x - t0) * 0.01
print(f"{n} particles, tick {tick - 2}: pass A {len(A)} waves, first start {f(A[:,0].min()):.2f} last start {f(A[:,0].max()):.2f} last end {f(A[:,1].max()):.2f} us, "
      f"wave life median {np.median(A[:,1]-A[:,0])*0.01:.2f} p95 {np.percentile(A[:,1]-A[:,0],95)*0.01:.2f}")
print(f"                 pass B {len(B)} waves, first start {f(B[:,0].min()):.2f} last start {f(B[:,0].max()):.2f} last end {f(B[:,1].max()):.2f} us, "
      f"wave life median {np.median(B[:,1]-B[:,0])*0.01:.2f} p95 {np.percentile(B[:,1]-B[:,0],95)*0.01:.2f}")
end = max(A[:, 1].max(), B[:, 1].max())
steps = np.linspace(t0, end, 41)[:-1]
ra = [((A[:, 0] <= x) & (A[:, 1] > x)).sum() / 256 for x in steps]
rb = [((B[:, 0] <= x) & (B[:, 1] > x)).sum() / 256 for x in steps]
print("resident waves per CU at 2.5 % steps of the launch (pass A / pass B):")
print("  A: " + " ".join(f"{x:4.1f}" for x in ra))
print("  B: " + " ".join(f"{x:4.1f}" for x in rb))
print(f"launch span {(end - t0) * 0.01:.2f} us")
a_end = A[:, 1].max()
for name, W in (("A", A), ("B", B)):
    life = (W[:, 1] - W[:, 0]) * 0.01
    st = f(W[:, 0])
    qs = np.quantile(st, [0, .25, .5, .75, 1.0])
    print(f"pass {name} wave life by start-time quartile: " + "  ".join(
        f"[{qs[k]:.0f}-{qs[k+1]:.0f} us] {np.median(life[(st >= qs[k]) & (st <= qs[k+1])]):.2f}" for k in range(4)))
lateB = B[B[:, 0] > a_end]
if len(lateB):
    print(f"pass B waves that start after pass A's last wave has ended ({f(a_end):.1f} us): {len(lateB)}, life median {np.median(lateB[:,1]-lateB[:,0])*0.01:.2f} us")
