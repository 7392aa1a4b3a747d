"""Occupancy of the chip over pass A / pass B, from a -DSC_TIMELINE build (sc_device.h): every wave's start and end on the
100 MHz clock and its hardware slot.   python scripts/timeline.py [particles] [waves per workgroup]

Prints, per kernel: span, workgroup life, how many workgroups are resident per CU over time, and -- the question this was
written for -- how long a CU's slot stays empty between one workgroup's end and the next one's start."""
import copy, ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import bench, sand_crate_amd as sc
from sand_crate_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
wpb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wc, d = bench.world_for(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
p, v = bench.synthetic_state(n)
s.particles = p; s.particle_velocities = v
import time
s.run(20); s.synchronize()
t_run = time.perf_counter(); s.run(20); s.synchronize(); period = (time.perf_counter() - t_run) / 20 * 1e6
print(f"tick period over 20 ticks: {period:.1f} us (the gap between pass B's last wave and the next tick's first wave is this minus the tick's span below)")
lib = N.load()
buf = np.zeros((8, 1 << 16, 4), dtype=np.int64)
lib.sc_debug_timeline.restype = C.c_int
lib.sc_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sc_debug_timeline(s.engine._ctx, buf.ctypes.data_as(C.c_void_p)) == 0
# the last two ticks: pass B and the scan alternate between two slots by tick parity
def first_last(k):
    st = buf[k][buf[k][:, 0] > 0]
    return st[:, 0].min(), st[:, 1].max()
(b1s, b1e), (b6s, b6e), (s2s, s2e), (s7s, s7e) = first_last(1), first_last(6), first_last(2), first_last(7)
older_b, newer_scan = ((b1s, b1e), (s7s, s7e)) if b1s < b6s else ((b6s, b6e), (s2s, s2e))
print(f"pass B spans of the last two ticks: {(b1e - b1s) * 0.01:.2f} / {(b6e - b6s) * 0.01:.2f} us; scan spans {(s2e - s2s) * 0.01:.2f} / {(s7e - s7s) * 0.01:.2f} us; "
      f"scan start -> pass B end within the older tick: {(older_b[1] - min(s2s, s7s)) * 0.01:.2f} us")
print(f"pass B of the tick before: last wave ends -> {(newer_scan[0] - older_b[1]) * 0.01:.2f} us -> first wave of the last tick's scan; "
      f"tick period by the two scans' first waves {abs(s2s - s7s) * 0.01:.2f} us")
older_B = buf[1].copy() if b1s < b6s else buf[6].copy()
if b1s < b6s:  # keep the newer tick in the slots the table below reads
    buf[1], buf[2] = buf[6], buf[2] if s2s > s7s else buf[7]
else:
    buf[2] = buf[2] if s2s > s7s else buf[7]
buf[6] = older_B
# the tick as a whole: when every kernel's first wave started and its last wave ended (us from the scan's first wave)
order = ((2, "scan"), (3, "scatter"), (5, "sort_big"), (4, "reorder"), (0, "pass A"), (1, "pass B"))
tick0 = min(buf[k][buf[k][:, 0] > 0][:, 0].min() for k, _ in order if (buf[k][:, 0] > 0).any())
prev_end = None
for k, label in order:
    st = buf[k][buf[k][:, 0] > 0]
    if not len(st):
        continue
    a, b = (st[:, 0].min() - tick0) * 0.01, (st[:, 1].max() - tick0) * 0.01
    gap = "" if prev_end is None else f"   gap after the previous kernel's last wave {a - prev_end:6.2f} us"
    print(f"{label:9s} first wave starts {a:8.2f} us, last wave ends {b:8.2f} us  (span {b - a:6.2f}){gap}")
    prev_end = b
for k, label in ((0, "pass A"), (1, "pass B (the run's last tick: no look-ahead)"), (6, "pass B (the tick before: with the next tick's wall pass)")):
    st = buf[k]
    ok = st[:, 0] > 0
    st = st[ok]
    nw = len(st)
    t0 = st[:, 0].min()
    start = (st[:, 0] - t0) * 0.01  # us
    end = (st[:, 1] - t0) * 0.01
    hw, xcc = st[:, 2], st[:, 3] & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ncu = len(np.unique(cuid))
    span = end.max()
    life = end - start
    print(f"{label}: {nw} waves on {ncu} CUs, span {span:.1f} us, wave life median {np.median(life):.2f} p95 {np.percentile(life, 95):.2f} us; "
          f"sum of wave lives / (span x SIMDs) = {life.sum() / (span * ncu * 4):.2f} waves per SIMD")
    ss = np.sort(start)
    print("    start of the k-th wave (us):", {q: round(float(ss[q - 1]), 1) for q in (256, 1024, 4096, 6144, 8192, 12288, 16384) if q <= nw})
    # resident waves per CU over time
    grid = np.linspace(0, span, 41)[1:-1]
    res = [(((start <= g) & (end > g)).sum() / ncu) for g in grid]
    print("    resident waves per CU at 2.5 % steps of the span:", " ".join(f"{r:.1f}" for r in res))
    # per CU: gap between a wave's start and the latest earlier end of a wave on that CU (slot refill time)
    gaps = []
    for c in np.unique(cuid):
        m = cuid == c
        s_c, e_c = np.sort(start[m]), np.sort(end[m])
        for sv in s_c[s_c > e_c[0]]:
            prev = e_c[np.searchsorted(e_c, sv, side="right") - 1]
            gaps.append(sv - prev)
    gaps = np.array(gaps)
    if len(gaps):
        print(f"    start of a wave minus the last end of a wave on its CU before it: median {np.median(gaps):.2f} mean {gaps.mean():.2f} "
              f"p90 {np.percentile(gaps, 90):.2f} us ({len(gaps)} starts)")
    # per CU busy integral
    per_cu = np.array([life[cuid == c].sum() for c in np.unique(cuid)])
    print(f"    per CU: sum of wave lives min {per_cu.min():.0f} median {np.median(per_cu):.0f} max {per_cu.max():.0f} us; waves per CU min "
          f"{np.bincount(cuid)[np.unique(cuid)].min()} max {np.bincount(cuid)[np.unique(cuid)].max()}")
    print(f"    waves by SIMD: {np.bincount(simd)}")
