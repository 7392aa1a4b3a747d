"""pile-up regime: per-workgroup durations of pass A / pass B against the tile's cell occupancy (-DSC_STAMPS build)"""
import copy, ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import bench, sand_crate_amd as sc
from sand_crate_amd import _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
wc, d = bench.world_for(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
p, v = bench.synthetic_state(n)
s.particles = p; s.particle_velocities = v
s.run(T); s.synchronize()
lib = N.load()
buf = np.zeros((6, 1 << 16, 24), dtype=np.int64)
lib.sc_debug_stamps.restype = C.c_int
lib.sc_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sc_debug_stamps(s.engine._ctx, buf.ctypes.data_as(C.c_void_p)) == 0
waves = (n + 63) // 64
names = {0: ["start", "loaded", "staged", "scan right", "scan next row", "scan left", "scan prev row", "pair math", "lists out"],
         1: ["start", "loaded", "staged", "pair loop", "v staged", "finish", "K1", "stores"]}
for k, label in ((0, "pass A"), (1, "pass B")):
    st = buf[k, :waves, :].astype(np.float64)
    ok = st[:, len(names[k]) - 1] > 0
    st = st[ok]
    life = (st[:, len(names[k]) - 1] - st[:, 0]) / 2100.0   # us at ~2.1 GHz
    total = st[:, 10]
    print(f"{label}: {ok.sum()} waves, sum of wave lives {life.sum():.0f} us, median {np.median(life):.1f}, p90 {np.percentile(life, 90):.1f}, p99 {np.percentile(life, 99):.1f}, max {life.max():.1f}")
    for lo, hi in ((0, 961), (961, 1101), (1101, 4000), (4000, 20000), (20000, 65536), (65536, 10**9)):
        m = (total >= lo) & (total < hi)
        if m.any():
            ph = np.diff(st[m][:, :len(names[k])], axis=1).mean(axis=0)
            if k == 0:
                w = st[m][:, 14].astype(np.int64)
                print(f"      rounds/wave {st[m][:,12].mean():.1f}  stagings {(st[m][:,13].astype(np.int64) & 255).mean():.1f}  coop steps {st[m][:,15].mean():.1f}  lanes wanting scan1..4: {(w & 255).mean():.1f} {((w>>8)&255).mean():.1f} {((w>>16)&255).mean():.1f} {((w>>24)&255).mean():.1f}")
            if k == 0:
                print("      ticks in the windowed scans: " + " ".join(f"{nm}={st[m][:, 16 + q].mean():.0f}" for q, nm in enumerate(["rounds+barriers", "staging", "serial", "wave-wide", "between scans"])))
            print(f"   tiles with {lo} <= entries < {hi}: {m.sum():6d} waves  total {life[m].sum():9.0f} us  mean {life[m].mean():7.1f} us  max {life[m].max():7.1f}   phases(ticks): " + " ".join(f"{nm}={x:.0f}" for nm, x in zip(names[k][1:], ph)))
    if k == 0:
        order = np.argsort(life)[-12:]
        for o in order:
            w = int(st[o, 14])
            print(f"      slow wave: life {life[o]:7.1f} us  entries {int(st[o,10])}  tile {int(st[o,11])}  rounds {int(st[o,12])} stagings {int(st[o,13]) & 255} inner steps {(int(st[o,13]) >> 8) & 0xffffff} (scan2: {int(st[o,13]) >> 40}) coop {int(st[o,15])} sumC before/after scan2 {int(st[o,9]) & 0xffffffff}/{int(st[o,9]) >> 32} wants {w&255} {(w>>8)&255} {(w>>16)&255} {(w>>24)&255}  phases " + " ".join(f"{x:.0f}" for x in np.diff(st[o, :9])[:4]) + "  windowed: " + " ".join(f"{nm}={st[o, 16 + q]:.0f}" for q, nm in enumerate(["rounds+barriers", "staging", "serial", "wave-wide", "between"])))
        # how the kernel's critical path looks: start/end of waves relative to the first start (ticks -> us)
        t0 = st[:, 0].min()
        print(f"      kernel span by stamps {(st[:, 8].max() - t0) / 2100:.1f} us; waves starting after 100 us: {(st[:,0] - t0 > 100 * 2100).sum()}; last start {(st[:,0].max() - t0) / 2100:.1f} us")

# k_sort_big: one task per workgroup (the first pass of its task loop), slots 0-6 = phase boundaries, slot 8 = keys in the task
st = buf[2, :4 * 4096, :].astype(np.float64)
st = st[st[:, 6] > 0]
if len(st):
    names2 = ["loaded", "samples sorted", "bins found", "bin starts", "keys binned", "ranked + stored"]
    life = (st[:, 6] - st[:, 0]) / 2100.0
    print(f"k_sort_big: {len(st)} waves, life median {np.median(life):.1f} us, max {life.max():.1f}; kernel span {(st[:, 6].max() - st[:, 0].min()) / 2100:.1f} us")
    for lo, hi in ((0, 513), (513, 1025), (1025, 1537), (1537, 2049)):
        m = (st[:, 8] >= lo) & (st[:, 8] < hi)
        if m.any():
            ph = np.diff(st[m][:, :7], axis=1).mean(axis=0)
            print(f"   tasks of {lo}..{hi - 1} keys: {m.sum():5d} waves, life {life[m].mean():6.1f} us, phases (us): " + " ".join(f"{nm}={x / 2100:.1f}" for nm, x in zip(names2[0:], ph)) + f"; largest bin seen by a wave: mean {st[m][:, 9].mean():.0f}, max {st[m][:, 9].max():.0f}")
