"""Long run of the contract workload (well into its pile-up regime) with a sanity check every block of ticks: no error
flag, no NaN, ids unique, and the last tick of the block repeated on a fresh context gives the same state bit for bit
(the sort of big buckets, the grouped cell counts and the dense-tile paths are all choices that must not show).
   python scripts/soak.py [particles] [blocks of 100 ticks]"""
import copy, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import bench, sand_crate_amd as sc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
wc, d = bench.world_for(n)
p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
s.particles = p; s.particle_velocities = v
for block in range(blocks):
    t0 = time.perf_counter()
    s.run(99); s.synchronize()
    dt = (time.perf_counter() - t0) / 99
    s.save_checkpoint("/tmp/soak_ckpt.npz")
    s.run(1); s.synchronize()
    after = s.engine.download()
    # the same tick on a fresh context restored from the checkpoint: the upload order differs from the storage order the
    # running context had, so every order-dependent choice inside the tick (arrival order in the buckets, atomics,
    # runs of equal cells, block placement) differs -- the state after the tick must not
    chk = sc.Crate.from_checkpoint("/tmp/soak_ckpt.npz", capacity=n + 1024)
    chk.run(1); chk.synchronize()
    again = chk.engine.download()
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(after, again))
    chk.engine.close()
    pos = after[0]
    cell = np.floor(pos[:, 1] / d).astype(np.int64) * 100000 + np.floor(pos[:, 0] / d).astype(np.int64)
    _, c = np.unique(cell, return_counts=True)
    ok = np.isfinite(pos).all() and len(np.unique(after[3])) == len(after[3])
    print(f"ticks {100 * (block + 1):5d}  {dt * 1e6:8.1f} us/tick  particles {len(pos)}  max/cell {c.max()}  cells>96 {(c > 96).sum()}  "
          f"|v|max {np.abs(after[1]).max():.1f}  finite+unique ids {ok}  restored context repeats the tick bit for bit: {same}", flush=True)
    assert ok and same
