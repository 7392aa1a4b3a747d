"""per-tick cost of one slab rank (in-process chain of 2 slabs on one GPU) against the single-domain tick of the same
per-GPU size"""
import copy, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import bench, sand_crate_amd as sc
from sand_crate_amd.slab import SlabChain
per = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
nslab = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = per * nslab
wc, d = bench.world_for(n)
p, v = bench.synthetic_state(n)
import itertools
for overlap, cap, flag in ((False, "x", False), (False, "y", False), (True, "x", False), (True, "y", False), (True, "y", True)):
    chain = SlabChain(copy.deepcopy(wc), p, v, nslab, noise="counter", noise_seed=1, overlap=overlap, axis=cap, band_flag=flag)
    chain.run(5); chain.synchronize()
    eng = chain.members[0].engine
    eng.reset_timing(); eng.enable_timing(True)
    t0 = time.perf_counter(); chain.run(20); chain.synchronize(); dt = (time.perf_counter() - t0) / 20
    eng.enable_timing(False)
    tm = {k: round(1000 * ms / 20, 1) for k, (ms, c) in eng.timing().items() if c}
    print(f"chain of {nslab} x {per}, axis {cap}, overlap {overlap}, one launch + flag {flag}: {dt*1e3:.4f} ms per tick for all slabs = {dt*1e3/nslab:.4f} per slab; member 0 kernels per tick: {tm}  sum {sum(tm.values()):.1f}", flush=True)
    del chain
wc1, d1 = bench.world_for(per)
p1, v1 = bench.synthetic_state(per)
s = sc.Crate(copy.deepcopy(wc1), noise="counter", noise_seed=1, capacity=per + 1024)
s.particles = p1; s.particle_velocities = v1
s.run(5); s.synchronize()
e = s.engine; e.reset_timing(); e.enable_timing(True)
t0 = time.perf_counter(); s.run(20); s.synchronize(); dt = (time.perf_counter() - t0) / 20
e.enable_timing(False)
tm = {k: round(1000 * ms / 20, 1) for k, (ms, c) in e.timing().items() if c}
print(f"single domain {per}: {dt*1e3:.4f} ms per tick; kernels {tm} sum {sum(tm.values()):.1f}")
