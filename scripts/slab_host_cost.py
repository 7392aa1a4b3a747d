"""What a tick costs the HOST in `SlabCrate.run` against the single-domain `Crate.run`: a slab of one
rank (no neighbors, no exchange: the Python loop, its library calls and the rigid-body update are what differs), at
2,097,152 particles -- BASELINE.json configs[4]'s share of one GPU -- and at 4,096, where the GPU's tick is short and the
wall time per tick IS the host's.   python scripts/slab_host_cost.py"""
import copy, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
torch.cuda.init()  # (before the library's own first HIP call)
import bench, sand_crate_amd as sc
from sand_crate_amd.slab import SlabCrate


def single(n):
    wc, d = bench.world_for(n)
    p, v = bench.synthetic_state(n)
    s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
    s.particles, s.particle_velocities = p, v
    return s


def slab(n):
    wc, d = bench.world_for(n)
    p, v = bench.synthetic_state(n)
    return SlabCrate(copy.deepcopy(wc), p, v, device=0, noise="counter", noise_seed=1, axis="y")


for n, ticks in ((4096, 400), (2097152, 100)):
    for name, make in (("Crate.run", single), ("SlabCrate.run (one rank)", slab)):
        s = make(n)
        s.run(10); s.synchronize()
        t0 = time.perf_counter(); s.run(ticks); t1 = time.perf_counter(); s.synchronize(); t2 = time.perf_counter()
        # the host alone: four ticks enqueued on an idle queue (the library lets the host run four ticks ahead)
        t3 = time.perf_counter(); s.run(4); t4 = time.perf_counter(); s.synchronize()
        print(f"{n:8d} particles  {name:26s} {1e6 * (t2 - t0) / ticks:8.1f} us per tick ({1e6 * (t1 - t0) / ticks:7.1f} until run() returned); "
              f"host alone, 4 ticks on an idle queue: {1e6 * (t4 - t3) / 4:6.1f} us per tick", flush=True)
        del s
