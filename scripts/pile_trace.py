"""The contract workload run into its pile-up regime, for a kernel trace: nothing is printed but the wall time per
window.   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python scripts/pile_trace.py [particles] [ticks]
scripts/pile_trace_summary.py <dir> then averages every kernel over the LAST launches of the run."""
import copy, sys, time
sys.path.insert(0, ".")
import torch
torch.cuda.init()
import bench, sand_crate_amd as sc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
total = int(sys.argv[2]) if len(sys.argv) > 2 else 500
wc, d = bench.world_for(n)
p, v = bench.synthetic_state(n)
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
s.particles = p; s.particle_velocities = v
for w0 in range(0, total, 50):
    t0 = time.perf_counter()
    s.run(50); s.synchronize()
    print(f"ticks {w0:4d}-{w0 + 50:4d}  {(time.perf_counter() - t0) / 50 * 1e3:7.4f} ms/tick", flush=True)
