"""Kernel times of ONE tick on a frozen state: the state after 20 ticks of the contract workload (made once, with the
library that is in place then) is uploaded again before every measured tick, so variants that change the physics
(ablations) are still timed on identical inputs.   python scripts/frozen_time.py <tag> [particles] [reps]"""
import copy, os, sys
sys.path.insert(0, ".")
import numpy as np
import bench, sand_crate_amd as sc
tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
wc, d = bench.world_for(n)
settle = int(os.environ.get("FROZEN_TICKS", "20"))  # 450: the pile-up regime of the contract workload at 1,048,576
path = f"/tmp/frozen_{n}_{settle}.npz"
s = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
if not os.path.exists(path):
    p, v = bench.synthetic_state(n)
    s.particles = p; s.particle_velocities = v
    s.run(settle); s.synchronize()
    p, v, _, _ = s.engine.download()
    np.savez(path, p=p, v=v)
z = np.load(path); p, v = z["p"], z["v"]
e = s.engine
acc = {}
for r in range(reps + 2):
    s.particles = p; s.particle_velocities = v
    s.synchronize(); e.reset_timing(); e.enable_timing(True)
    s.run(2); s.synchronize(); e.enable_timing(False)
    if r >= 2:
        for k, (ms, c) in e.timing().items():
            if c: acc.setdefault(k, []).append(1000 * ms / c)
med = {k: float(np.median(x)) for k, x in acc.items()}
print(f"{tag:32s} A {med.get('neighbors_density', 0):6.1f}  B {med.get('force_integrate', 0):6.1f}  reorder {med.get('reorder', 0):5.1f} scatter {med.get('scatter', 0):5.1f} scan {med.get('cell_scan', 0):5.1f} wall_bin {med.get('wall_bin', 0):5.1f}", flush=True)
