"""Worker launched by torch.distributed.run: runs a SlabCrate for a few ticks and has rank 0 save the
gathered state.  --backend oracle needs no GPU (gloo); --backend hip puts every rank on cuda:0."""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def synthetic_world(n, noise_level, vel, margin=0.0, seed=5, skew=1.0):
    import sand_crate_amd as sc
    rs = np.random.RandomState(seed)
    d = float(np.sqrt(12 / (np.pi * n)))
    p = rs.rand(n, 2) * (1 - 2 * margin) + margin
    if skew != 1.0:
        p[:, 0] = p[:, 0] ** skew
    v = (rs.rand(n, 2) - 0.5) * vel
    cfg = sc.load_config(ROOT / "config" / "wave_machine.yaml")
    co = cfg.world_config.coefficients
    co.update(particle_radius=d / 2, dt=0.002 * d / 0.01, collider_noise_level=noise_level, max_particles=n)
    cfg.world_config.particle_sources = []
    return cfg.world_config, p, v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="oracle")
    ap.add_argument("--particles", type=int, default=3000)
    ap.add_argument("--ticks", type=int, default=4)
    ap.add_argument("--vel", type=float, default=30.0)
    ap.add_argument("--noise", default="counter")
    ap.add_argument("--margin", type=float, default=0.0)
    ap.add_argument("--mixed", action="store_true", help="run(2), physics_tick(), run(rest): look-ahead on and off")
    ap.add_argument("--halo-capacity", type=int, default=0, help="records per halo message (0 = SlabCrate's default)")
    ap.add_argument("--rebalance-every", type=int, default=0)
    ap.add_argument("--skew", type=float, default=1.0, help="x -> x ** skew: more particles on the left")
    ap.add_argument("--axis", default="x", choices=["x", "y"], help="slabs of columns (x) or of rows (y)")
    ap.add_argument("--scene", default="", help="a YAML scene as shipped (its particle sources included), started empty")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch.distributed as dist

    from sand_crate_amd.slab import SlabCrate
    dist.init_process_group("gloo")
    wc, p, v = synthetic_world(a.particles, 0.1 if a.noise == "counter" else 0.0, a.vel, margin=a.margin, skew=a.skew)
    if a.scene:
        import sand_crate_amd as sc
        wc = sc.load_config(ROOT / "config" / a.scene).world_config
        p, v = np.zeros((0, 2)), np.zeros((0, 2))
    backend = None
    if a.backend == "oracle":
        from slab_oracle_backend import OracleSlabBackend
        backend = OracleSlabBackend(halo_capacity=a.particles, noise=a.noise, noise_seed=9)
    sim = SlabCrate(wc, p, v, device=0, noise=a.noise, noise_seed=9, backend=backend,
                    halo_capacity=a.halo_capacity or None, rebalance_every=a.rebalance_every, axis=a.axis)
    first_cuts = list(sim.slabs)
    if a.mixed:
        sim.run(2)
        sim.physics_tick()
        sim.run(a.ticks - 3)
    else:
        sim.run(a.ticks)
    sim.synchronize()
    count = sim.global_particle_count()
    gp, gv, gpr, gids = sim.gather_state()
    if dist.get_rank() == 0:
        np.savez(a.out, particles=gp, velocities=gv, pressure=gpr, ids=gids, count=count,
                 slabs=np.array(sim.slabs, dtype=np.float64), first_slabs=np.array(first_cuts, dtype=np.float64),
                 rebalances=sim.rebalances)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
