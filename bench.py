#!/usr/bin/env python3
"""Throughput of the SandCrate particle update on MI355X: particle-steps per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P_PER_GPU]

Workload (BASELINE.json configs[1] at N=1): P = 262,144 synthetic
uniformly seeded particles per GPU in the wave_machine.yaml world (its coefficients, both rigid
bodies incl. the motored wall, no particle source), particle diameter d = sqrt(12 / (pi P_total))
so that a particle has ~12 neighbors, dt scaled with d, collider_noise_level 0.1 from a
counter-based device RNG.  A step is one `physics_tick` of all particles; state is resident in HBM
before the timed region and nothing is read back inside it.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline      the dominant kernel's algorithmic bytes / measured HIP-event time vs the 8 TB/s HBM peak
  kernels       the same for every kernel of the tick
  cpu_baseline  the oracle's loop-structured tick (stands for the reference's NumPy path, which cannot
                travel to the GPU box) timed on this host, one core, on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s is the measured copy rate)
# Algorithmic bytes per particle-step of each kernel, float64 SoA:
#   wall_bin  read x,y 16 + write x,y 16 + cell id 4                      = 36
#   scatter   read cell id 4 + write slot 4                               =  8
#   reorder   read x,y,vx,vy 32 + perm 4, write 32 + id 4                 = 72
#   density   read x,y 16 -> write P, sx, sy 24                           = 40
#   force     read x,y,vx,vy,P,sx,sy 56 -> write x,y,vx,vy 32             = 88
# The neighbor-list kernel has no algorithmic bytes: a materialised list is an implementation
# choice the contract figure does not pay for.
ALGO_BYTES = {"wall_bin": 36, "scatter": 8, "reorder": 72, "density": 40, "neighbors_density": 40, "force_integrate": 88,
              "neighbors": 0, "cell_scan": 0, "noise_offsets": 0, "append": 0}
TICK_BYTES = 244
FORCE_BYTES = 128


def measured_traffic(particles_per_gpu: int, kernel: str):
    """HBM-side bytes per launch of `kernel` from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, calibrated on a float64 copy: scripts/collect_traffic.sh -> profiles/r01_traffic_<N>.json).
    bench.py cannot collect hardware counters itself; it reports the committed measurement of the same
    workload, or None when there is none for this size."""
    path = ROOT / "profiles" / f"r01_traffic_{particles_per_gpu}.json"
    if not path.exists():
        return None, None
    data = json.loads(path.read_text())
    k = data["kernels"].get(kernel)
    if not k:
        return None, None
    return k["traffic_bytes"], f"profiles/{path.name}: {data['source']}; {data['note']}"


def synthetic_state(n: int, seed: int = 1234):
    rs = np.random.RandomState(seed)
    p = rs.rand(n, 2) * 0.96 + 0.02
    v = (rs.rand(n, 2) - 0.5) * 0.1
    return p, v


def world_for(n_total: int):
    import sand_crate_amd as sc
    cfg = sc.load_config(ROOT / "config" / "wave_machine.yaml")
    d = float(np.sqrt(12.0 / (np.pi * n_total)))
    co = cfg.world_config.coefficients
    co["particle_radius"] = d / 2
    co["dt"] = 0.002 * (d / 0.01)
    co["max_particles"] = n_total
    cfg.world_config.particle_sources = []
    return cfg.world_config, d


def cpu_baseline(sample_n: int):
    """One tick of the oracle's loop-structured restatement on `sample_n` particles of the same
    synthetic generator (same neighbor density), single core."""
    from oracle.tick import BodyState, tick_core
    from oracle.tick_loops import tick_loops
    from oracle.world import build_bodies
    wc, d = world_for(sample_n)
    p, v = synthetic_state(sample_n)
    co = dict(wc.coefficients)
    co["gravity"] = np.array(co["gravity"], dtype=np.float64)
    bodies = build_bodies(wc.rigid_bodies)
    for b in bodies:
        b.advance(co["dt"])
    seg = np.vstack([b.segments for b in bodies])
    bs = [BodyState(np.asarray(b.position, float), np.asarray(b.center_velocity, float),
                    float(b.angular_clockwise_velocity), len(b)) for b in bodies]
    rs = np.random.RandomState(0)
    t0 = time.perf_counter()
    tick_loops(p, v, seg, bs, co, eta_source=lambda total: rs.rand(total, 2))
    t_loop = time.perf_counter() - t0
    t0 = time.perf_counter()
    tick_core(p, v, seg, bs, co, eta_u01=lambda total: rs.rand(total, 2))
    t_vec = time.perf_counter() - t0
    return {
        "value": sample_n / t_loop, "unit": "particle-steps/s", "cores": 1, "kind": "port",
        "sample": f"1 tick of oracle.tick_loops (per-particle Python loops like the reference's crate.py) on "
                  f"{sample_n} particles of the same synthetic generator, {t_loop:.1f} s; the path is "
                  f"single-threaded; host has {os.cpu_count()} logical cores",
        "vectorised_port_value": sample_n / t_vec,
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=262144, help="particles per GPU")
    ap.add_argument("--cpu-sample", type=int, default=262144,
                    help="particles in the CPU baseline tick (0 = skip); the default is one tick of the full workload, ~12 s")
    ap.add_argument("--noise", default="counter", choices=["counter", "none"])
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: every rank uses cuda:0 and the gloo backend (halo staged through the "
                         "host), to exercise the N > 1 code path on a one-GPU box; the numbers mean nothing")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region (no roofline in the output)")
    args = ap.parse_args()

    import torch

    import sand_crate_amd as sc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    n_total = args.particles * world

    wc, d = world_for(n_total)
    p, v = synthetic_state(n_total)
    if world > 1:
        import torch.distributed as dist
        from sand_crate_amd.slab import SlabCrate
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        barrier = dist.barrier
    else:
        def barrier():
            return None

    def make_sim():
        import copy
        w = copy.deepcopy(wc)
        if world > 1:
            return SlabCrate(w, p, v, device=local_rank, noise=args.noise, noise_seed=1)
        s = sc.Crate(w, device=local_rank, noise=args.noise, noise_seed=1, capacity=n_total + 1024)
        s.particles = p
        s.particle_velocities = v
        return s

    # first-use costs of the runtime (code object load, first launches) are paid on a throwaway simulation, so that
    # they fall neither into the W warm-up steps' state nor -- with --warmup 0 -- into the timed region
    primer_world, _ = world_for(4096)
    primer = sc.Crate(primer_world, device=local_rank, noise=args.noise, noise_seed=1, capacity=8192)
    primer.particles, primer.particle_velocities = synthetic_state(4096)
    primer.run(3)
    primer.synchronize()
    del primer

    sim = make_sim()

    def run(k):
        sim.run(k)

    device_flags = []

    def settle(s):
        """Synchronise; a condition the device flagged (a particle lost to NaN, a halo buffer that
        overflowed) is reported in the JSON line instead of aborting the measurement."""
        from sand_crate_amd._native import NativeError
        try:
            s.synchronize()
        except NativeError as err:
            device_flags.append(str(err))

    run(args.warmup)
    settle(sim)
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides, no per-kernel events
    eng = sim.engine
    barrier()
    torch.cuda.synchronize()
    settle(sim)
    t0 = time.perf_counter()
    run(args.steps)
    settle(sim)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- kernel durations: the same ticks replayed from the same initial state, every launch
    # bracketed by two HIP events on the stream the kernels run on.  Kept out of the timed region because the 14 event records per tick
    # cost ~15 % wall time at this size (measured); kernel durations themselves are unaffected.
    timing = {}
    n_live = sim.particle_count if world == 1 else sim.global_particle_count()
    if not args.no_kernel_events:
        sim = make_sim()  # same initial state, same ticks as the timed region
        eng = sim.engine
        run(args.warmup)
        settle(sim)
        eng.reset_timing()
        eng.enable_timing(True)
        run(args.steps)
        settle(sim)
        eng.enable_timing(False)
        timing = eng.timing()

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0 and args.no_kernel_events:
        print(json.dumps({"metric": "particle-steps/sec", "value": n_total * args.steps / elapsed,
                          "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1000.0 * elapsed / args.steps, "note": "no per-kernel events"}))
    elif rank == 0:
        per_gpu = args.particles
        kernels = {}
        for name, (ms, launches) in timing.items():
            if launches == 0:
                continue
            avg_us = 1000.0 * ms / launches
            gbps = ALGO_BYTES.get(name, 0) * per_gpu / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
            kernels[name] = {"avg_us": round(avg_us, 3), "launches": launches, "us_per_tick": round(1000.0 * ms / args.steps, 3),
                             "algo_bytes_per_particle": ALGO_BYTES.get(name, 0), "achieved_GBps": round(gbps, 1)}
        # BASELINE.json's metric is "% HBM-BW roofline in force kernel": the roofline object describes the fused
        # force + integrate kernel (pass B).  Pass A takes about the same time per tick; it, the pair and the
        # whole tick are reported next to it.
        dom = "force_integrate"
        tick_us = sum(k["us_per_tick"] for k in kernels.values())  # kernels that run once in a while count by their share
        pass_a = "neighbors_density" if "neighbors_density" in kernels else "density"
        force_us = kernels[pass_a]["avg_us"] + kernels["force_integrate"]["avg_us"]
        traffic, traffic_src = measured_traffic(per_gpu, dom)
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBps"], "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(kernels[dom]["achieved_GBps"] / HBM_PEAK_GBPS, 5), "traffic": traffic,
            "traffic_source": traffic_src,
            "note": "this path is not HBM bound on MI355X: at this size kernels are chains of cold-L2 misses plus "
                    "fp64 VALU issue, from ~1M particles on fp64 VALU issue alone (DESIGN.md section 6)",
            "measured_over": f"a replay of the same {args.warmup}+{args.steps} ticks from the same initial state right after the timed region, HIP events around every launch",
            "algorithmic_bytes_per_launch": ALGO_BYTES[dom] * per_gpu,
            "avg_launch_us": kernels[dom]["avg_us"],
            "pass_a": {"kernel": pass_a, "bytes_per_particle": ALGO_BYTES[pass_a], "us": kernels[pass_a]["avg_us"],
                       "achieved_GBps": kernels[pass_a]["achieved_GBps"],
                       "frac": round(kernels[pass_a]["achieved_GBps"] / HBM_PEAK_GBPS, 5)},
            "force_pair": {"kernels": f"{pass_a} + force_integrate", "bytes_per_particle": FORCE_BYTES,
                           "us": round(force_us, 3),
                           "achieved_GBps": round(FORCE_BYTES * per_gpu / (force_us * 1e-6) / 1e9, 1),
                           "frac": round(FORCE_BYTES * per_gpu / (force_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5)},
            "whole_tick": {"bytes_per_particle": TICK_BYTES, "kernel_us_sum": round(tick_us, 3),
                           "achieved_GBps": round(TICK_BYTES * per_gpu / (tick_us * 1e-6) / 1e9, 1),
                           "frac": round(TICK_BYTES * per_gpu / (tick_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5)},
        }
        line = {
            "metric": "particle-steps/sec", "value": n_total * args.steps / elapsed, "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.particles} synthetic uniform particles per GPU ({n_total} total), "
                                   f"wave_machine.yaml world, d=sqrt(12/(pi*P)) (~12 neighbors), "
                                   f"collider noise 0.1 ({args.noise} RNG)",
                       "particles_per_gpu": args.particles, "particles_total": n_total, "live_after_run": int(n_live),
                       "parallelism": "single GPU" if world == 1 else f"{world} x-slabs, halo exchange per tick"},
            "roofline": roofline, "kernels": kernels,
        }
        if device_flags:
            line["device_flags"] = sorted(set(device_flags))
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
