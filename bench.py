#!/usr/bin/env python3
"""Throughput of the SandCrate particle update on MI355X: particle-steps per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P_PER_GPU]

Workload.  N=1: BASELINE.json configs[2], the largest single-GPU configuration --
1,048,576 synthetic uniformly seeded particles in the wave_machine.yaml world (its coefficients, both rigid
bodies incl. the motored wall, no particle source), particle diameter d = sqrt(12 / (pi P_total)) so that a
particle has ~12 neighbors, dt scaled with d, collider_noise_level 0.1 from a counter-based device RNG.
N=4 / N=8: configs[3] / configs[4] (4,194,304 / 16,777,216 particles in all, slabs of rows, halo exchange per tick);
N=2: 1,048,576 per GPU.  A step is one `physics_tick` of all particles; state is resident in HBM before the
timed region and nothing is read back inside it.  The W+K-step measurement is repeated (--repeats, default 5)
from the same initial state; `value` is the median repetition, all repetitions are listed.  A second
context of the same size (N > 1: a single-domain one of a slab's size, on every rank) runs --clock-warmup ticks (default 100) right before every repetition's W warm-up steps: the
timed region is a few milliseconds behind an upload and would otherwise find the GPU's clocks down.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline      SURVEY.md M4's force-pair figure: 128 algorithmic bytes per particle / (pass A + pass B) measured
                with HIP events on the kernels' stream, against the 8 TB/s HBM peak; pass A, pass B and the whole
                tick as sub-objects; `fp64_issue`: the VALU issue roofline from committed SQ counters
  kernels       time and algorithmic GB/s of every kernel of the tick
  cpu_baseline  the oracle's loop-structured tick (stands for the reference's NumPy path, which cannot
                travel to the GPU box) timed on this host, one core, on a bounded sample (262,144 particles)
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


def latest_profile(pattern: str):
    """The newest round's committed profile matching profiles/<pattern> (r03_... before r02_...), or None."""
    found = sorted((ROOT / "profiles").glob(pattern), reverse=True)
    return found[0] if found else None


def measured_traffic(particles_per_gpu: int, kernel: str):
    """HBM-side bytes per launch of `kernel` from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, calibrated on a float64 copy: scripts/collect_traffic.sh -> profiles/r01_traffic_<N>.json).
    bench.py cannot collect hardware counters itself; it reports the committed measurement of the same
    workload, or None when there is none for this size."""
    path = latest_profile(f"r*_traffic_{particles_per_gpu}.json")
    if path is None:
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


# BASELINE.json configs: [2] 1M particles on one GPU (the largest single-GPU configuration, the N=1 workload),
# [3] 4M over 4 GPUs, [4] 16M over 8 GPUs with the wave_machine motored wall; N=2 keeps [2]'s 1M per GPU.
TOTAL_BY_GPUS = {1: 1048576, 2: 2097152, 4: 4194304, 8: 16777216}


def fp64_issue(particles_per_gpu: int, kernels: dict):
    """VALU issue roofline of the force pair from the committed SQ counters (profiles/r02_sq_<N>.json, written by
    scripts/collect_sq.sh on the same workload): wave-instructions issued per launch against what 1,024 SIMDs can
    issue in the measured kernel time.  None when no counter file exists for this size."""
    path = latest_profile(f"r*_sq_{particles_per_gpu}.json")
    if path is None:
        return None
    data = json.loads(path.read_text())
    out = {"source": f"profiles/{path.name}: {data.get('source', '')}", "simds": 1024, "clock_GHz": 2.4}
    for name in ("neighbors_density", "force_integrate"):
        k, t = data["kernels"].get(name), kernels.get(name)
        if not k or not t:
            continue
        busy_cycles = 4.0 * k["SQ_ACTIVE_INST_VALU"] / 1024  # quad-cycles summed over SIMDs -> cycles per SIMD
        out[name] = {"valu_wave_insts_per_particle": round(k["SQ_INSTS_VALU"] / particles_per_gpu, 2),
                     "valu_busy_us_at_2.4GHz": round(busy_cycles / 2400.0, 2),
                     "frac_of_kernel_time": round(busy_cycles / 2400.0 / t["avg_us"], 3)}
    return out


def rocprof_pair(particles_per_gpu: int):
    """The force pair by the committed rocprofv3 kernel trace of this very command (scripts/profile_round.sh ->
    scripts/summarize_trace.py -> profiles/rNN_kernel_calls_<particles>.json): per-call means over the calls with the
    workload's grid only -- the --stats CSV of the same run averages the 4,096-particle launches of the primer below
    into pass B.  The profiler's durations carry no event overhead (the HIP events of the live measurement add ~2 us per
    kernel); the profiler runs are taken without the clock warm-up of the live measurement (--clock-warmup 0: under the
    profiler the second context's one-time queue set-up lands inside one kernel's duration).  None when no summary is
    committed for this size."""
    path = latest_profile(f"r*_kernel_calls_{particles_per_gpu}.json")
    if path is None:
        return None
    pair = json.loads(path.read_text()).get("pair")
    if not pair:
        return None
    return {"source": f"profiles/{path.name} (rocprofv3 --kernel-trace of this command with --clock-warmup 0, per-call means, primer left out)",
            "pass_a_us": pair["pass_a_us"], "pass_b_us": pair["pass_b_us"], "avg_launch_us": pair["avg_launch_us"],
            "achieved_GBps": pair["achieved_GBps"], "frac": pair["frac"]}


def regimes(make_sim, settle, per_gpu: int):
    """The same workload beyond the timed region: it heats up (12 neighbors per particle is far denser than this
    fluid's equilibrium) and from some tick on (about 150 at 1,048,576 particles) half of the particles sit in cells
    of thousands along the walls.  One run of 475 ticks, wall time per tick over five windows;
    the headline `value` is the first, uniform one."""
    sim = make_sim()
    out = {}
    done = 0
    for name, upto in (("ticks_0_4_warmup", 5), ("ticks_5_24_uniform", 25), ("ticks_25_104", 105), ("ticks_105_124", 125),
                       ("ticks_125_424", 425), ("ticks_425_474_pile_up", 475)):
        t0 = time.perf_counter()
        sim.run(upto - done)
        settle(sim)
        out[name] = {"ms_per_step": round(1000.0 * (time.perf_counter() - t0) / (upto - done), 5)}
        done = upto
    total = sum(out[k]["ms_per_step"] * n for k, n in (("ticks_5_24_uniform", 20), ("ticks_25_104", 80)))
    out["ticks_5_104"] = {"ms_per_step": round(total / 100.0, 5), "particle_steps_per_s": round(per_gpu * 100.0 / (total / 1000.0), 1)}
    # the sustained figure: everything after the warm-up steps of this one run, pile-up regime included
    spans = (("ticks_5_24_uniform", 20), ("ticks_25_104", 80), ("ticks_105_124", 20), ("ticks_125_424", 300), ("ticks_425_474_pile_up", 50))
    total = sum(out[k]["ms_per_step"] * n for k, n in spans)
    out["sustained_ticks_5_474"] = {"ms_per_step": round(total / 470.0, 5), "particle_steps_per_s": round(per_gpu * 470.0 / (total / 1000.0), 1)}
    del out["ticks_0_4_warmup"]
    return out


def drop_in_ticks(ticks: int = 300):
    """The path the reference's viewer calls -- `Crate.physics_tick()` once per frame on config/wave_machine.yaml as
    shipped (particle source active, collider noise from NumPy's MT19937 stream) -- timed tick by tick with the
    stream generated on the device (noise="host", no per-tick synchronisation) and with the host drawing it
    (noise="host-sync": count readback + np.random + upload every tick)."""
    import sand_crate_amd as sc
    out = {"workload": f"config/wave_machine.yaml from tick 0, {ticks} x physics_tick()"}
    for mode in ("host", "host-sync"):
        crate = sc.Crate(sc.load_config(ROOT / "config" / "wave_machine.yaml").world_config, noise=mode)
        for _ in range(20):
            crate.physics_tick()
        crate.synchronize()
        t0 = time.perf_counter()
        for _ in range(ticks):
            crate.physics_tick()
        crate.synchronize()
        dt = time.perf_counter() - t0
        out[mode] = {"ms_per_tick": round(1000.0 * dt / ticks, 4), "particles_at_end": int(crate.particle_count)}
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=0,
                    help="particles per GPU (default: BASELINE.json's configuration for this GPU count: 1,048,576 per GPU "
                         "at 1, 2 and 4 GPUs, 2,097,152 per GPU -- 16,777,216 in all -- at 8)")
    ap.add_argument("--clock-warmup", type=int, default=100,
                    help="ticks a second context of the same size runs before each repetition's warm-up steps (GPU clocks; "
                         "N > 1: a single-domain context of a slab's size on every rank; 0: off)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the W+K-step measurement is repeated this many times from the same initial state; `value` is the "
                         "median repetition (each repetition times exactly K steps)")
    ap.add_argument("--cpu-sample", type=int, default=262144,
                    help="particles in the CPU baseline tick (0 = skip); 262,144 is one tick of ~13 s.  N > 1: rank 0 times "
                         "a quarter of it (65,536 particles, ~3 s) behind the timed region while the other ranks wait")
    ap.add_argument("--noise", default="counter", choices=["counter", "none"])
    ap.add_argument("--slab-axis", default="y", choices=["x", "y"],
                    help="N > 1: cut the domain into slabs of rows (y: the halo bands are the ends of the sorted order, the "
                         "halo overlap costs least) or of columns (x)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: every rank uses cuda:0 and the gloo backend (halo staged through the "
                         "host), to exercise the N > 1 code path on a one-GPU box; the numbers mean nothing")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="no replay with HIP events around the kernels (no roofline in the output)")
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
    n_total = args.particles * world if args.particles > 0 else TOTAL_BY_GPUS.get(world, 1048576 * world)
    per_gpu = n_total // world

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

    slab_sim = []
    band_mode = {}
    # The contract workload piles up along floor and ceiling -- with slabs of rows into the first and the last slab --, so a
    # window longer than the driver's (5 + 20 ticks: still uniform) re-derives the cuts from the global histogram now and
    # then (one small all-reduce; SlabCrate.rebalance_every).  In the driver's window the cuts stay where they are.
    rebalance_every = 50 if world > 1 and args.warmup + args.steps > 100 else 0

    def pick_band_mode(sim):
        """Halo overlap with slabs of rows: the force kernel either runs as two launches with an event in between or as
        one launch whose band blocks release the side stream through a flag that a one-thread kernel polls
        (sc_set_band_flag: cheaper, unless the side stream shares a hardware queue with the context's stream -- then every
        tick costs the poll's time-out).  Both are timed here on a few untimed ticks and the faster one is kept; all ranks
        take the same decision (the slowest rank's time counts)."""
        if not (sim.overlap and sim.axis == "y"):
            return
        from sand_crate_amd._native import NativeError
        took = {}
        where = "cpu" if args.rehearse_on_one_gpu else f"cuda:{local_rank}"

        def over_ranks(value, op):
            t = torch.tensor([value], dtype=torch.float64, device=where)
            dist.all_reduce(t, op=op)
            return float(t.item())

        for flag in (True, False):
            sim.reload(p, v)
            seconds, ok = float("inf"), True
            try:
                ok = sim.set_band_flag(flag) == flag
                if ok:
                    sim.run(3)
                    sim.synchronize()
            except (NativeError, RuntimeError):
                ok = False
            # a rank that gave up must not leave its neighbors waiting for the messages of the timed ticks: go on only
            # if every rank got this far (a device condition surfaces in synchronize(), after all ticks were enqueued,
            # so the exchanges of the ticks above completed on every rank)
            if over_ranks(1.0 if ok else 0.0, dist.ReduceOp.MIN) > 0.5:
                try:
                    barrier()
                    t0 = time.perf_counter()
                    sim.run(8)
                    sim.synchronize()
                    seconds = time.perf_counter() - t0
                except (NativeError, RuntimeError):
                    pass
            took[flag] = over_ranks(min(seconds, 1e9), dist.ReduceOp.MAX)
        sim.reload(p, v)
        if min(took.values()) >= 1e9:  # neither form got through its ticks: exchange on the context's own stream
            sim.backend.set_overlap(False)
            sim.overlap = False
        sim.set_band_flag(took[True] < took[False])
        band_mode.update({"one launch + flag": round(1e3 * took[True] / 8, 4), "two launches": round(1e3 * took[False] / 8, 4),
                          "kept": "one launch + flag" if sim.band_flag else "two launches" if sim.overlap else "no overlap"})

    def make_sim():
        import copy
        w = copy.deepcopy(wc)
        if world > 1:  # one communicator and one set of halo buffers for the whole run; the state is uploaded again
            if slab_sim:
                slab_sim[0].reload(p, v)
            else:
                slab_sim.append(SlabCrate(w, p, v, device=local_rank, noise=args.noise, noise_seed=1, axis=args.slab_axis,
                                          rebalance_every=rebalance_every))
                pick_band_mode(slab_sim[0])
            return slab_sim[0]
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

    device_flags = []

    def settle(s):
        """Synchronise; a condition the device flagged (a particle lost to NaN, a halo buffer that
        overflowed) is reported in the JSON line instead of aborting the measurement."""
        from sand_crate_amd._native import NativeError
        try:
            s.synchronize()
        except NativeError as err:
            device_flags.append(str(err))

    # ---- timed region, repeated: W untimed warm-up steps, then exactly K steps bracketed by barrier +
    # synchronize on both sides, no per-kernel events; every repetition starts from the same initial state.
    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The timed region is a few milliseconds and every repetition starts behind an upload, with the GPU idle: a second
    # context of the same size runs CLOCK_WARMUP_TICKS ticks right before each repetition's W warm-up steps, so that the
    # timed steps find the clocks where a longer run has them (repetitions used to get faster from the first to the
    # last by 2-3 %).  Its work is over before the timed region starts (device-wide synchronize below).
    heater = None
    if args.clock_warmup > 0:
        if world == 1:
            heater = make_sim()
        else:  # every rank heats its own GPU with a single-domain workload of a slab's size: no communication
            hw, _ = world_for(per_gpu)
            heater = sc.Crate(hw, device=local_rank, noise=args.noise, noise_seed=1, capacity=per_gpu + 1024)
            heater.particles, heater.particle_velocities = synthetic_state(per_gpu)
    reps = []
    sim = None
    for _ in range(max(1, args.repeats)):
        sim = None  # release the previous repetition's device memory before allocating again
        sim = make_sim()
        if heater is not None:
            heater.run(args.clock_warmup)
            settle(heater)
        sim.run(args.warmup)
        settle(sim)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        settle(sim)
        t0 = time.perf_counter()
        sim.run(args.steps)
        settle(sim)
        torch.cuda.synchronize()
        barrier()
        reps.append(max_over_ranks(time.perf_counter() - t0))
    order = sorted(reps)
    elapsed = order[(len(order) - 1) // 2]  # the median repetition (lower middle for an even count)
    transport = getattr(sim, "transport", None)

    # ---- kernel durations: the same ticks replayed from the same initial state, every launch bracketed by two
    # HIP events on the stream the kernels run on.  Kept out of the timed region because the event records cost
    # wall time; kernel durations themselves are unaffected.
    timing = {}
    n_live = sim.particle_count if world == 1 else sim.global_particle_count()
    if not args.no_kernel_events:
        sim = None
        sim = make_sim()  # same initial state, same ticks as the timed region
        eng = sim.engine
        if heater is not None:  # ... and the same clocks
            heater.run(args.clock_warmup)
            settle(heater)
        sim.run(args.warmup)
        settle(sim)
        eng.reset_timing()
        eng.enable_timing(True)
        if world > 1:
            sim.time_exchanges(True)
        sim.run(args.steps)
        settle(sim)
        eng.enable_timing(False)
        timing = eng.timing()
    heater = None
    halo_by_rank = None
    if world > 1:  # what every rank's halo exchange moved (and, from the replay above, took)
        import torch.distributed as dist
        halo_by_rank = [None] * world
        dist.all_gather_object(halo_by_rank, slab_sim[0].exchange_stats())
        slab_sim[0].time_exchanges(False)

    if world == 1:
        scaling = "weak"
    else:  # BASELINE.json's configurations: 1,048,576 per GPU at 1, 2 and 4 GPUs, 2,097,152 per GPU (configs[4]) at 8
        why = f"BASELINE.json's configuration for {world} GPUs" if args.particles <= 0 else "--particles"
        scaling = "weak" if per_gpu == TOTAL_BY_GPUS[1] else (f"weak, with {per_gpu / TOTAL_BY_GPUS[1]:g}x the per-GPU work of N=1 "
                                                             f"({per_gpu} particles per GPU: {why})")
    base = {"metric": "particle-steps/sec", "value": n_total * args.steps / elapsed, "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": {"count": len(reps), "ms_per_step": [round(1000.0 * r / args.steps, 5) for r in reps],
                        "min": round(1000.0 * order[0] / args.steps, 5), "max": round(1000.0 * order[-1] / args.steps, 5),
                        "value_is": "median repetition",
                        "clock_warmup_ticks_before_each": args.clock_warmup},
            "config": {"workload": f"{per_gpu} synthetic uniform particles per GPU ({n_total} total), "
                                   f"wave_machine.yaml world incl. the motored wall, d=sqrt(12/(pi*P)) (~12 neighbors), "
                                   f"collider noise 0.1 ({args.noise} RNG), ticks {args.warmup}..{args.warmup + args.steps - 1}",
                       "particles_per_gpu": per_gpu, "particles_total": n_total, "live_after_run": int(n_live),
                       "parallelism": "single GPU" if world == 1 else f"{world} slabs of {'rows' if args.slab_axis == 'y' else 'columns'}, halo exchange per tick",
                       "transport": transport, "halo_overlap": bool(slab_sim[0].overlap) if world > 1 else None,
                       "halo_overlap_band_mode_ms_per_tick": band_mode or None,
                       "rebalance_every": rebalance_every if world > 1 else None, "halo_by_rank": halo_by_rank}}
    if rank == 0 and args.no_kernel_events:
        base["note"] = "no per-kernel events"
        print(json.dumps(base))
    elif rank == 0:
        kernels = {}
        for name, (ms, launches) in timing.items():
            if launches == 0:
                continue
            avg_us = 1000.0 * ms / launches
            gbps = ALGO_BYTES.get(name, 0) * per_gpu / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
            kernels[name] = {"avg_us": round(avg_us, 3), "launches": launches, "us_per_tick": round(1000.0 * ms / args.steps, 3),
                             "algo_bytes_per_particle": ALGO_BYTES.get(name, 0), "achieved_GBps": round(gbps, 1)}
        # BASELINE.json's metric is "% HBM-BW roofline in force kernel"; SURVEY.md section 8d (M4) defines it over
        # the two force kernels together: 128 B per particle-step / (t_passA + t_passB).  That pair is the headline
        # `frac`; pass A, pass B and the whole tick are reported next to it.
        tick_us = sum(k["us_per_tick"] for k in kernels.values())  # kernels that run once in a while count by their share
        pass_a = "neighbors_density" if "neighbors_density" in kernels else "density"
        dom = "force_integrate"
        force_us = kernels[pass_a]["avg_us"] + kernels[dom]["avg_us"]
        pair_gbps = FORCE_BYTES * per_gpu / (force_us * 1e-6) / 1e9
        traffic_a, traffic_src = measured_traffic(per_gpu, pass_a)
        traffic_b, _ = measured_traffic(per_gpu, dom)
        traffic = traffic_a + traffic_b if traffic_a is not None and traffic_b is not None else None

        def sub(name):
            return {"kernel": name, "bytes_per_particle": ALGO_BYTES[name], "us": kernels[name]["avg_us"],
                    "achieved_GBps": kernels[name]["achieved_GBps"],
                    "frac": round(kernels[name]["achieved_GBps"] / HBM_PEAK_GBPS, 5),
                    "traffic": measured_traffic(per_gpu, name)[0]}

        roofline = {
            "bound": "hbm", "kernel": f"{pass_a} + {dom} (the force pair of SURVEY.md M4)",
            "achieved": round(pair_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(pair_gbps / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": FORCE_BYTES * per_gpu, "avg_launch_us": round(force_us, 3),
            "measured_over": f"a replay of the same {args.warmup}+{args.steps} ticks from the same initial state right "
                             f"after the timed region, HIP events around every launch on the kernels' own stream",
            "pass_a": sub(pass_a), "pass_b": sub(dom),
            "whole_tick": {"bytes_per_particle": TICK_BYTES, "kernel_us_sum": round(tick_us, 3),
                           "achieved_GBps": round(TICK_BYTES * per_gpu / (tick_us * 1e-6) / 1e9, 1),
                           "frac": round(TICK_BYTES * per_gpu / (tick_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5)},
            "fp64_issue": fp64_issue(per_gpu, kernels),
            "rocprof": rocprof_pair(per_gpu),
        }
        line = dict(base)
        line["roofline"] = roofline
        line["kernels"] = kernels
        if device_flags:
            line["device_flags"] = sorted(set(device_flags))
        if world == 1:
            line["regimes"] = regimes(make_sim, settle, per_gpu)
            if per_gpu != 262144:  # BASELINE.json configs[1] heats up sooner (the tick doubles from tick ~105 on)
                def small_sim():
                    import copy
                    sw, _ = world_for(262144)
                    s_ = sc.Crate(copy.deepcopy(sw), device=local_rank, noise=args.noise, noise_seed=1, capacity=262144 + 1024)
                    s_.particles, s_.particle_velocities = synthetic_state(262144)
                    return s_
                line["regimes_262144"] = regimes(small_sim, settle, 262144)
            line["drop_in_physics_tick"] = drop_in_ticks()
        if args.cpu_sample > 0:
            # the reference NumPy path on this box's host cores, in the same run (N > 1: a smaller sample on rank 0, the
            # path is flat in the particle count -- SURVEY.md section 6)
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample if world == 1 else max(4096, args.cpu_sample // 4))
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
