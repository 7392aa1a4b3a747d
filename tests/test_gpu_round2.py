"""GPU tests added in round 2: fixes of the round-1 review (staging of uploaded ids, the host-side
particle bound in host-noise mode, free rigid bodies in `Crate.run`), called through the C ABI."""
import copy

import numpy as np
import pytest

from test_gpu_parity import synthetic, wave_world

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import sand_crate_amd
    return sand_crate_amd


def test_upload_with_ids_twice_within_one_staging_allocation(sc):
    """The second upload lies in (0.8, 1.0] of the staging size the first one allocated (1000 -> room for 1756):
    the ids must neither overrun the staging buffers nor race the append kernel (sc_upload_state_ids)."""
    eng = sc.Engine(4096)
    rs = np.random.RandomState(3)
    for n in (1000, 1700, 1756, 900):
        p = rs.rand(n, 2) * 0.9 + 0.05
        v = rs.rand(n, 2) - 0.5
        ids = rs.permutation(10 * n)[:n].astype(np.int64)
        eng.upload_with_ids(p, v, ids)
        gp, gv, _, gids = eng.download()
        order = np.argsort(ids)
        assert np.array_equal(gids, ids[order])
        assert np.array_equal(gp, p[order]) and np.array_equal(gv, v[order])
    eng.close()


def test_emit_remove_emit_without_downloads_keeps_the_capacity_bound(sc):
    """noise='host-sync': a source keeps emitting while the particles leave the box a few ticks later, and nobody reads the
    state.  The host-side bound of the stored count must follow the live count (sc_step_stats), not the total ever
    emitted -- otherwise sc_append_particles reports a capacity overflow although the box is almost empty."""
    wc = sc.load_config("config/wave_machine.yaml").world_config
    wc.coefficients["max_particles"] = 64
    wc.rigid_bodies = []  # no walls: everything leaves through y > 1 + r (crate.py:152)
    wc.particle_sources = [dict(radius=0.02, position=[0.5, 0.95], velocity=[0.0, 6.0], flow=4000, active_ticks=10 ** 9,
                                noise=0.01)]
    crate = sc.Crate(wc, noise="host-sync", capacity=256)
    emitted = []
    real_append = crate.engine.append
    crate.engine.append = lambda p, v: (emitted.append(len(p)), real_append(p, v))[1]
    for _ in range(300):
        crate.physics_tick()
    assert sum(emitted) > 4 * 256          # far more than the capacity went through the box ...
    assert crate.engine.capacity == 256    # ... which never had to grow
    assert 0 < crate.particle_count <= 64
    assert len(crate.particles) == crate.particle_count


def test_run_equals_ticks_with_a_free_body(sc):
    """A `free` rigid body accelerates under gravity after every tick (crate.py:311-314); Crate.run(k) must move the
    walls exactly like k physics_tick() calls."""
    n = 4000
    p, v, d = synthetic(n, seed=9)
    wc = wave_world(sc, d, 0.1)
    wc.rigid_bodies = copy.deepcopy(wc.rigid_bodies) + [
        {"free": {"name": "raft", "segments": [[[0.3, 0.0], [0.5, 0.0]]], "position": [0.0, 0.4],
                  "center_velocity": [0.05, 0.0]}}]
    a = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=3, capacity=n + 16)
    b = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=3, capacity=n + 16)
    for c in (a, b):
        c.particles = p
        c.particle_velocities = v
    a.run(6)
    for _ in range(6):
        b.physics_tick()
    assert np.array_equal(a.segments, b.segments)
    assert not np.array_equal(a.segments[-1], np.array([[0.3, 0.4], [0.5, 0.4]]))  # the raft did move
    pa, va, _, ida = a.engine.download()
    pb, vb, _, idb = b.engine.download()
    assert np.array_equal(ida, idb) and np.array_equal(pa, pb) and np.array_equal(va, vb)
    np.testing.assert_array_equal(a.rigid_bodies[-1].center_velocity, b.rigid_bodies[-1].center_velocity)


# ------------------------------------------------------------------ NumPy's global stream on the device (N2)
def scene(sc, name):
    return sc.load_config(f"config/{name}.yaml").world_config


@pytest.mark.parametrize("name,ticks", [("stirring_cup", 120), ("wave_machine", 90)])
def test_device_stream_equals_host_drawn_stream(sc, name, ticks):
    """noise="host" (sources and collider noise drawn on the device from NumPy's MT19937 state) against
    noise="host-sync" (the host draws the same stream with np.random): the same particles, bit for bit, tick
    after tick -- and the stream handed back to np.random stands where the host-drawn one stands."""
    dev = sc.Crate(scene(sc, name), noise="host")
    for _ in range(ticks):
        dev.physics_tick()
    pd, vd, prd, idd = dev.engine.download()
    dev.sync_host_rng()
    after_dev = np.random.rand(5)
    ref = sc.Crate(scene(sc, name), noise="host-sync")  # seeds np.random again (crate.py:22)
    for _ in range(ticks):
        ref.physics_tick()
    pr, vr, prr, idr = ref.engine.download()
    after_ref = np.random.rand(5)
    assert len(pd) == len(pr) > 50
    assert np.array_equal(idd, idr)
    assert np.array_equal(pd, pr) and np.array_equal(vd, vr) and np.array_equal(prd, prr)
    assert np.array_equal(after_dev, after_ref)
    assert dev.particle_count == ref.particle_count


def test_device_stream_noise_block_straddles_state_refills(sc):
    """The collider-noise block of one tick drawn on the device (k_rng_noise) for stream positions that make doubles
    straddle the 624-word state blocks, against the same tick with the host's np.random.rand block."""
    from sand_crate_amd import _native as N
    from test_gpu_parity import synthetic, wave_world
    n = 3000
    p, v, d = synthetic(n, seed=4)
    wc = wave_world(sc, d, 0.1)
    for skip in (0, 1, 311, 623):
        outs = []
        for device_stream in (True, False):
            np.random.seed(123)
            np.random.rand(skip)             # move the stream: odd counts leave an odd position
            if skip % 2:
                np.random.randint(0, 2 ** 32, dtype=np.uint64)  # one more 32-bit word: doubles no longer align with blocks
            crate = sc.Crate(copy.deepcopy(wc), noise="host-sync")
            eng = crate.engine
            name, key, pos, _, _ = np.random.get_state()
            if device_stream:
                eng.rng_set_state(key, pos)
            eng.upload(p, v)
            for b in crate.rigid_bodies:
                b.apply_velocity(crate.dt)
            crate._send_tick_inputs()
            eng.step_begin()
            if not device_stream:
                eng.set_noise_host(np.random.rand(eng.step_stats().neighbor_slots, 2))
            eng.step_finish()
            out = eng.download()
            state = eng.rng_get_state() if device_stream else np.random.get_state()[1:3]
            outs.append((out, state))
            eng.close()
        (a, sa), (b, sb) = outs
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.array_equal(sa[0], sb[0]) and sa[1] == sb[1]


def test_sources_on_the_device_respect_max_particles_and_capacity(sc):
    """noise="host": emission is capped at max_particles minus the stored count on the device (crate.py:142), the
    box never holds more, and the host-side bounds follow without a synchronising call per tick."""
    wc = sc.load_config("config/wave_machine.yaml").world_config
    wc.coefficients["max_particles"] = 64
    wc.rigid_bodies = []
    wc.particle_sources = [dict(radius=0.02, position=[0.5, 0.95], velocity=[0.0, 6.0], flow=4000, active_ticks=10 ** 9,
                                noise=0.01)]
    crate = sc.Crate(wc, noise="host", capacity=256)
    for _ in range(300):
        crate.physics_tick()
    assert crate.engine.capacity == 256
    assert 0 < crate.particle_count <= 64
    ids = crate.engine.download()[3]
    assert ids.max() > 1000 and len(np.unique(ids)) == len(ids)  # far more particles than the box holds went through


# ------------------------------------------------------------------ checkpoint / resume (N3)
@pytest.mark.parametrize("noise", ["host", "host-sync", "counter"])
def test_resume_from_checkpoint_is_bit_equal(sc, tmp_path, noise):
    """40 ticks in one go against 20 ticks, save, a NEW crate from the file, 20 more: particles, ids, pressure, walls
    and -- for the MT19937 modes -- the random stream continue exactly (wave_machine.yaml: a particle source that
    is still emitting and a motored wall)."""
    a = sc.Crate(scene(sc, "wave_machine"), noise=noise, noise_seed=3)
    for _ in range(40):
        a.physics_tick()
    want = a.engine.download()
    want_segments, want_draw = a.segments.copy(), None
    if noise != "counter":
        a.sync_host_rng()
        want_draw = np.random.rand(3)

    b = sc.Crate(scene(sc, "wave_machine"), noise=noise, noise_seed=3)
    for _ in range(20):
        b.physics_tick()
    b.gravity = np.array([0.0, 9.8])  # coefficients travel as they stand (here unchanged)
    b.save_checkpoint(tmp_path / "ck.npz")
    np.random.seed(99)  # whatever happens to the global stream in between must not matter
    c = sc.Crate.from_checkpoint(tmp_path / "ck.npz")
    assert c.tick == 20 and c.particle_count == b.particle_count
    assert np.array_equal(c.particles, b.particles) and np.array_equal(c.segments, b.segments)
    for _ in range(20):
        c.physics_tick()
    got = c.engine.download()
    assert c.tick == a.tick == 40
    for x, y in zip(got, want):
        assert np.array_equal(x, y)
    assert np.array_equal(c.segments, want_segments)
    if want_draw is not None:
        c.sync_host_rng()
        assert np.array_equal(np.random.rand(3), want_draw)


def test_checkpoint_transfer_overlaps_later_ticks(sc, tmp_path):
    """begin_checkpoint returns at once; ticks that follow do not disturb what it captured."""
    n = 20000
    p, v, d = synthetic(n, seed=8)
    crate = sc.Crate(wave_world(sc, d, 0.1), noise="counter", noise_seed=2, capacity=n + 16)
    crate.particles = p
    crate.particle_velocities = v
    crate.run(3)
    at_begin = crate.engine.download()
    segments = crate.segments.copy()
    crate.begin_checkpoint()
    crate.run(4)                      # runs while the snapshot travels
    crate.finish_checkpoint(tmp_path / "ck.npz")
    moved_on = crate.engine.download()
    assert not np.array_equal(moved_on[0], at_begin[0])
    back = sc.Crate.from_checkpoint(tmp_path / "ck.npz")
    assert back.tick == 3
    assert np.array_equal(back.particles, at_begin[0]) and np.array_equal(back.particle_velocities, at_begin[1])
    assert np.array_equal(back.segments, segments)
    back.run(4)
    for x, y in zip(back.engine.download(), moved_on):
        assert np.array_equal(x, y)
    with pytest.raises(RuntimeError):
        crate.finish_checkpoint(tmp_path / "none.npz")


# ------------------------------------------------------------------ the Forces block of the HUD (N4)
def test_force_monitor_matches_the_oracle_phases(sc):
    """Mean |dv| per force phase as the reference's ForceMonitor takes it (force_monitor.py:23-33 around
    crate.py:110-123), from the force kernel's side sums, against the oracle's per-phase velocities; and the monitor
    changes no result."""
    from oracle.scene import OracleCrate
    from oracle.tick import counter_noise_key, counter_noise_u01, tick_core
    from oracle.world import World
    n = 20000
    p, v, d = synthetic(n, seed=31, margin=0.0, vel=20.0)  # fast, up to the walls: every phase has work
    wc = wave_world(sc, d, 0.1)
    wc.coefficients["max_particles"] = n
    plain = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=6, capacity=n + 16)
    hud = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=6, capacity=n + 16)
    hud.show_forces(True)
    for c in (plain, hud):
        c.particles = p
        c.particle_velocities = v
        c.physics_tick()
    for x, y in zip(plain.engine.download(), hud.engine.download()):
        assert np.array_equal(x, y)
    orc = OracleCrate(World(wc.rigid_bodies, [], dict(wc.coefficients)))
    for b in orc.rigid_bodies:
        b.advance(orc.coef["dt"])
    from oracle.tick import remove_outside
    ids = np.arange(n)
    p1, v1, ids = remove_outside(p, v, orc.coef["particle_radius"], ids)
    out = tick_core(p1, v1, orc.segments, orc.body_states(), orc.coef, eta_u01=counter_noise_u01(ids, counter_noise_key(6, 0)))
    dt, g = orc.coef["dt"], np.asarray(orc.coef["gravity"], dtype=np.float64)
    norm = lambda a: float(np.mean(np.sqrt(a[:, 0] ** 2 + a[:, 1] ** 2)))  # noqa: E731
    want = {
        "tension": norm(out["v_after_tension"] - v1),
        "gravity": float(np.sqrt(((dt * g) ** 2).sum())),
        "pressure": norm(out["v_after_pressure"] - (out["v_after_tension"] + dt * g[None])),
        "viscosity": norm(out["v_after_viscosity"] - out["v_after_pressure"]),
        "wall_bounce": norm(out["v_after_bounce"] - out["v_after_viscosity"]),
        "continuous_collision": norm(out["v_after_bounce"] * out["ccd_factor"][:, None] - out["v_after_bounce"]),
    }
    assert want["wall_bounce"] > 0 and want["continuous_collision"] > 0
    for name, value in want.items():
        got = hud._force_ema[name] / 0.2  # the first tick of the EMA: 0.2 x the tick's mean
        assert got == pytest.approx(value, rel=1e-7, abs=1e-13), name
    text = hud.debug_prints
    assert "Forces:" in text and text.index("Timing:") < text.index("Forces:") < text.index("- dt:")
    for name in want:
        assert f"{name}:" in text
    hud.show_forces(False)
    hud.physics_tick()
    assert "Forces:" not in hud.debug_prints


def test_headless_driver_checkpoints_and_resumes(sc, tmp_path):
    """`python -m sand_crate_amd.main` with --checkpoint-every: the files it writes while ticking resume to the same
    final state as the uninterrupted run."""
    from sand_crate_amd.main import main
    whole = main("config/stirring_cup.yaml", tmp_path / "a", variants=1, ticks=60, checkpoint_every=20)
    files = sorted((tmp_path / "a" / "variant_00").glob("checkpoint_*.npz"))
    assert [f.name for f in files] == ["checkpoint_000020.npz", "checkpoint_000040.npz", "checkpoint_000060.npz"]
    rest = main("config/stirring_cup.yaml", tmp_path / "b", variants=1, ticks=20, resume=files[1], record_every=20)
    assert rest[0]["ticks"] == 60 and rest[0]["particles"] == whole[0]["particles"]
    end_a = sc.Crate.from_checkpoint(files[2])
    rec_b = np.load(tmp_path / "b" / "variant_00" / "state.npz")
    assert rec_b["ticks"].tolist() == [60]
    assert np.array_equal(rec_b["particles_0"], end_a.particles)


# ------------------------------------------------------------------ the pile-up regime
def pile_up_state(d, seed=21):
    """What the contract workload turns into: piles of thousands in single cells (spread over the cell, thin along a
    line, two cells wide), sparse particles in the rows beside them whose windows cover the piles, everything fast
    enough to change cells every tick, in random storage order."""
    rs = np.random.RandomState(seed)
    cell = lambda cx, cy: np.array([cx * d, cy * d])  # noqa: E731
    parts = [
        cell(20, 30) + rs.rand(3000, 2) * d * 0.98,                                         # a pile filling its cell
        cell(40, 12) + np.column_stack((rs.rand(2600) * d * 1.9, 0.93 * d + rs.rand(2600) * d * 0.05)),  # thin, 2 cells
        cell(60, 50) + np.column_stack((np.full(1500, 0.5 * d), rs.rand(1500) * d * 0.97)),  # exact x ties
        cell(18, 29) + rs.rand(700, 2) * np.array([6 * d, d]),                               # sparse rows beside them
        cell(18, 31) + rs.rand(700, 2) * np.array([6 * d, d]),
        cell(38, 11) + rs.rand(700, 2) * np.array([6 * d, d]),
        cell(38, 13) + rs.rand(700, 2) * np.array([6 * d, d]),
        rs.rand(4000, 2) * 0.9 + 0.05,
    ]
    p = np.vstack(parts)
    p = p[rs.permutation(len(p))]
    v = (rs.rand(len(p), 2) - 0.5) * (2.5 * d / (0.002 * d / 0.01))  # up to ~1.2 cells per tick
    return p, v


@pytest.mark.parametrize("noise", ["none", "counter", "host"])
def test_pile_up_ticks_match_the_oracle(sc, noise):
    """Consecutive ticks of a pile-up state -- big buckets sorted chunk by chunk, scrambled waves grouped by cell,
    tiles beyond the LDS budget with strided threads, cooperative walks with probe skips, renumbered table ranges for
    pass B and (host mode) for the density pass as its own launch -- each against the oracle from the same state."""
    from oracle.scene import OracleCrate
    from oracle.tick import counter_noise_key, counter_noise_u01, remove_outside, tick_core
    from oracle.world import World
    d = 0.012
    p, v = pile_up_state(d)
    n = len(p)
    wc = wave_world(sc, d, 0.0 if noise == "none" else 0.1)
    wc.coefficients["max_particles"] = n
    crate = sc.Crate(wc, noise=noise, noise_seed=9, capacity=n + 64)  # seeds NumPy's generator like the reference
    crate.particles = p
    crate.particle_velocities = v
    orc = OracleCrate(World(wc.rigid_bodies, [], dict(wc.coefficients)))  # ... and so does this
    ids = np.arange(n)
    for t in range(2):
        crate.physics_tick()
        for b in orc.rigid_bodies:
            b.advance(orc.coef["dt"])
        p, v, ids = remove_outside(p, v, orc.coef["particle_radius"], ids)
        if noise == "none":
            eta = None
        elif noise == "counter":
            eta = counter_noise_u01(ids, counter_noise_key(9, t))
        else:
            eta = lambda total: np.random.rand(total, 2)  # noqa: E731  the stream the device continues (crate.py:169)
        out = tick_core(p, v, orc.segments, orc.body_states(), orc.coef, eta_u01=eta)
        gp, gv, gpr, gids = crate.engine.download()
        assert np.array_equal(gids, ids)
        assert out["neighbor_counts"].max() == 20 and (out["neighbor_counts"] < 20).any()
        np.testing.assert_allclose(gp, out["particles"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gv, out["velocities"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(gpr, out["pressure"], rtol=1e-9, atol=1e-12)
        p, v = gp, gv


def test_lists_that_reach_across_more_than_a_row_can_number(sc):
    """A neighbor-table row holds twelve bits per entry: slots of the tile pass A publishes for pass B, up to 4,095.  Sparse
    particles in the rows beside a pile of 6,000 in one cell are within reach of a handful of its particles each, spread
    over the whole of the pile's x order -- the published ranges of their tiles hold far more than 4,095 entries, so those
    tiles keep their entries in the 32-bit table (csrc/sc_tiled.h: kRowSlotMax).  Two ticks against the oracle."""
    from oracle.scene import OracleCrate
    from oracle.tick import remove_outside, tick_core
    from oracle.world import World
    d = 0.012
    rs = np.random.RandomState(11)
    cell = lambda cx, cy: np.array([cx * d, cy * d])  # noqa: E731
    parts = [
        cell(30, 40) + rs.rand(6000, 2) * d * 0.98,                                         # the pile: one cell, full
        cell(29, 41) + np.column_stack((rs.rand(16) * 3 * d, 0.955 * d + rs.rand(16) * 0.02 * d)),  # above it, at the far side of their cells
        cell(29, 39) + np.column_stack((rs.rand(16) * 3 * d, 0.005 * d + rs.rand(16) * 0.02 * d)),  # below it, likewise
        rs.rand(1500, 2) * 0.9 + 0.05,
    ]
    n_pile = 6000
    p = np.vstack(parts)
    perm = rs.permutation(len(p))
    p = p[perm]
    v = (rs.rand(len(p), 2) - 0.5) * 0.2 * d / (0.002 * d / 0.01)
    n = len(p)
    wc = wave_world(sc, d, 0.0)
    wc.coefficients["max_particles"] = n
    crate = sc.Crate(wc, noise="none", capacity=n + 64)
    crate.particles = p
    crate.particle_velocities = v
    orc = OracleCrate(World(wc.rigid_bodies, [], dict(wc.coefficients)))
    ids = np.arange(n)
    row_of = perm  # the row of `parts` a particle came from, by id
    for t in range(2):
        crate.physics_tick()
        for b in orc.rigid_bodies:
            b.advance(orc.coef["dt"])
        p, v, ids = remove_outside(p, v, orc.coef["particle_radius"], ids)
        out = tick_core(p, v, orc.segments, orc.body_states(), orc.coef, eta_u01=None)
        if t == 0:  # the geometry does what the docstring says: the lists of the particles above the pile -- a handful of
            # sorted neighbors, one tile -- name pile particles more than 4,095 apart in the sorted order
            q = out["fixed_positions"]
            order = np.lexsort((q[:, 0], np.floor(q[:, 1] / d)))
            rank = np.empty(len(q), dtype=np.int64)
            rank[order] = np.arange(len(q))
            nbrs, cnt = out["neighbor_table"], out["neighbor_counts"]
            is_pile = row_of[ids] < n_pile  # (by position in the tick's arrays; upload order was shuffled)
            above = np.flatnonzero((np.floor(q[:, 1] / d) == 41) & (np.floor(q[:, 0] / d) >= 29) & (np.floor(q[:, 0] / d) < 32))
            named = np.concatenate([nbrs[i][:cnt[i]] for i in above])
            named = named[is_pile[named]]
            assert len(above) >= 8 and rank[named].max() - rank[named].min() > 4095
        gp, gv, gpr, gids = crate.engine.download()
        assert np.array_equal(gids, ids)
        np.testing.assert_allclose(gp, out["particles"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gv, out["velocities"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(gpr, out["pressure"], rtol=1e-9, atol=1e-12)
        p, v = gp, gv


def test_cell_tables_with_every_cell_on_one_slot(sc):
    """The workgroup cell tables of the pile-up variants (scatter and pass B's fused cell count: cell_tab_* in
    csrc/sc_kernels.h) on the worst input for their hash: hundreds of particles, each alone in a cell, whose cells all hash
    to ONE slot -- a workgroup of the sorted order then walks a single probing chain through its table -- beside the pile
    that makes the host launch those variants.  Three ticks against the oracle."""
    from oracle.scene import OracleCrate
    from oracle.tick import remove_outside, tick_core
    from oracle.world import World
    d = 0.0025  # 400 x 400 cells: ~300 of them per slot of the 512-slot table
    r = d / 2
    # the library's grid (build_world in csrc/sandcrate_hip.hip): local cell = (row - row0) * ncols + (col - col0)
    cmin, cmax = int(np.floor(-r / d)) - 3, int(np.floor((1 + r) / d)) + 3
    row0 = col0 = cmin - 1
    ncols = cmax - cmin + 3
    cx, cy = np.meshgrid(np.arange(12, 388), np.arange(12, 388))
    cx, cy = cx.ravel(), cy.ravel()
    cell = (cy - row0) * ncols + (cx - col0)
    slot = ((cell.astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(23)
    pick = np.flatnonzero(slot == np.bincount(slot.astype(np.int64)).argmax())
    assert len(pick) >= 256
    rs = np.random.RandomState(5)
    singles = np.column_stack((cx[pick] + 0.5, cy[pick] + 0.5)) * d
    pile = np.array([200 * d, 100 * d]) + rs.rand(400, 2) * d * 0.98
    p = np.vstack((singles, pile))
    p = p[rs.permutation(len(p))]
    p0 = p.copy()
    v = np.zeros_like(p)
    n = len(p)
    wc = wave_world(sc, d, 0.0)
    wc.coefficients["max_particles"] = n
    crate = sc.Crate(wc, noise="none", capacity=n + 64)
    crate.particles = p
    crate.particle_velocities = v
    orc = OracleCrate(World(wc.rigid_bodies, [], dict(wc.coefficients)))
    ids = np.arange(n)
    for t in range(3):
        crate.physics_tick()
        for b in orc.rigid_bodies:
            b.advance(orc.coef["dt"])
        p, v, ids = remove_outside(p, v, orc.coef["particle_radius"], ids)
        out = tick_core(p, v, orc.segments, orc.body_states(), orc.coef, eta_u01=None)
        gp, gv, gpr, gids = crate.engine.download()
        assert np.array_equal(gids, ids)
        np.testing.assert_allclose(gp, out["particles"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gv, out["velocities"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(gpr, out["pressure"], rtol=1e-9, atol=1e-12)
        p, v = gp, gv
    # the same ticks in one run() -- pass B's fused cell count (the table without a barrier) instead of k_wall_bin's --
    # end in the same state, bit for bit
    wc2 = wave_world(sc, d, 0.0)
    wc2.coefficients["max_particles"] = n
    fused = sc.Crate(wc2, noise="none", capacity=n + 64)
    fused.particles = p0
    fused.particle_velocities = np.zeros_like(p0)
    fused.run(1)
    fused.synchronize()  # (the host has seen the first tick's big bucket: the next call launches the grouping variants)
    fused.run(2)
    fp, fv, fpr, fids = fused.engine.download()
    assert np.array_equal(fids, gids)
    assert np.array_equal(fp, gp) and np.array_equal(fv, gv) and np.array_equal(fpr, gpr)


def test_pile_up_tick_does_not_depend_on_the_storage_order(sc, tmp_path):
    """The contract workload run into its pile-up regime (262,144 particles, tick 240: cells of thousands, particles
    crossing cells every tick).  A context restored from a checkpoint holds the particles in upload order instead of
    the running context's storage order, so every order-dependent choice inside the tick differs -- arrival order in
    the buckets, which are then sorted chunk by chunk, runs of equal cells and the grouping of scrambled waves, the
    atomics of the cell counts -- and the state after the tick must be the same bit for bit."""
    import bench
    n = 262144
    wc, d = bench.world_for(n)
    p, v = bench.synthetic_state(n)
    crate = sc.Crate(copy.deepcopy(wc), noise="counter", noise_seed=1, capacity=n + 1024)
    crate.particles = p
    crate.particle_velocities = v
    crate.run(240)
    crate.save_checkpoint(tmp_path / "pile.npz")
    crate.run(1)
    after = crate.engine.download()
    cells = np.floor(after[0][:, 1] / d).astype(np.int64) * 100000 + np.floor(after[0][:, 0] / d).astype(np.int64)
    assert np.unique(cells, return_counts=True)[1].max() > 2048  # buckets of several sorting chunks
    again = sc.Crate.from_checkpoint(tmp_path / "pile.npz", capacity=n + 1024)
    again.run(1)
    for a, b in zip(after, again.engine.download()):
        assert np.array_equal(a, b, equal_nan=True)


def test_tiles_just_over_the_force_kernels_budget_match_the_oracle(sc):
    """A fluid nine times denser than the contract workload (34 particles per cell): the candidate ranges of a block
    hold 3 x (256 + two or three cells) = 960..1100 entries -- staged in LDS by the search (budget 1100) but beyond the
    force kernel's 960 -- so the search renumbers its table to the ranges the lists reach and the force kernel stages
    those.  Ticks against the oracle."""
    from oracle.scene import OracleCrate
    from oracle.tick import counter_noise_key, counter_noise_u01, remove_outside, tick_core
    from oracle.world import World
    n = 30000
    p, v, d = synthetic(n, seed=77, margin=0.02, vel=0.1)
    d *= 3.0
    wc = wave_world(sc, d, 0.1)
    wc.coefficients["max_particles"] = n
    crate = sc.Crate(wc, noise="counter", noise_seed=5, capacity=n + 64)
    crate.particles = p
    crate.particle_velocities = v
    orc = OracleCrate(World(wc.rigid_bodies, [], dict(wc.coefficients)))
    ids = np.arange(n)
    for t in range(2):
        crate.physics_tick()
        for b in orc.rigid_bodies:
            b.advance(orc.coef["dt"])
        p, v, ids = remove_outside(p, v, orc.coef["particle_radius"], ids)
        out = tick_core(p, v, orc.segments, orc.body_states(), orc.coef, eta_u01=counter_noise_u01(ids, counter_noise_key(5, t)))
        gp, gv, gpr, gids = crate.engine.download()
        assert np.array_equal(gids, ids)
        if t == 0:
            per_cell = n / ((1 - 0.04) / d) ** 2
            assert 30 < per_cell < 40 and out["neighbor_counts"].mean() > 19.5
        np.testing.assert_allclose(gp, out["particles"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gv, out["velocities"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(gpr, out["pressure"], rtol=1e-9, atol=1e-12)
        p, v = gp, gv


def test_bucket_sizes_around_every_ranking_path(sc):
    """k_reorder ranks a bucket from a window of 12 slots either side when both of its ends show there, from the
    bucket's keys in global memory up to 24, cooperatively above, and from k_sort_big's chunks (1024 slots) above 96:
    buckets of every size around those edges, side by side in one row (so that the windows hold other buckets' keys),
    with exact ties in x, at the very start and the very end of the sorted order -> the reference's (row, x, id)
    order (oracle/neighbors.py: strip_sort, pinned by the golden cases)."""
    from oracle.neighbors import strip_sort
    rs = np.random.RandomState(17)
    d = 0.05
    sizes = [1, 2, 11, 12, 13, 14, 22, 23, 24, 25, 26, 40, 95, 96, 97, 98, 130, 1023, 1024, 1025, 2047, 2048, 2049, 3, 12, 13, 1]
    pts = []
    for row, shuffle in ((0, False), (3, True)):  # row 3: the same sizes in another order, ties everywhere
        order = rs.permutation(len(sizes)) if shuffle else np.arange(len(sizes))
        for k, j in enumerate(order):
            n = sizes[j]
            x = (k + rs.rand(n) * 0.999) * d
            if shuffle or n % 2 == 0:  # exact ties: a handful of distinct x per cell
                x = (k + np.round(rs.rand(n) * 5) / 5.01 * 0.999) * d
            pts.append(np.column_stack((x, (row + rs.rand(n) * 0.999) * d)))
    pts = np.vstack(pts)
    pts = pts[rs.permutation(len(pts))]
    rows, order, counts, table = sc.neighbor_search(pts, d)
    ref_rows, ref_order = strip_sort(pts, d)
    assert np.array_equal(order, ref_order)
    assert np.array_equal(rows, ref_rows)
    assert counts.max() <= 20 and len(table) == len(pts)  # the lists themselves: test_pile_up_buckets_sort_and_rank


# ------------------------------------------------------------------ checkpoints: the corners ADVICE.md (round 2) named
def test_checkpoint_of_a_slab_context_with_halo_overlap(sc):
    """A slab context whose side stream already exists (halo overlap creates it) takes a checkpoint all the same -- the
    snapshot's events and pinned buffers are created on their own -- and gives back what it holds."""
    from test_gpu_parity import synthetic, wave_world
    n = 30000
    p, v, d = synthetic(n, seed=12)
    eng = sc.Engine(n + 64)
    from sand_crate_amd import _native as N
    eng.set_noise_mode(N.NOISE_COUNTER, 5)
    crate_like = sc.Crate(wave_world(sc, d, 0.1), noise="counter", capacity=16)  # only for the tick inputs
    eng.set_slab(-2 ** 40, 2 ** 40, 3, False, False)   # one slab that owns everything
    eng.set_halo_overlap(True)                         # -> the side stream exists before the first checkpoint
    ids = np.arange(n, dtype=np.int64)[::-1].copy()    # any ids: they come back sorted
    eng.upload_with_ids(p, v, ids)
    eng.checkpoint_begin()
    snap = eng.checkpoint_finish()
    order = np.argsort(ids)
    assert np.array_equal(snap["ids"], ids[order])
    assert np.array_equal(snap["particles"], p[order]) and np.array_equal(snap["velocities"], v[order])
    eng.close()
    crate_like.engine.close()


def test_checkpoint_is_refused_while_a_promised_tick_is_pending(sc):
    """After sc_set_next_inputs promised the next tick the storage arrays hold that tick's wall pass already:
    sc_checkpoint_begin says so instead of capturing a state that would run the wall pass twice."""
    from sand_crate_amd import _native as N
    from test_gpu_parity import synthetic, wave_world
    n = 5000
    p, v, d = synthetic(n, seed=13)
    crate = sc.Crate(wave_world(sc, d, 0.1), noise="counter", noise_seed=1, capacity=n + 16)
    crate.particles = p
    crate.particle_velocities = v
    eng = crate.engine
    for b in crate.rigid_bodies:
        b.apply_velocity(crate.dt)
    now = crate._pack_tick_inputs()
    for b in crate.rigid_bodies:
        b.apply_velocity(crate.dt)
    nxt = crate._pack_tick_inputs()
    eng.tick(now, nxt)                                  # promises the tick after
    with pytest.raises(N.NativeError) as err:
        eng.checkpoint_begin()
    assert err.value.code == N.ERR_STATE
    eng.tick(nxt, None)                                 # the promised tick runs; nothing is pending any more
    eng.checkpoint_begin()
    assert len(eng.checkpoint_finish()["ids"]) == n


def big_flow_world(sc):
    wc = scene(sc, "wave_machine")
    wc.particle_sources = [dict(radius=0.3, position=[0.05, 0.95], velocity=[3, 0.0], flow=20000, noise=0.0, active_ticks=500)]
    return wc  # flow * dt = 40 > 30: NumPy's legacy binomial takes its BTPE branch (particle_source.py:18)


def test_device_stream_with_the_btpe_binomial(sc):
    """A source with flow * dt > 30 draws its count by BTPE; the device has that branch too (sc_rng.h), so noise="host"
    stays on the device -- no warning, no fallback -- and gives what the host-drawn stream gives, bit for bit."""
    import warnings
    dev = sc.Crate(big_flow_world(sc), noise="host")
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        for _ in range(40):
            dev.physics_tick()
    assert dev._noise == "host"
    got = dev.engine.download()
    dev.sync_host_rng()
    after_dev = np.random.rand(4)
    ref = sc.Crate(big_flow_world(sc), noise="host-sync")
    for _ in range(40):
        ref.physics_tick()
    want = ref.engine.download()
    assert len(want[3]) > 500 and want[3].max() > 1200  # ~40 particles per tick (many have left the box again)
    for x, y in zip(got, want):
        assert np.array_equal(x, y)
    assert np.array_equal(after_dev, np.random.rand(4))


def test_resume_after_the_fallback_to_the_host_stream(sc, tmp_path):
    """A crate that started with the stream on the device and fell back to noise="host-sync" (what a source with a time
    step above one half makes it do; here the fallback is taken by hand after the first tick) checkpoints the HOST
    stream -- the device's copy is stale by then -- and the resumed run continues the uninterrupted one bit for bit."""
    def start():
        crate = sc.Crate(big_flow_world(sc), noise="host")
        crate.physics_tick()
        with pytest.warns(RuntimeWarning, match="host-sync"):
            crate._fall_back_to_host_stream()
        assert crate._noise == "host-sync"
        return crate

    def run(crate, ticks):
        for _ in range(ticks):
            crate.physics_tick()

    a = start()
    run(a, 23)
    want = a.engine.download()
    b = start()
    run(b, 11)
    b.save_checkpoint(tmp_path / "ck.npz")
    np.random.seed(7)
    c = sc.Crate.from_checkpoint(tmp_path / "ck.npz")
    run(c, 12)
    for x, y in zip(c.engine.download(), want):
        assert np.array_equal(x, y)


# ------------------------------------------------------------------ the bucket scan on its own (round 3)
def test_bucket_scan_and_task_list_over_many_workgroups(tmp_path):
    """k_scan_cells by itself (scripts/scan_check.hip, compiled here): bucket starts == a host prefix sum and the list
    of k_sort_big's tasks == one task per 1024 slots of every bucket above 96, for 1 .. 2442 workgroups with no, a few
    and thousands of such buckets -- the parity tests' worlds have a handful of scan workgroups, the contract workload
    140, and a wrong task list (round 3: a wave total read from the wrong lane) shows only with many."""
    import shutil
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "scan_check"
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-w", f"-I{root / 'include'}",
                    f"-I{root / 'sand_crate_amd' / 'csrc'}", str(root / "scripts" / "scan_check.hip"), "-o", str(exe)],
                   check=True, timeout=300)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [ln for ln in res.stdout.splitlines() if "cells," in ln]
    assert len(lines) == 12 and all("starts ok, tasks ok" in ln for ln in lines), res.stdout
