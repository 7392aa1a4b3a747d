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
