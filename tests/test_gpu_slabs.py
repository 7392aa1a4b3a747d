"""The multi-GPU configurations of BASELINE.json on one GPU: all slabs of the domain in one process
(`SlabChain`: one library context per slab on cuda:0, halo messages moved by device-to-device copies), checked
against the single-domain `Crate` bit for bit -- configs[3] (4,194,304 particles over 4 slabs) and configs[4]
(16,777,216 particles, wave_machine world with its motored wall moving, 8 slabs) at their full sizes -- plus the
message-size agreement and the re-balancing of the cuts.  Everything goes through the C ABI."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import torch
    torch.cuda.init()  # torch's HIP runtime must come up before the library's in a process that uses both
    import sand_crate_amd
    return sand_crate_amd


def bench_world(n):
    import bench
    wc, d = bench.world_for(n)
    p, v = bench.synthetic_state(n)
    return wc, p, v, d


def single_domain(sc, wc, p, v, ticks, noise="counter", seed=1):
    crate = sc.Crate(copy.deepcopy(wc), noise=noise, noise_seed=seed, capacity=len(p) + 1024)
    crate.particles = p
    crate.particle_velocities = v
    crate.run(ticks)
    out = crate.engine.download()
    segments = crate.segments.copy()
    crate.engine.close()
    return out, segments


def assert_chain_equals_single(chain, single):
    gp, gv, gpr, gids = chain.gather_state()
    sp, sv, spr, sids = single
    assert np.array_equal(gids, sids)
    assert np.array_equal(gp, sp)
    assert np.array_equal(gv, sv)
    assert np.array_equal(gpr, spr)


def lean_properties(sc, wc, p, v, d):
    """Size-independent properties of the sort and the neighbor lists, vectorised so that millions of particles
    take seconds: lexsort order, every listed neighbor within one diameter, counts capped at 20 and equal to a
    brute-force disc count on a sample, symmetry of untrimmed lists on a sample of edges."""
    from sand_crate_amd import _native as N
    n = len(p)
    eng = sc.Engine(n)
    eng.set_noise_mode(N.NOISE_NONE, 0)
    eng.upload(p, v)
    crate_like = sc.Crate(copy.deepcopy(wc), noise="none", capacity=16)  # only to build the tick inputs
    crate_like._engine.close()
    crate_like._engine = eng
    for b in crate_like.rigid_bodies:
        b.apply_velocity(crate_like.dt)
    crate_like._send_tick_inputs()
    eng.step_begin()
    rows, sorted_ids = eng.download_sort()
    ids, counts, nbrs, fixed = eng.download_neighbors()
    eng.step_finish()
    eng.close()
    assert len(rows) == n
    fx = np.empty((n, 2))
    fx[ids] = fixed
    ref_rows = np.floor(fx[:, 1] / d).astype(np.int64)
    ref_order = np.lexsort((fx[:, 0], ref_rows))
    assert np.array_equal(sorted_ids, ref_order)
    assert np.array_equal(rows, ref_rows[ref_order])
    valid = np.arange(20)[None, :] < counts[:, None]
    assert counts.max() <= 20 and (nbrs[valid] >= 0).all() and (nbrs[~valid] == -1).all()
    src, dst = np.repeat(ids, counts), nbrs[valid]
    delta = fx[dst] - fx[src]
    assert (src != dst).all() and (np.sqrt(delta[:, 0] ** 2 + delta[:, 1] ** 2) <= d).all()
    cnt = np.empty(n, dtype=np.int64)
    cnt[ids] = counts
    both = (cnt[src] < 20) & (cnt[dst] < 20)
    keys = np.sort(src[both] * n + dst[both])
    rs = np.random.RandomState(0)
    pick = rs.choice(len(keys), min(len(keys), 500000), replace=False)
    rev = (keys[pick] % n) * n + keys[pick] // n
    pos = np.minimum(np.searchsorted(keys, rev), len(keys) - 1)
    assert (keys[pos] == rev).all()
    for i in rs.choice(n, 100, replace=False):
        dd = fx - fx[i]
        assert cnt[i] == min(int((np.sqrt(dd[:, 0] ** 2 + dd[:, 1] ** 2) <= d).sum()) - 1, 20)
    return float(counts.mean())


def test_config3_four_slabs_at_4m_equal_single_gpu(sc):
    """BASELINE.json configs[3]: 4,194,304 particles slab-sharded four ways, ghost-particle halo per step."""
    from sand_crate_amd.slab import SlabChain
    n, ticks = 4194304, 3
    wc, p, v, d = bench_world(n)
    single, segments = single_domain(sc, wc, p, v, ticks)
    chain = SlabChain(copy.deepcopy(wc), p, v, 4, noise="counter", noise_seed=1, axis="y")  # as bench.py cuts it
    assert all(m.overlap for m in chain.members)
    chain.run(ticks)
    chain.synchronize()
    assert sum(chain.owned_counts()) == len(single[3]) == n
    assert min(chain.owned_counts()) > 0.9 * n / 4
    assert_chain_equals_single(chain, single)
    assert np.array_equal(chain.members[0].rigid_bodies[1].segments, segments[4:])  # the motored wall moved alike
    mean_neighbors = lean_properties(sc, wc, p, v, d)
    assert 11.0 < mean_neighbors < 13.5


def test_config4_eight_slabs_at_16m_with_the_motored_wall(sc):
    """BASELINE.json configs[4]: 16,777,216 particles, wave_machine.yaml forcing (its motored wall moves every
    tick), eight slabs -- the full size; the chain and the single domain fit one MI355X together."""
    from sand_crate_amd.slab import SlabChain
    n, ticks = 16777216, 2
    wc, p, v, d = bench_world(n)
    assert any("motored" in body for body in wc.rigid_bodies)
    single, segments = single_domain(sc, wc, p, v, ticks)
    chain = SlabChain(copy.deepcopy(wc), p, v, 8, noise="counter", noise_seed=1, axis="y")  # as bench.py cuts it
    assert all(m.overlap for m in chain.members)
    chain.run(ticks)
    chain.synchronize()
    counts = chain.owned_counts()
    assert sum(counts) == len(single[3]) == n and min(counts) > 0.9 * n / 8
    assert_chain_equals_single(chain, single)
    start = sc.Crate(copy.deepcopy(wc), noise="none", capacity=16).segments
    assert not np.array_equal(segments, start)  # the wall did move
    # ... and the size-independent properties of the sort and the lists at this very size (the chain's memory goes first)
    del chain, single
    import gc
    gc.collect()
    mean_neighbors = lean_properties(sc, wc, p, v, d)
    assert 11.0 < mean_neighbors < 13.5


def test_message_sizes_follow_the_halo_counts(sc):
    """After six ticks of history a message carries the records its direction had six ticks earlier plus headroom,
    not the whole buffer; both ends derive the same size on their own (the chain refuses to move a message
    whose ends disagree), and the results stay those of the single domain."""
    from sand_crate_amd.slab import SlabChain
    n, ticks = 200000, 14
    wc, p, v, d = bench_world(n)
    single, _ = single_domain(sc, wc, p, v, ticks)
    chain = SlabChain(copy.deepcopy(wc), p, v, 3, noise="counter", noise_seed=1)
    cap = chain.members[0].backend.halo_capacity
    chain.run(5)
    chain.run(ticks - 5)  # a second run(): its first tick packs explicitly, the rest ride on the force kernel
    chain.synchronize()
    rec = np.array(chain.message_records)
    assert rec.shape == (ticks, 4)
    assert (rec[:6] == cap).all()                      # no history yet: whole buffers
    assert (rec[6:] < cap).all() and (rec[6:] % 256 == 0).all()
    band = 3 * (1.0 / d) * n * d * d                   # particles in three columns
    assert (rec[6:] > band).all() and (rec[6:] < 2.0 * band + 1280).all()
    assert_chain_equals_single(chain, single)


def test_message_cut_short_is_reported(sc):
    """A message that carries fewer records than its header announces lost ghosts: SC_ERR_CAPACITY at the next
    synchronising call."""
    from sand_crate_amd._native import NativeError
    from sand_crate_amd.slab import SlabChain
    n = 60000
    wc, p, v, d = bench_world(n)
    chain = SlabChain(copy.deepcopy(wc), p, v, 2, noise="counter", noise_seed=1)
    a, b = chain.members
    for m in chain.members:
        m._begin_tick()
        m._pack(whole_messages=True)
    for m in chain.members:
        m._sizes = (64, 64, 64, 64)  # far fewer than the band holds
    chain._move_messages()
    for m in chain.members:
        m._end_tick(False)
    with pytest.raises(NativeError, match="halo buffer was too small"):
        a.synchronize()


def test_rebalanced_cuts_on_the_gpu(sc):
    """The chain starts from badly placed cuts (particles crowd to the left, the cuts are evenly spaced); every two
    ticks the cuts are re-derived from the summed column histograms (sc_column_histogram) and the next halo
    message moves the particles that changed owner.  Results stay those of the single domain."""
    from sand_crate_amd.slab import SlabChain, column_of
    n, ticks = 60000, 9
    wc, p, v, d = bench_world(n)
    p = p.copy()
    p[:, 0] = p[:, 0] ** 1.6
    cols = column_of(p[:, 0], d)
    width = int(cols.max()) + 1
    even = [width // 4, width // 2, 3 * width // 4]
    single, _ = single_domain(sc, wc, p, v, ticks)
    big = 40000  # room for a quarter of a slab changing owner in one message
    chain = SlabChain(copy.deepcopy(wc), p, v, 4, noise="counter", noise_seed=1, rebalance_every=2, cuts=even,
                      halo_capacity=big, capacity=n)
    start_counts = [int(m._own_mask.sum()) for m in chain.members]
    for _ in range(ticks):  # tick by tick: every tick packs explicitly
        chain.run(1)
    chain.synchronize()
    assert chain.members[0].rebalances >= 3
    assert all(m.slabs == chain.slabs for m in chain.members)
    new_cuts = [lo for lo, _ in chain.slabs[1:]]
    assert all(b < a for a, b in zip(even, new_cuts))  # towards the crowd
    counts = chain.owned_counts()
    assert sum(counts) == n and max(counts) < max(start_counts)
    hists = [m.backend.column_histogram(*m._histogram_window()) for m in chain.members]
    assert [int(h.sum()) for h in hists] == counts
    assert_chain_equals_single(chain, single)
    # and with look-ahead between the re-balancing ticks (run(k) promises where it may)
    chain2 = SlabChain(copy.deepcopy(wc), p, v, 4, noise="counter", noise_seed=1, rebalance_every=2, cuts=even,
                       halo_capacity=big, capacity=n)
    chain2.run(ticks)
    chain2.synchronize()
    assert chain2.members[0].rebalances >= 3 and chain2.slabs == chain.slabs
    assert_chain_equals_single(chain2, single)


def test_halo_overlap_splits_the_force_kernel_and_changes_nothing(sc):
    """Halo overlap (configs[4]): with the next tick promised the force kernel runs as two launches -- the blocks that
    may pack halo records, then the interior -- and the messages are moved on the side streams in between.  Same
    particles as without overlap and as the single domain; both launches show up in the kernel timing."""
    from sand_crate_amd.slab import SlabChain
    n, ticks = 300000, 6
    wc, p, v, d = bench_world(n)
    single, _ = single_domain(sc, wc, p, v, ticks)
    for overlap in (True, False):
        chain = SlabChain(copy.deepcopy(wc), p, v, 3, noise="counter", noise_seed=1, overlap=overlap)
        assert all(m.overlap == overlap for m in chain.members)
        eng = chain.members[1].engine
        eng.reset_timing()
        eng.enable_timing(True)
        chain.run(ticks)
        chain.synchronize()
        eng.enable_timing(False)
        launches = eng.timing()["force_integrate"][1]
        assert launches == (2 * (ticks - 1) + 1 if overlap else ticks)  # the last tick of a run() promises nothing
        assert_chain_equals_single(chain, single)


def test_particle_too_fast_for_the_overlapped_message_is_reported(sc):
    """An interior block whose particle ends the tick inside a halo band although it started more than two columns
    away from it: the message had left -- SC_ERR_DOMAIN, not a silently missing ghost."""
    from sand_crate_amd._native import NativeError
    from sand_crate_amd.slab import SlabChain, column_of
    n = 120000
    wc, p, v, d = bench_world(n)
    chain = SlabChain(copy.deepcopy(wc), p, v, 2, noise="counter", noise_seed=1, overlap=True)
    cut = chain.slabs[1][0]
    col = column_of(p[:, 0], d)
    v = v.copy()
    fast = np.flatnonzero((col > cut - 40) & (col < cut - 30))[:50]  # well inside slab 0 ...
    v[fast, 0] = 35 * d / wc.coefficients["dt"]                      # ... and 35 columns to the right in one tick
    chain = SlabChain(copy.deepcopy(wc), p, v, 2, noise="counter", noise_seed=1, overlap=True)
    chain.run(3)
    with pytest.raises(NativeError, match="missed the overlapped halo message"):
        chain.synchronize()


@pytest.mark.parametrize("overlap", [False, True])
def test_slabs_in_a_pile_up_state_equal_the_single_domain(sc, overlap):
    """The pile-up paths (sorted big buckets, grouped cell counts, strided dense tiles, renumbered table ranges) under
    the slab decomposition: ghosts next to piles, halo messages packed by the grouping force kernel."""
    from sand_crate_amd.slab import SlabChain
    from test_gpu_parity import wave_world
    from test_gpu_round2 import pile_up_state
    d, ticks = 0.012, 4
    p, v = pile_up_state(d)
    wc = wave_world(sc, d, 0.1)
    wc.coefficients["max_particles"] = len(p)
    single, _ = single_domain(sc, wc, p, v, ticks)
    chain = SlabChain(copy.deepcopy(wc), p, v, 3, noise="counter", noise_seed=1, overlap=overlap)
    chain.run(ticks)
    chain.synchronize()
    assert sum(chain.owned_counts()) == len(single[3])
    assert_chain_equals_single(chain, single)


@pytest.mark.parametrize("overlap,band_flag", [(False, False), (True, False), (True, True)])
def test_row_slabs_equal_the_single_domain(sc, overlap, band_flag):
    """Slabs of rows (axis="y"): ghosts, migration, halo overlap (as two launches of the force kernel, and as one whose
    band blocks release the side stream through a polled flag) and the look-ahead packing decided by floor(y / d); the
    uniform workload with its motored wall, and the pile-up state."""
    from sand_crate_amd.slab import SlabChain
    from test_gpu_parity import wave_world
    from test_gpu_round2 import pile_up_state
    n, ticks = 300000, 6
    wc, p, v, d = bench_world(n)
    single, _ = single_domain(sc, wc, p, v, ticks)
    chain = SlabChain(copy.deepcopy(wc), p, v, 3, noise="counter", noise_seed=1, overlap=overlap, axis="y", band_flag=band_flag)
    assert all(m.band_flag == band_flag for m in chain.members)  # (one launch of the force kernel + the polled flag)
    chain.run(ticks)
    chain.synchronize()
    assert min(chain.owned_counts()) > 0.8 * n / 3
    assert_chain_equals_single(chain, single)
    rows = np.floor(p[:, 1] / d).astype(np.int64)
    assert chain.slabs[0][1] == chain.slabs[1][0] and (rows < chain.slabs[0][1]).sum() > 0.3 * n  # cut by rows
    del chain
    d2, ticks = 0.012, 4
    p, v = pile_up_state(d2)
    wc = wave_world(sc, d2, 0.1)
    wc.coefficients["max_particles"] = len(p)
    single, _ = single_domain(sc, wc, p, v, ticks)
    # (no overlap here: the equal-count cuts fall right beside the piles' rows, and what a pile of 3000 in one cell
    # throws out crosses more rows per tick than an overlapped message allows for -- that is reported, see
    # test_particle_too_fast_for_the_overlapped_message_is_reported)
    chain = SlabChain(copy.deepcopy(wc), p, v, 3, noise="counter", noise_seed=1, overlap=False, axis="y")
    chain.run(ticks)
    chain.synchronize()
    assert_chain_equals_single(chain, single)


@pytest.mark.parametrize("axis", ["x", "y"])
def test_wave_machine_scene_with_its_source_under_slabs(sc, axis):
    """config/wave_machine.yaml unchanged -- it starts empty and its particle source emits for 500 ticks
    (crate.py:138-147) -- on two slabs, through the source's whole active time and beyond: every slab draws the same new
    particles and keeps the ones it owns; the result is the single-domain `Crate` (same host draws, counter noise) bit
    for bit."""
    from pathlib import Path
    from sand_crate_amd.slab import SlabChain
    ticks = 520
    cfg = Path(__file__).resolve().parent.parent / "config" / "wave_machine.yaml"
    crate = sc.Crate(sc.load_config(cfg).world_config, noise="counter", noise_seed=3)  # seeds np.random (crate.py:22)
    for _ in range(ticks):
        crate.physics_tick()
    single = crate.engine.download()
    crate.engine.close()
    assert len(single[3]) > 2500
    empty = np.zeros((0, 2))
    chain = SlabChain(sc.load_config(cfg).world_config, empty, empty, 2, noise="counter", noise_seed=3, axis=axis)
    chain.run(ticks)
    chain.synchronize()
    counts = chain.owned_counts()
    assert sum(counts) == len(single[3])
    if axis == "x":
        assert min(counts) > 200  # the fluid did spread over both slabs
    assert_chain_equals_single(chain, single)
