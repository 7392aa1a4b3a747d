"""Multi-GPU: the particle update sharded into slabs, one process per GPU, ghost particles
exchanged with the two neighbor ranks every tick (the reference has no
multi-device code, so this is new design, checked against the single-domain result).

Decomposition
    column = floor(x / diameter) of a particle's position at the start of the tick (`axis="x"`), or its row
    floor(y / diameter) (`axis="y"`).  Rank k owns the columns / rows [lo_k, hi_k); cuts are chosen once from the
    initial histogram so that every rank starts with about the same number of particles.  Results do not depend on
    the axis.  Columns keep the ranks balanced when the particles settle under gravity; rows make the halo bands
    the first and last blocks of the (row-major) sorted order, so that the halo overlap costs almost nothing
    (with columns a third of the force kernel's blocks hold band particles and its split into two launches costs
    ~25 us per tick at a million particles; bench.py uses rows: its window is the uniform start of the workload).
Ghost band
    3 columns on each side.  Interaction range is one diameter after the hard wall fix, which moves
    a particle by at most 0.1 d per wall contact, and pass B needs pressure and surface normal of
    the neighbors of owned particles, which need THEIR neighbors: 2 x 1.2 d = 2.4 d < 3 columns.
Per tick
    pack (device) -> one fixed-size message to each neighbor (RCCL send/recv over xGMI when the
    process group is NCCL; staged through the host for gloo) -> unpack (device) -> the ordinary tick.
    A record is (x, y, vx, vy, id); record 0 of a message is the count, so the host never needs to
    know how many particles cross.  A particle that has left its slab is in the message too and is
    owned by the receiver from then on (migration rides the halo message).  There is no other
    collective on the data path.
Message sizes
    a message carries the records its direction had six ticks earlier plus headroom, not the whole buffer
    (Engine.halo_sizes: sender and receiver derive the size from the same published count, nothing synchronises).
Re-balancing
    with `rebalance_every=K` the ranks add their column histograms every K ticks (one small all-reduce), derive
    the same new cuts and let the next halo message move the particles that changed owner; a cut moves by at
    most a quarter of the narrower slab next to it per re-balance.
Same results as one GPU
    ids are global, tie-breaks use ids, and the collider noise is the counter-based hash of
    (seed, tick, id, slot), so an owned particle sees the same neighbor list, in the same order,
    with the same noise as in the single-domain run.
"""
from __future__ import annotations

import copy
import math
import time
import os

import numpy as np

from .crate import _NOISE_MODES, _TICK_COEFFICIENTS, tick_geometry
from .particle_source import build_particle_sources
from .rigid_body import build_rigid_bodies

HALO_COLUMNS = 3
HALO_FIELDS = 5


def torch_empty_like_cpu(t):
    return t.new_empty(t.shape, device="cpu")


def column_of(x: np.ndarray, diameter: float) -> np.ndarray:
    return np.floor(np.asarray(x, dtype=np.float64) / diameter).astype(np.int64)


def partition_columns(columns: np.ndarray, n_slabs: int, halo: int = HALO_COLUMNS) -> list[tuple[int, int]]:
    """Cuts [lo_k, hi_k) at column granularity with about equal particle counts; the outer slabs
    are open-ended.  Every slab is at least 2*halo + 2 columns wide so that a particle can be a
    ghost of at most one neighbor on each side."""
    big = 2 ** 40
    if n_slabs == 1:
        return [(-big, big)]
    cmin, cmax = int(columns.min()), int(columns.max())
    min_width = 2 * halo + 2
    if (cmax - cmin + 1) < n_slabs * min_width:
        raise ValueError(f"{cmax - cmin + 1} columns cannot be split into {n_slabs} slabs of >= {min_width} columns")
    hist = np.bincount(columns - cmin, minlength=cmax - cmin + 1)
    cum = np.cumsum(hist)
    cuts = []
    prev = cmin
    for k in range(1, n_slabs):
        target = cum[-1] * k / n_slabs
        c = cmin + int(np.searchsorted(cum, target, side="left")) + 1
        c = max(c, prev + min_width)
        c = min(c, cmax + 1 - (n_slabs - k) * min_width)
        cuts.append(c)
        prev = c
    bounds = [-big] + cuts + [big]
    return [(bounds[k], bounds[k + 1]) for k in range(n_slabs)]


def rebalanced_cuts(hist: np.ndarray, col0: int, slabs: list[tuple[int, int]], budget: int,
                    halo: int = HALO_COLUMNS) -> list[tuple[int, int]]:
    """New cuts from the global column histogram (`hist[k]` = particles in column col0 + k): every cut moves
    towards the equal-count position, but by no more columns than hold `budget` particles (they all travel in
    one halo message) and never past a quarter of the narrower slab next to it; slabs keep their minimum width.
    A pure function of its arguments, so every rank arrives at the same cuts."""
    n = len(slabs)
    if n == 1:
        return list(slabs)
    cum = np.concatenate(([0], np.cumsum(hist)))
    total = int(cum[-1])
    first, last = col0, col0 + len(hist)  # the columns the histogram covers stand in for the open outer edges
    edges = [first] + [lo for lo, _ in slabs[1:]] + [last]
    min_width = 2 * halo + 2
    cuts = []
    for k in range(1, n):
        old = edges[k]
        target = col0 + int(np.searchsorted(cum, total * k / n, side="left"))
        reach = max(1, min(old - edges[k - 1], edges[k + 1] - old) // 4)
        new = min(max(target, old - reach), old + reach)
        step = 1 if new > old else -1
        c, moved = old, 0
        while c != new:  # column by column until the message budget is used up
            col = c if step > 0 else c - 1
            moved += int(hist[min(max(col - col0, 0), len(hist) - 1)])
            if moved > budget:
                break
            c += step
        cuts.append(c)
    for k in range(len(cuts)):  # minimum widths, left to right
        lo = (cuts[k - 1] if k else first) + min_width
        hi = last - (len(cuts) - k) * min_width
        cuts[k] = min(max(cuts[k], lo), hi)
    big = 2 ** 40
    bounds = [-big] + cuts + [big]
    return [(bounds[k], bounds[k + 1]) for k in range(n)]


def draw_new_particles(sources, tick: int, dt: float, max_particles: int, count: int):
    """crate.py:138-147: the active sources in order, each seeing the count the previous one left (the draws come from the
    global NumPy stream: binomial, rand, rand per source -- particle_source.py:17-24).  -> [(positions, velocities), ...]"""
    out = []
    for source in sources:
        if source.active_ticks <= tick:
            continue
        new_p, new_v = source.generate_particles(dt=dt, max_particles=max_particles - count)
        if new_p is not None:
            out.append((new_p, new_v))
            count += len(new_p)
    return out


def draw_with_count_bound(sources, tick: int, dt: float, max_particles: int, bound: int, exact_count):
    """The same draws WITHOUT counting the particles first, whenever the count cannot matter: the reference takes
    min(binomial, max_particles - count) (crate.py:142), so with an upper bound of the count in its place a draw that
    stays below the room it was given is the reference's draw.  Only when some source filled its room does the exact count
    decide: the stream is rewound and the draw repeated with `exact_count()` (one synchronisation and, across ranks, one
    all-reduce -- per emitting tick before this, now only near max_particles).  -> (draws, count bound after them)"""
    state = np.random.get_state()
    count, binding = bound, False
    out = []
    for source in sources:
        if source.active_ticks <= tick:
            continue
        room = max_particles - count
        new_p, new_v = source.generate_particles(dt=dt, max_particles=room)
        n = 0 if new_p is None else len(new_p)
        binding |= n >= room
        if new_p is not None:
            out.append((new_p, new_v))
            count += n
    if not binding:
        return out, count
    np.random.set_state(state)
    exact = int(exact_count())
    out = draw_new_particles(sources, tick, dt, max_particles, exact)
    return out, exact + sum(len(p) for p, _ in out)


class HipSlabBackend:
    """The compute side of one slab on one GPU: an `Engine` in slab mode plus halo buffers held as
    torch tensors (device memory and stream plumbing only)."""

    def __init__(self, capacity: int, halo_capacity: int, device: int, noise: str, noise_seed: int):
        import torch

        from . import _native as N
        from .engine import Engine
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.engine = Engine(capacity, device=device)
        self.engine.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.engine.set_noise_mode(_NOISE_MODES[noise], noise_seed)
        self.halo_capacity = int(halo_capacity)
        shape = ((self.halo_capacity + 1) * HALO_FIELDS,)
        self.send_left = torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.send_right = torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.recv_left = torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.recv_right = torch.zeros(shape, dtype=torch.float64, device=self.device)
        self._N = N
        self._params_key = None
        self._inputs = None
        self.packed_ahead = False
        self._promised = None
        self.overlap = False
        self._side = None

    def load(self, particles, velocities, ids) -> None:
        self.engine.upload_with_ids(particles, velocities, ids)

    def append(self, particles, velocities, ids) -> None:
        self.engine.append_with_ids(particles, velocities, ids)

    def set_axis(self, axis: int) -> None:
        self.engine.set_slab_axis(axis)

    def set_slab(self, lo, hi, halo, has_left, has_right) -> None:
        self.engine.set_slab(lo, hi, halo, has_left, has_right)

    def set_tick_inputs(self, coef, gravity, segments, padded, bodies) -> None:
        if self._promised is not None and self._promised[0] is coef:  # the very inputs the last step() promised
            self._inputs = self._promised[1]
        else:
            self._inputs = self.engine.pack_inputs(coef, gravity, segments, padded, bodies)
        key = (tuple(coef.values()), float(gravity[0]), float(gravity[1]))
        if key != self._params_key:  # the halo kernels need the grid (diameter) before the tick itself runs
            self.engine.set_params(gravity=gravity, **coef)
            self._params_key = key

    def pack(self) -> None:
        self.engine.halo_pack(self.send_left.data_ptr(), self.send_right.data_ptr(), self.halo_capacity)

    def message_sizes(self, whole: bool = False) -> tuple[int, int, int, int]:
        """Records to (send left, receive from the left, send right, receive from the right) in the coming exchange."""
        if whole:
            return (self.halo_capacity,) * 4
        return self.engine.halo_sizes(self.halo_capacity)

    def unpack(self, from_left: bool, from_right: bool, sizes=None) -> None:
        sizes = sizes or (self.halo_capacity,) * 4
        self.engine.halo_unpack(self.recv_left.data_ptr() if from_left else None, sizes[1],
                                self.recv_right.data_ptr() if from_right else None, sizes[3])

    def column_histogram(self, col0: int, n_columns: int) -> np.ndarray:
        return self.engine.column_histogram(col0, n_columns)

    def set_band_flag(self, on: bool) -> None:
        self.engine.set_band_flag(on)

    def set_overlap(self, on: bool) -> None:
        """Halo overlap: the exchange runs on the context's side stream next to the interior blocks of the force
        kernel (sc_set_halo_overlap)."""
        self.engine.set_halo_overlap(on)
        self.overlap = bool(on)
        self._side = self.torch.cuda.ExternalStream(self.engine.side_stream(), device=self.device) if on else None

    def side_stream(self):
        return self._side

    def bundled_rccl(self) -> str | None:
        """torch's own librccl, the fallback path for dlopen when no copy is loaded yet."""
        import os
        cand = os.path.join(os.path.dirname(self.torch.__file__), "lib", "librccl.so")
        return cand if os.path.exists(cand) else None

    def exchange_rccl(self, left: int | None, right: int | None, sizes=None) -> None:
        """One RCCL group on the engine's stream: send/recv with both neighbors (sc_halo_exchange)."""
        self.engine.halo_exchange(self.send_left.data_ptr(), self.recv_left.data_ptr(), -1 if left is None else left,
                                  self.send_right.data_ptr(), self.recv_right.data_ptr(),
                                  -1 if right is None else right, sizes or (self.halo_capacity,) * 4)

    def step(self, next_inputs=None) -> None:
        """The tick, one library call.  With the next tick's inputs promised, the force kernel also runs that
        tick's removal / wall pass and packs its halo message into the send buffers (sc_set_next_inputs)."""
        nxt = self.engine.pack_inputs(*next_inputs) if next_inputs is not None else None
        self.engine.tick(self._inputs, nxt)
        self.packed_ahead = nxt is not None
        self._promised = (next_inputs[0], nxt) if nxt is not None else None

    def synchronize(self) -> None:
        self.engine.synchronize()

    def owned_count(self) -> int:
        return self.engine.owned_count()

    def download_owned(self):
        return self.engine.download()


class SlabCrate:
    """`Crate.run()`-style stepping of one slab per rank.  Every rank constructs it with the FULL
    initial state (so all ranks derive the same cuts) and keeps its own slab."""

    def __init__(self, world_config, particles, velocities, *, device: int = 0, noise: str = "counter",
                 noise_seed: int = 0, group=None, backend=None, halo_capacity: int | None = None,
                 capacity: int | None = None, transport: str | None = None, rebalance_every: int = 0,
                 cuts: list[int] | None = None, overlap: bool | None = None, rank: int | None = None,
                 world: int | None = None, axis: str = "x", band_flag: bool = False):
        """`rank` / `world` given: a member of an in-process `SlabChain` (the chain moves the messages and adds the
        histograms); otherwise they come from torch.distributed."""
        import torch.distributed as dist
        if noise == "host":
            raise ValueError("slabs need noise='counter' or 'none' (the host MT19937 stream is one global sequence)")
        self.dist = dist
        self.group = group
        self._chained = rank is not None
        if self._chained:
            self.rank, self.world = int(rank), int(world)
        else:
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rebalance_every = int(rebalance_every)
        self.rebalances = 0
        self.world_config = world_config
        self.rigid_bodies = build_rigid_bodies(world_config.rigid_bodies)
        # crate.py:138-147 under slabs: every rank draws the SAME new particles from the same host stream (np.random,
        # seeded like crate.py:22) and keeps the ones it owns, under their global ids -- what the single-domain `Crate`
        # does in its counter / none noise modes, so slabs reproduce it bit for bit (see `_emit`)
        self.particle_sources = build_particle_sources(world_config.particle_sources)
        if self.particle_sources and not hasattr(backend if backend is not None else HipSlabBackend, "append"):
            raise ValueError("this backend cannot append particles: particle sources need one that can")
        for name, value in world_config.coefficients.items():
            setattr(self, name, value)
        self.gravity = np.array(world_config.coefficients["gravity"], dtype=np.float64)
        self.tick = 0
        self._pad_cache = {}

        p = np.ascontiguousarray(particles, dtype=np.float64).reshape(-1, 2)
        v = np.ascontiguousarray(velocities, dtype=np.float64).reshape(-1, 2)
        d = self.particle_radius * 2
        if axis not in ("x", "y"):
            raise ValueError("axis must be 'x' (slabs of columns) or 'y' (slabs of rows)")
        self.axis = axis
        cols = column_of(p[:, 1 if axis == "y" else 0], d)
        # (a scene that starts empty -- the YAML scenes do, their sources fill them -- is cut into equal widths of the box)
        self.slabs = partition_columns(cols if len(cols) else np.arange(0, int(math.ceil(1.0 / d)) + 1, dtype=np.int64), self.world)
        if cuts is not None:  # the caller's cut columns instead of the equal-count ones
            if len(cuts) != self.world - 1 or any(b - a < 2 * HALO_COLUMNS + 2 for a, b in zip(cuts[:-1], cuts[1:])):
                raise ValueError("cuts: world - 1 increasing columns, at least 2 * halo + 2 apart")
            bounds = [-2 ** 40] + [int(c) for c in cuts] + [2 ** 40]
            self.slabs = [(bounds[k], bounds[k + 1]) for k in range(self.world)]
        self.lo, self.hi = self.slabs[self.rank]
        own = (cols >= self.lo) & (cols < self.hi)
        ids = np.flatnonzero(own).astype(np.int64)
        n_own = int(own.sum())
        rows = max(1.0, 1.0 / d)
        # what the world may hold: the initial particles, or -- with sources -- whatever they may still add up to
        # max_particles (crate.py:142; `Crate` sizes its context the same way), so that a scene that starts empty
        # does not run out of room or of halo records mid-run
        expected = len(p)
        if self.particle_sources:
            expected = max(expected, int(world_config.coefficients.get("max_particles", 0)))
        expect_halo = HALO_COLUMNS * rows * (expected / max(rows * rows, 1.0))
        if halo_capacity is None:
            halo_capacity = int(3.0 * expect_halo) + 4096
        if capacity is None:
            capacity = int(1.15 * expected / self.world) + 4 * halo_capacity + 1024
            capacity = max(capacity, n_own + 4 * halo_capacity + 1024)
        self.left = self.rank - 1 if self.rank > 0 else None
        self.right = self.rank + 1 if self.rank < self.world - 1 else None
        self.capacity = int(capacity)
        self.backend = backend if backend is not None else HipSlabBackend(capacity, halo_capacity, device, noise, noise_seed)
        if axis == "y" or hasattr(self.backend, "set_axis"):
            self.backend.set_axis(1 if axis == "y" else 0)
        self.backend.set_slab(self.lo, self.hi, HALO_COLUMNS, self.left is not None, self.right is not None)
        self._own_mask = own
        self._next_id = len(p)  # ids are global: every rank numbers the emitted particles alike
        self._count_bound = len(p)  # an upper bound of the global particle count (removals are not counted): `_emit`
        if self.particle_sources and not self._chained:
            np.random.seed(0)   # crate.py:22 (a chain seeds once, for all its members)
        self.backend.load(p[own], v[own], ids)
        self.halo_capacity = int(halo_capacity)
        if overlap is None:  # by default only where a particle may cross many cells per tick and still be packed in time
            overlap = axis == "y"
        self.band_flag = False
        self.overlap = bool(overlap) and self.world > 1 and hasattr(self.backend, "set_overlap")
        if self.overlap:
            self.backend.set_overlap(True)
        self._initial_slabs = list(self.slabs)
        self._now = None      # inputs of the coming tick, when the previous one promised them
        self._sizes = None    # message sizes of the coming exchange
        self._host_staged = not self._chained and dist.is_initialized() and dist.get_backend(group) != "nccl"
        self._stage = {}
        self.transport = "chain" if self._chained else "torch"
        want = (transport or os.environ.get("SANDCRATE_TRANSPORT", "rccl")).lower()
        if want not in ("rccl", "torch"):
            raise ValueError("transport must be 'rccl' or 'torch'")
        if (want == "rccl" and self.world > 1 and not self._chained and not self._host_staged
                and hasattr(self.backend, "exchange_rccl")):
            self._try_rccl()  # with overlap on, its proof exchange also proves RCCL on the side stream
        if self.overlap and not self._chained and self.transport != "rccl" and os.environ.get("SANDCRATE_OVERLAP") != "force":
            # the torch.distributed fallback is the safety net: keep it on the context's own stream
            self.backend.set_overlap(False)
            self.overlap = False
        if band_flag:
            self.set_band_flag(True)

    def set_band_flag(self, on: bool) -> bool:
        """Halo overlap with slabs of rows in ONE launch of the force kernel plus a polling kernel on the side stream
        (sc_set_band_flag) instead of two launches: cheaper when the side stream has a hardware queue of its own,
        disastrous (a 50 ms time-out per tick, then an error) when it shares one with the context's stream.  Only
        between exchanges (right after construction, `reload` or `synchronize` of a finished `run`).  -> in effect"""
        on = bool(on) and self.overlap and self.axis == "y" and hasattr(self.backend, "set_band_flag")
        if hasattr(self.backend, "set_band_flag"):
            self.backend.set_band_flag(on)
        self.band_flag = on
        return on

    def reload(self, particles, velocities) -> None:
        """Start over from a state with the SAME particle positions as the one this object was built with (same
        cuts, same owners): bodies back to their YAML placement, tick 0, the rank's particles uploaded again.
        The communicator and the halo buffers are kept -- bench.py repeats its measurement this way."""
        p = np.ascontiguousarray(particles, dtype=np.float64).reshape(-1, 2)
        v = np.ascontiguousarray(velocities, dtype=np.float64).reshape(-1, 2)
        if len(p) != len(self._own_mask):
            raise ValueError("reload needs the particle count the slabs were cut for")
        self.synchronize()
        self.rigid_bodies = build_rigid_bodies(self.world_config.rigid_bodies)
        self._pad_cache = {}
        self.tick = 0
        self._now = None
        self._next_id = len(p)
        if self.particle_sources and not self._chained:
            np.random.seed(0)
        if self.slabs != self._initial_slabs:
            self._apply_cuts(self._initial_slabs)
        own = self._own_mask
        self.backend.load(p[own], v[own], np.flatnonzero(own).astype(np.int64))
        for name in ("packed_ahead", "_promised", "_inputs", "_params_key"):
            if hasattr(self.backend, name):
                setattr(self.backend, name, False if name == "packed_ahead" else None)

    # ------------------------------------------------------------------ stepping
    @property
    def engine(self):
        return self.backend.engine

    def _tick_inputs(self):
        coef = {name: getattr(self, name) for name in _TICK_COEFFICIENTS}
        seg, pad, bodies = tick_geometry(self.rigid_bodies, self.particle_radius, self._pad_cache)
        return coef, np.array(self.gravity, dtype=np.float64), seg, pad, bodies

    def _try_rccl(self) -> None:
        """Set up the library's own RCCL communicator for the slab chain and prove it with one exchange of a
        known pattern; every step is agreed on by all ranks, so either all of them use it or none does (then
        the torch.distributed P2P path stays)."""
        import torch
        from ._native import NativeError
        dist, be = self.dist, self.backend
        dev = be.device

        def all_ok(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return bool(flag.item())

        path = be.bundled_rccl()
        # every rank must be able to load librccl BEFORE anyone enters the collective communicator set-up: a rank
        # that fails there would return at once and leave the others blocked inside ncclCommInitRank
        if not all_ok(be.engine.comm_available(path)):
            return
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        ok = True
        if self.rank == 0:
            try:
                uid = torch.frombuffer(bytearray(be.engine.comm_unique_id(path)), dtype=torch.uint8).to(dev)
            except (NativeError, RuntimeError, OSError):
                ok = False
        if not all_ok(ok):
            return
        dist.broadcast(uid, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        try:
            be.engine.comm_init(bytes(uid.cpu().numpy().tobytes()), self.rank, self.world, path)
        except (NativeError, RuntimeError, OSError):
            ok = False
        if not all_ok(ok):
            return
        # proof: every rank sends (rank, side) stamps and must receive its neighbors' stamps
        try:
            be.send_left.fill_(float(2 * self.rank))
            be.send_right.fill_(float(2 * self.rank + 1))
            be.recv_left.fill_(-1.0)
            be.recv_right.fill_(-1.0)
            be.exchange_rccl(self.left, self.right)
            # A send / receive pair that never completes would otherwise hold this rank until the launcher's limit with
            # nothing on the screen.  Nothing can be enqueued behind a wedged exchange either, so there is no falling
            # back from here: the wait is bounded and says what to do (SANDCRATE_TRANSPORT=torch skips this path).
            limit = float(os.environ.get("SANDCRATE_RCCL_PROOF_TIMEOUT", "120"))
            stream = be.side_stream() if self.overlap and be.side_stream() is not None else torch.cuda.current_stream(dev)
            import time
            t0 = time.monotonic()
            while not stream.query():
                if time.monotonic() - t0 > limit:
                    raise TimeoutError(f"rank {self.rank}: the RCCL proof exchange with ranks {self.left} / {self.right} did not "
                                       f"complete within {limit:.0f} s; run with SANDCRATE_TRANSPORT=torch to use the "
                                       f"torch.distributed transport instead")
                time.sleep(0.002)
            be.engine.synchronize()
            torch.cuda.synchronize(dev)
            if self.left is not None:
                ok = ok and bool((be.recv_left == float(2 * self.left + 1)).all().item())
            if self.right is not None:
                ok = ok and bool((be.recv_right == float(2 * self.right)).all().item())
        except (NativeError, RuntimeError):
            ok = False
        for t in (be.send_left, be.send_right, be.recv_left, be.recv_right):
            t.zero_()
        torch.cuda.synchronize(dev)
        if all_ok(ok):
            self.transport = "rccl"

    def time_exchanges(self, on: bool = True) -> None:
        """Bracket every halo exchange with two events on the stream it is enqueued on (`exchange_stats`); off by default."""
        self._xchg_events = [] if on and hasattr(self.backend, "torch") else None

    def exchange_stats(self) -> dict:
        """What this rank's halo exchanges moved and took: records per message of the last exchange (as sent: the count
        six ticks earlier + 50 % + 1024, in steps of 256 -- sc_halo_sizes), and, after `time_exchanges()` and a
        `synchronize()`, the mean time from an exchange's enqueueing on its stream to its completion there (with halo
        overlap that includes the side stream's wait for the band blocks of the force kernel)."""
        out = {"rank": self.rank, "transport": self.transport, "overlap": bool(self.overlap), "rebalances": self.rebalances}
        if self._sizes is not None:
            sl, rl, sr, rr = self._sizes
            out.update(records_to_left=int(sl) if self.left is not None else 0, records_from_left=int(rl) if self.left is not None else 0,
                       records_to_right=int(sr) if self.right is not None else 0, records_from_right=int(rr) if self.right is not None else 0)
        ev = getattr(self, "_xchg_events", None)
        if ev:
            out["exchange_us_mean"] = round(1000.0 * sum(a.elapsed_time(b) for a, b in ev) / len(ev), 2)
            out["exchanges_timed"] = len(ev)
        if getattr(self, "host_us_per_tick", None):  # of the last run(): is the rank bound by its host or by its GPU?
            out["host_us_per_tick"] = self.host_us_per_tick
        return out

    def _exchange(self) -> None:
        """One message each way with each existing neighbor, of the sizes agreed for this tick."""
        if self.world == 1 or self._chained:
            return
        ev = getattr(self, "_xchg_events", None)
        if ev is None:
            return self._exchange_now()
        torch = self.backend.torch
        stream = self.backend.side_stream() if self.overlap else torch.cuda.current_stream(self.backend.device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)  # (with halo overlap the side stream then waits for the force kernel's band blocks: that wait is inside)
        self._exchange_now()
        b.record(stream)
        ev.append((a, b))

    def _exchange_now(self) -> None:
        dist, be = self.dist, self.backend
        sl, rl, sr, rr = self._sizes
        if self.transport == "rccl":
            be.exchange_rccl(self.left, self.right, self._sizes)
            return
        words = lambda records: (records + 1) * HALO_FIELDS  # noqa: E731  (+ the header record)
        pairs = []  # (peer, what to send, where to receive)
        if self.left is not None:
            pairs.append((self.left, be.send_left[:words(sl)], be.recv_left[:words(rl)]))
        if self.right is not None:
            pairs.append((self.right, be.send_right[:words(sr)], be.recv_right[:words(rr)]))
        if self._host_staged:
            staged = [(peer, send.to("cpu"), torch_empty_like_cpu(recv), recv) for peer, send, recv in pairs]
            ops = []
            for peer, s_cpu, r_cpu, _ in staged:
                ops.append(dist.P2POp(dist.isend, s_cpu, peer, self.group))
                ops.append(dist.P2POp(dist.irecv, r_cpu, peer, self.group))
            for work in dist.batch_isend_irecv(ops):
                work.wait()
            for _, _, r_cpu, recv in staged:
                recv.copy_(r_cpu)
        else:
            ops = []
            for peer, send, recv in pairs:
                ops.append(dist.P2POp(dist.isend, send, peer, self.group))
                ops.append(dist.P2POp(dist.irecv, recv, peer, self.group))
            if self.overlap:  # on the side stream, next to the interior blocks of the last force kernel
                be.engine.halo_overlap_begin()
                with be.torch.cuda.stream(be.side_stream()):
                    for work in dist.batch_isend_irecv(ops):
                        work.wait()
                be.engine.halo_overlap_end()
            else:
                for work in dist.batch_isend_irecv(ops):
                    work.wait()  # stream-ordered for NCCL: the current stream waits, the host does not

    # -- one tick in phases (SlabChain interleaves them across its members)
    def _rebalance_due(self, tick: int) -> bool:
        return self.rebalance_every > 0 and self.world > 1 and tick > 0 and tick % self.rebalance_every == 0

    def _histogram_window(self) -> tuple[int, int]:
        """Columns of the region particles live in, [-r, 1 + r] (crate.py:152), with the wall-fix margin."""
        d, r = self.particle_radius * 2, self.particle_radius
        col0 = int(math.floor(-r / d)) - 3
        return col0, int(math.floor((1 + r) / d)) + 3 - col0 + 1

    def _apply_cuts(self, slabs) -> None:
        self.slabs = list(slabs)
        self.lo, self.hi = self.slabs[self.rank]
        self.backend.set_slab(self.lo, self.hi, HALO_COLUMNS, self.left is not None, self.right is not None)

    def _begin_tick(self) -> None:
        """The coming tick's inputs go to the backend (the grid the halo kernels use depends on them)."""
        if self._now is None:
            for body in self.rigid_bodies:
                body.apply_velocity(self.dt)
            self._now = self._tick_inputs()
        self.backend.set_tick_inputs(*self._now)

    def _rebalance(self, global_hist=None) -> bool:
        """On schedule: new cuts from the global column histogram (one small all-reduce).  -> cuts changed."""
        if not self._rebalance_due(self.tick):
            return False
        col0, ncols = self._histogram_window()
        if global_hist is None:
            import torch
            hist = torch.from_numpy(self.backend.column_histogram(col0, ncols))
            if not self._host_staged:
                hist = hist.to(self.backend.device)
            self.dist.all_reduce(hist, group=self.group)
            global_hist = hist.cpu().numpy()
        # what may change owner in one go: the message capacity minus what the band itself needs
        budget = max(0, self.halo_capacity // 2)
        new = rebalanced_cuts(np.asarray(global_hist, dtype=np.int64), col0, self.slabs, budget)
        if new == self.slabs:
            return False
        self._apply_cuts(new)
        self.rebalances += 1
        return True

    def _pack(self, whole_messages: bool) -> None:
        if self.world == 1:
            return
        # a backend that was promised this tick's inputs packed its halo message at the end of the previous
        # tick (HipSlabBackend: in the force kernel's epilogue)
        if not getattr(self.backend, "packed_ahead", False):
            self.backend.pack()
        self._sizes = self.backend.message_sizes(whole=whole_messages)

    def _end_tick(self, promise_next: bool) -> None:
        be = self.backend
        if self.world > 1:
            be.unpack(self.left is not None, self.right is not None, self._sizes)
        nxt = None
        for body in self.rigid_bodies:  # crate.py:311-314: this tick's gravity step on free bodies
            if body.moves and not body.driven:
                body.center_velocity = body.center_velocity + self.dt * self.gravity
        if promise_next:  # nobody can edit coefficients inside run(): the next tick's inputs are known
            for body in self.rigid_bodies:
                body.apply_velocity(self.dt)
            nxt = self._tick_inputs()
        be.step(nxt)
        self.tick += 1
        self._now = nxt

    def _may_promise(self, k: int, n_ticks: int) -> bool:
        # a tick that re-balances packs its message itself, after the cuts are known: its predecessor must not; nor must
        # the predecessor of a tick that emits particles (they belong into that tick's halo message)
        return k + 1 < n_ticks and not self._rebalance_due(self.tick + 1) and not self._sources_active(self.tick + 1)

    def _sources_active(self, tick: int) -> bool:
        return any(src.active_ticks > tick for src in self.particle_sources)

    def _emit(self, drawn=None) -> None:
        """create_new_particles (crate.py:138-147), first thing in a tick: per active source `generate_particles` with the
        room left under max_particles -- the GLOBAL count, one all-reduce per emitting tick -- drawn identically on every
        rank; a rank appends the particles whose column (row) it owns.  `drawn`: the particles a `SlabChain` drew for all
        its members."""
        if not self._sources_active(self.tick):
            return
        if drawn is None:
            # (the count only matters near max_particles: `_count_bound`, what was loaded plus everything emitted since)
            drawn, self._count_bound = draw_with_count_bound(self.particle_sources, self.tick, self.dt, int(self.max_particles),
                                                             self._count_bound, self.global_particle_count)
        d = self.particle_radius * 2
        for new_p, new_v in drawn:
            ids = self._next_id + np.arange(len(new_p), dtype=np.int64)
            self._next_id += len(new_p)
            col = column_of(new_p[:, 1 if self.axis == "y" else 0], d)
            own = (col >= self.lo) & (col < self.hi)
            if own.any():
                self.backend.append(new_p[own], new_v[own], ids[own])

    def run(self, n_ticks: int) -> None:
        if self._chained:
            raise RuntimeError("a chain member is stepped by its SlabChain")
        t0 = t_head = time.perf_counter()
        head = min(n_ticks, 4)  # (the library lets the host run four ticks ahead: these are enqueued without waiting)
        for k in range(n_ticks):
            self._begin_tick()  # (first: the count the emission needs is taken on the coming tick's grid)
            self._emit()
            self._pack(whole_messages=self._rebalance())
            self._exchange()
            self._end_tick(self._may_promise(k, n_ticks))
            if k + 1 == head:
                t_head = time.perf_counter()
        # what the host spent per tick: enqueueing alone (the first ticks of the call, nothing to wait for) and over the
        # whole call (the host waits once it is four ticks ahead of the GPU: then this is the GPU's tick)
        self.host_us_per_tick = {"enqueue_first_ticks": round(1e6 * (t_head - t0) / max(head, 1), 1),
                                 "whole_call": round(1e6 * (time.perf_counter() - t0) / max(n_ticks, 1), 1), "ticks": n_ticks}

    def physics_tick(self) -> None:
        self.run(1)

    def synchronize(self) -> None:
        self.backend.synchronize()

    # ------------------------------------------------------------------ results
    def owned_state(self):
        """-> particles, velocities, pressure, ids of the particles this rank owns (id order)."""
        # Particles that moved out of the slab during the last tick still sit here until the next
        # exchange, so every particle is reported by exactly one rank: the one that integrated it.
        return self.backend.download_owned()

    def global_particle_count(self) -> int:
        n = self.backend.owned_count()
        if self.world == 1:
            return n
        import torch
        dev = "cpu" if self._host_staged else self.backend.device
        t = torch.tensor([n], dtype=torch.int64, device=dev)
        self.dist.all_reduce(t, group=self.group)
        return int(t.item())

    def gather_state(self):
        """All ranks -> (particles, velocities, pressure, ids) of the whole domain in id order (a
        verification helper: it moves everything through the host)."""
        mine = self.owned_state()
        if self.world == 1:
            return mine
        parts = [None] * self.world
        self.dist.all_gather_object(parts, mine, group=self.group)
        p = np.concatenate([x[0] for x in parts])
        v = np.concatenate([x[1] for x in parts])
        pr = np.concatenate([x[2] for x in parts])
        ids = np.concatenate([x[3] for x in parts])
        order = np.argsort(ids, kind="stable")
        return p[order], v[order], pr[order], ids[order]


class SlabChain:
    """All slabs of a domain in ONE process on ONE GPU: `n_slabs` chain members (each its own library context),
    halo messages moved by device-to-device copies, histograms added on the host.  The same code path as one
    process per GPU except for the transport -- what the tests use to run the multi-GPU configurations of
    BASELINE.json on a one-GPU box, bit-equal to the single-domain run."""

    def __init__(self, world_config, particles, velocities, n_slabs: int, *, device: int = 0, noise: str = "counter",
                 noise_seed: int = 0, halo_capacity: int | None = None, capacity: int | None = None,
                 rebalance_every: int = 0, cuts: list[int] | None = None, overlap: bool | None = None,
                 backend_factory=None, axis: str = "x", band_flag: bool = False):
        self.members = []
        for k in range(n_slabs):
            backend = backend_factory(k) if backend_factory is not None else None
            self.members.append(SlabCrate(copy.deepcopy(world_config), particles, velocities, device=device, noise=noise,
                                          noise_seed=noise_seed, halo_capacity=halo_capacity, capacity=capacity,
                                          rebalance_every=rebalance_every, cuts=cuts, overlap=overlap, rank=k,
                                          world=n_slabs, backend=backend, axis=axis, band_flag=band_flag))
        self.tick = 0
        if self.members[0].particle_sources:
            np.random.seed(0)  # crate.py:22
        self.message_records = []  # per tick: the records every message carried (left-to-right, then right-to-left)

    @property
    def slabs(self):
        return self.members[0].slabs

    def _move_messages(self) -> None:
        words = lambda records: (records + 1) * HALO_FIELDS  # noqa: E731
        sent = []
        overlap = all(m.overlap for m in self.members)
        for a, b in zip(self.members[:-1], self.members[1:]):
            to_right, from_left = a._sizes[2], b._sizes[1]
            to_left, from_right = b._sizes[0], a._sizes[3]
            if to_right != from_left or to_left != from_right:
                raise RuntimeError(f"slabs {a.rank} and {b.rank} disagree on their message sizes: "
                                   f"{to_right} vs {from_left}, {to_left} vs {from_right}")
            if overlap:  # each copy on the receiver's side stream, behind the band blocks of both ends
                torch = a.backend.torch
                b.engine.halo_overlap_begin(a.engine)
                with torch.cuda.stream(b.backend.side_stream()):
                    b.backend.recv_left[:words(to_right)].copy_(a.backend.send_right[:words(to_right)], non_blocking=True)
                a.engine.halo_overlap_begin(b.engine)
                with torch.cuda.stream(a.backend.side_stream()):
                    a.backend.recv_right[:words(to_left)].copy_(b.backend.send_left[:words(to_left)], non_blocking=True)
            else:
                b.backend.recv_left[:words(to_right)].copy_(a.backend.send_right[:words(to_right)])
                a.backend.recv_right[:words(to_left)].copy_(b.backend.send_left[:words(to_left)])
            sent += [to_right, to_left]
        if overlap:
            for m in self.members:
                m.engine.halo_overlap_end()
        self.message_records.append(sent)

    def run(self, n_ticks: int) -> None:
        ms = self.members
        for k in range(n_ticks):
            for m in ms:
                m._begin_tick()
            if ms[0]._sources_active(ms[0].tick):  # one draw for all members (they share this process's np.random)
                drawn, ms[0]._count_bound = draw_with_count_bound(ms[0].particle_sources, ms[0].tick, ms[0].dt,
                                                                  int(ms[0].max_particles), ms[0]._count_bound,
                                                                  lambda: sum(self.owned_counts()))
                for m in ms:
                    m._emit(drawn)
            changed = False
            if ms[0]._rebalance_due(ms[0].tick):
                col0, ncols = ms[0]._histogram_window()
                hist = sum(m.backend.column_histogram(col0, ncols) for m in ms)
                changed = [m._rebalance(hist) for m in ms][0]
            for m in ms:
                m._pack(whole_messages=changed)
            self._move_messages()
            promise = ms[0]._may_promise(k, n_ticks)
            for m in ms:
                m._end_tick(promise)
            self.tick += 1

    def synchronize(self) -> None:
        for m in self.members:
            m.synchronize()

    def owned_counts(self) -> list[int]:
        return [m.backend.owned_count() for m in self.members]

    def gather_state(self):
        parts = [m.owned_state() for m in self.members]
        ids = np.concatenate([x[3] for x in parts])
        order = np.argsort(ids, kind="stable")
        return tuple(np.concatenate([x[k] for x in parts])[order] for k in range(4))
