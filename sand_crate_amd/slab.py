"""Multi-GPU: the particle update sharded into x-slabs, one process per GPU, ghost particles
exchanged with the two neighbor ranks every tick (the reference has no
multi-device code, so this is new design, checked against the single-domain result).

Decomposition
    column = floor(x / diameter) of a particle's position at the start of the tick.  Rank k owns the
    columns [lo_k, hi_k); cuts are chosen once from the initial column histogram so that every rank
    starts with about the same number of particles.
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
Same results as one GPU
    ids are global, tie-breaks use ids, and the collider noise is the counter-based hash of
    (seed, tick, id, slot), so an owned particle sees the same neighbor list, in the same order,
    with the same noise as in the single-domain run.
"""
from __future__ import annotations

import copy
import math
import os

import numpy as np

from .crate import _NOISE_MODES, _TICK_COEFFICIENTS, tick_geometry
from .rigid_body import build_rigid_bodies

HALO_COLUMNS = 3
HALO_FIELDS = 5


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

    def load(self, particles, velocities, ids) -> None:
        self.engine.upload_with_ids(particles, velocities, ids)

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

    def unpack(self, from_left: bool, from_right: bool) -> None:
        self.engine.halo_unpack(self.recv_left.data_ptr() if from_left else None,
                                self.recv_right.data_ptr() if from_right else None, self.halo_capacity)

    def bundled_rccl(self) -> str | None:
        """torch's own librccl, the fallback path for dlopen when no copy is loaded yet."""
        import os
        cand = os.path.join(os.path.dirname(self.torch.__file__), "lib", "librccl.so")
        return cand if os.path.exists(cand) else None

    def exchange_rccl(self, left: int | None, right: int | None) -> None:
        """One RCCL group on the engine's stream: send/recv with both neighbors (sc_halo_exchange)."""
        self.engine.halo_exchange(self.send_left.data_ptr(), self.recv_left.data_ptr(), -1 if left is None else left,
                                  self.send_right.data_ptr(), self.recv_right.data_ptr(),
                                  -1 if right is None else right, self.halo_capacity)

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
                 capacity: int | None = None, transport: str | None = None):
        import torch.distributed as dist
        if noise == "host":
            raise ValueError("slabs need noise='counter' or 'none' (the host MT19937 stream is one global sequence)")
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.world_config = world_config
        self.rigid_bodies = build_rigid_bodies(world_config.rigid_bodies)
        if world_config.particle_sources:
            raise ValueError("SlabCrate runs the update only; particle sources stay with the single-GPU Crate")
        for name, value in world_config.coefficients.items():
            setattr(self, name, value)
        self.gravity = np.array(world_config.coefficients["gravity"], dtype=np.float64)
        self.tick = 0
        self._pad_cache = {}

        p = np.ascontiguousarray(particles, dtype=np.float64).reshape(-1, 2)
        v = np.ascontiguousarray(velocities, dtype=np.float64).reshape(-1, 2)
        d = self.particle_radius * 2
        cols = column_of(p[:, 0], d)
        self.slabs = partition_columns(cols, self.world)
        self.lo, self.hi = self.slabs[self.rank]
        own = (cols >= self.lo) & (cols < self.hi)
        ids = np.flatnonzero(own).astype(np.int64)
        n_own = int(own.sum())
        rows = max(1.0, 1.0 / d)
        expect_halo = HALO_COLUMNS * rows * (len(p) / max(rows * rows, 1.0))
        if halo_capacity is None:
            halo_capacity = int(3.0 * expect_halo) + 4096
        if capacity is None:
            capacity = int(1.15 * len(p) / self.world) + 4 * halo_capacity + 1024
            capacity = max(capacity, n_own + 4 * halo_capacity + 1024)
        self.left = self.rank - 1 if self.rank > 0 else None
        self.right = self.rank + 1 if self.rank < self.world - 1 else None
        self.backend = backend if backend is not None else HipSlabBackend(capacity, halo_capacity, device, noise, noise_seed)
        self.backend.set_slab(self.lo, self.hi, HALO_COLUMNS, self.left is not None, self.right is not None)
        self._own_mask = own
        self.backend.load(p[own], v[own], ids)
        self._host_staged = dist.is_initialized() and dist.get_backend(group) != "nccl"
        self._stage = {}
        self._ops = None
        self.transport = "torch"
        want = (transport or os.environ.get("SANDCRATE_TRANSPORT", "rccl")).lower()
        if want not in ("rccl", "torch"):
            raise ValueError("transport must be 'rccl' or 'torch'")
        if want == "rccl" and self.world > 1 and not self._host_staged and hasattr(self.backend, "exchange_rccl"):
            self._try_rccl()

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

    def _exchange(self) -> None:
        """One message each way with each existing neighbor."""
        if self.world == 1:
            return
        dist, be = self.dist, self.backend
        if self.transport == "rccl":
            be.exchange_rccl(self.left, self.right)
            return
        pairs = []  # (peer, send tensor, recv tensor)
        if self.left is not None:
            pairs.append((self.left, be.send_left, be.recv_left))
        if self.right is not None:
            pairs.append((self.right, be.send_right, be.recv_right))
        if self._host_staged:
            staged = []
            for peer, send, recv in pairs:
                key = (peer, "r")
                if key not in self._stage:
                    self._stage[key] = send.new_empty(send.shape, device="cpu")
                staged.append((peer, send.to("cpu"), self._stage[key], recv))
            ops = []
            for peer, s_cpu, r_cpu, _ in staged:
                ops.append(dist.P2POp(dist.isend, s_cpu, peer, self.group))
                ops.append(dist.P2POp(dist.irecv, r_cpu, peer, self.group))
            for work in dist.batch_isend_irecv(ops):
                work.wait()
            for _, _, r_cpu, recv in staged:
                recv.copy_(r_cpu)
        else:
            if self._ops is None:  # the same buffers and peers every tick
                self._ops = []
                for peer, send, recv in pairs:
                    self._ops.append(dist.P2POp(dist.isend, send, peer, self.group))
                    self._ops.append(dist.P2POp(dist.irecv, recv, peer, self.group))
            for work in dist.batch_isend_irecv(self._ops):
                work.wait()  # stream-ordered for NCCL: the current stream waits, the host does not

    def run(self, n_ticks: int) -> None:
        be = self.backend
        now = None
        for k in range(n_ticks):
            if now is None:
                for body in self.rigid_bodies:
                    body.apply_velocity(self.dt)
                now = self._tick_inputs()
            be.set_tick_inputs(*now)
            if self.world > 1:
                # a backend that was promised this tick's inputs packed its halo message at the end of the
                # previous tick (HipSlabBackend: in the force kernel's epilogue)
                if not getattr(be, "packed_ahead", False):
                    be.pack()
                self._exchange()
                be.unpack(self.left is not None, self.right is not None)
            nxt = None
            for body in self.rigid_bodies:  # crate.py:311-314: this tick's gravity step on free bodies
                if body.moves and not body.driven:
                    body.center_velocity = body.center_velocity + self.dt * self.gravity
            if k + 1 < n_ticks:  # nobody can edit coefficients inside run(): the next tick's inputs are known
                for body in self.rigid_bodies:
                    body.apply_velocity(self.dt)
                nxt = self._tick_inputs()
            be.step(nxt)
            self.tick += 1
            now = nxt

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
