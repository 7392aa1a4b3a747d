"""`Engine`: one GPU context of libsandcrate_hip.so with NumPy in/out.

This is the thinnest Python layer over the C ABI (include/sandcrate_hip.h); `Crate` (crate.py)
builds the reference's `Crate.physics_tick()` surface on top of it.  Device state stays resident
between ticks; nothing is copied back unless asked for.
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass

import numpy as np

from . import _native as N


@dataclass
class StepStats:
    particles: int
    neighbor_slots: int
    max_neighbors: int
    wall_particles: int
    flags: int


_COEF_ORDER = ("dt", "particle_radius", "wall_collision_decay", "pressure_amplifier", "ignored_pressure",
               "collider_noise_level", "viscosity", "surface_smoothing", "target_pressure")


_BODY_FMT, _BODY_SIZE = "<5dii", C.sizeof(N.Body)
_PARAMS_FMT = f"<{len(_COEF_ORDER) + 2}d"
assert struct.calcsize(_BODY_FMT) == _BODY_SIZE and struct.calcsize(_PARAMS_FMT) == C.sizeof(N.Params)
assert N.TickInputs.params.offset == 0


class PackedInputs:
    """sc_tick_inputs plus the NumPy buffers it points into (kept alive with it)."""

    __slots__ = ("struct", "ref", "_seg", "_pad", "_bodies")

    def __init__(self, coef, gravity, segments, padded, bodies):
        seg = N.f64(segments).reshape(-1, 2, 2)
        pad = N.f64(padded).reshape(-1, 2, 2)
        if len(pad) != 2 * len(seg):
            raise ValueError("padded must hold two segments per wall segment")
        bodies = list(bodies)
        arr = (N.Body * max(len(bodies), 1))()
        # (packed straight into the C structs: this runs every tick, and a ctypes constructor per body and field costs
        # the host more than the GPU spends on a small scene's kernel)
        for k, (pos, vel, omega, nseg) in enumerate(bodies):
            struct.pack_into(_BODY_FMT, arr, _BODY_SIZE * k, pos[0], pos[1], vel[0], vel[1], omega, nseg, 0)
        t = N.TickInputs()
        struct.pack_into(_PARAMS_FMT, t, 0, *[coef[k] for k in _COEF_ORDER], gravity[0], gravity[1])
        t.segments = N.dptr(seg)
        t.padded = N.dptr(pad)
        t.bodies = arr
        t.n_segments = len(seg)
        t.n_bodies = len(bodies)
        self.struct, self.ref, self._seg, self._pad, self._bodies = t, C.byref(t), seg, pad, arr


class Engine:
    def __init__(self, capacity: int, device: int = 0):
        self._lib = N.load()
        self._ctx = N._P()
        N.check(self._lib.sc_create(int(device), int(capacity), C.byref(self._ctx)))
        self.capacity = int(capacity)
        self.device = int(device)

    # -- lifetime
    def close(self) -> None:
        if self._ctx:
            self._lib.sc_destroy(self._ctx)
            self._ctx = N._P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle: int) -> None:
        """Enqueue on this hipStream_t (0 = HIP's default stream), e.g. torch's current stream."""
        N.check(self._lib.sc_set_stream(self._ctx, N._P(int(stream_handle))))

    def use_own_stream(self) -> None:
        N.check(self._lib.sc_use_own_stream(self._ctx))

    # -- state
    def upload(self, particles, velocities) -> None:
        p, v = N.f64(particles).reshape(-1, 2), N.f64(velocities).reshape(-1, 2)
        if p.shape != v.shape:
            raise ValueError("particles and velocities must both be P x 2")
        N.check(self._lib.sc_upload_state(self._ctx, N.dptr(p), N.dptr(v), len(p)))

    def upload_with_ids(self, particles, velocities, ids) -> None:
        """Slab mode: the particles this GPU owns, carrying their global ids."""
        p, v = N.f64(particles).reshape(-1, 2), N.f64(velocities).reshape(-1, 2)
        i = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        if not (len(p) == len(v) == len(i)):
            raise ValueError("particles, velocities and ids must have one row per particle")
        N.check(self._lib.sc_upload_state_ids(self._ctx, N.dptr(p), N.dptr(v), N.i64ptr(i), len(p)))

    def append_with_ids(self, particles, velocities, ids) -> None:
        p, v = N.f64(particles).reshape(-1, 2), N.f64(velocities).reshape(-1, 2)
        i = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        if not (len(p) == len(v) == len(i)):
            raise ValueError("particles, velocities and ids must have one row per particle")
        N.check(self._lib.sc_append_particles_ids(self._ctx, N.dptr(p), N.dptr(v), N.i64ptr(i), len(p)))

    def append(self, particles, velocities) -> None:
        p, v = N.f64(particles).reshape(-1, 2), N.f64(velocities).reshape(-1, 2)
        N.check(self._lib.sc_append_particles(self._ctx, N.dptr(p), N.dptr(v), len(p)))

    def count(self) -> int:
        n = C.c_int64(0)
        N.check(self._lib.sc_count(self._ctx, C.byref(n)))
        return n.value

    def download(self, room: int | None = None):
        """-> particles (P,2), velocities (P,2), pressure (P,), ids (P,) in particle-index order."""
        room = self.capacity if room is None else int(room)
        xy = np.empty((room, 2))
        vxy = np.empty((room, 2))
        pr = np.empty(room)
        ids = np.empty(room, dtype=np.int64)
        n = C.c_int64(0)
        N.check(self._lib.sc_download_state(self._ctx, N.dptr(xy), N.dptr(vxy), N.dptr(pr), N.i64ptr(ids), room, C.byref(n)))
        k = n.value
        return xy[:k].copy(), vxy[:k].copy(), pr[:k].copy(), ids[:k].copy()

    # -- per-tick inputs
    def set_params(self, *, dt, particle_radius, wall_collision_decay, pressure_amplifier, ignored_pressure,
                   collider_noise_level, viscosity, surface_smoothing, target_pressure, gravity) -> None:
        g = np.asarray(gravity, dtype=np.float64).reshape(2)
        p = N.Params(float(dt), float(particle_radius), float(wall_collision_decay), float(pressure_amplifier),
                     float(ignored_pressure), float(collider_noise_level), float(viscosity), float(surface_smoothing),
                     float(target_pressure), float(g[0]), float(g[1]))
        N.check(self._lib.sc_set_params(self._ctx, C.byref(p)))

    def set_segments(self, segments, padded, bodies) -> None:
        """segments (S,2,2); padded (2S,2,2); bodies: iterable of (position, center_velocity, omega, n_segments)."""
        seg = N.f64(segments).reshape(-1, 2, 2)
        pad = N.f64(padded).reshape(-1, 2, 2)
        if len(pad) != 2 * len(seg):
            raise ValueError("padded must hold two segments per wall segment")
        bodies = list(bodies)
        arr = (N.Body * max(len(bodies), 1))()
        for k, (pos, vel, omega, nseg) in enumerate(bodies):
            pos = np.asarray(pos, dtype=np.float64).reshape(2)
            vel = np.asarray(vel, dtype=np.float64).reshape(2)
            arr[k] = N.Body(pos[0], pos[1], vel[0], vel[1], float(omega), int(nseg), 0)
        N.check(self._lib.sc_set_segments(self._ctx, N.dptr(seg), N.dptr(pad), len(seg), arr, len(bodies)))

    def set_next_inputs(self, *, gravity, segments, bodies, **coef) -> None:
        """Promise the inputs of the next tick (between step_begin and step_finish); see the header."""
        g = np.asarray(gravity, dtype=np.float64).reshape(2)
        p = N.Params(*(float(coef[k]) for k in ("dt", "particle_radius", "wall_collision_decay", "pressure_amplifier",
                                               "ignored_pressure", "collider_noise_level", "viscosity",
                                               "surface_smoothing", "target_pressure")), float(g[0]), float(g[1]))
        seg = N.f64(segments).reshape(-1, 2, 2)
        bodies = list(bodies)
        arr = (N.Body * max(len(bodies), 1))()
        for k, (pos, vel, omega, nseg) in enumerate(bodies):
            pos = np.asarray(pos, dtype=np.float64).reshape(2)
            vel = np.asarray(vel, dtype=np.float64).reshape(2)
            arr[k] = N.Body(pos[0], pos[1], vel[0], vel[1], float(omega), int(nseg), 0)
        N.check(self._lib.sc_set_next_inputs(self._ctx, C.byref(p), N.dptr(seg), len(seg), arr, len(bodies)))

    def pack_inputs(self, coef, gravity, segments, padded, bodies) -> "PackedInputs":
        """The inputs of one tick as one C struct (sc_tick_inputs): `coef` maps the nine per-tick coefficient
        names to values, `bodies` is an iterable of (position, center_velocity, omega, n_segments)."""
        return PackedInputs(coef, gravity, segments, padded, bodies)

    def tick(self, now: "PackedInputs", nxt: "PackedInputs | None" = None) -> None:
        """One whole tick in one library call (sc_tick); `nxt` promises the next tick's inputs."""
        N.check(self._lib.sc_tick(self._ctx, now.ref, nxt.ref if nxt is not None else None))

    def set_noise_mode(self, mode: int, seed: int = 0) -> None:
        N.check(self._lib.sc_set_noise_mode(self._ctx, int(mode), int(seed) & (2 ** 64 - 1)))

    # -- the tick
    def step_begin(self) -> None:
        N.check(self._lib.sc_step_begin(self._ctx))

    def step_stats(self) -> StepStats:
        s = N.Stats()
        N.check(self._lib.sc_step_stats(self._ctx, C.byref(s)))
        return StepStats(s.particles, s.neighbor_slots, s.max_neighbors, s.wall_particles, s.flags)

    def set_noise_host(self, u01) -> None:
        u = N.f64(u01).reshape(-1, 2)
        N.check(self._lib.sc_set_noise_host(self._ctx, N.dptr(u), len(u)))

    def step_finish(self) -> None:
        N.check(self._lib.sc_step_finish(self._ctx))

    def step(self, n_ticks: int = 1) -> None:
        N.check(self._lib.sc_step(self._ctx, int(n_ticks)))

    def synchronize(self) -> None:
        N.check(self._lib.sc_synchronize(self._ctx))

    def set_scan_patience(self, polls: int) -> None:
        """How often a workgroup of the bucket scan asks for a predecessor's total before the tick is abandoned
        (sc_set_scan_patience; negative: at once -- the path's test)."""
        N.check(self._lib.sc_set_scan_patience(self._ctx, int(polls)))

    # -- parity taps (between step_begin and step_finish)
    def download_sort(self):
        room = self.capacity
        rows = np.empty(room, dtype=np.int64)
        ids = np.empty(room, dtype=np.int64)
        n = C.c_int64(0)
        N.check(self._lib.sc_download_sort(self._ctx, N.i64ptr(rows), N.i64ptr(ids), room, C.byref(n)))
        return rows[:n.value].copy(), ids[:n.value].copy()

    def download_neighbors(self):
        """-> ids (P,), counts (P,), neighbor ids (P,20), fixed positions (P,2); one row per sorted slot."""
        room = self.capacity
        ids = np.empty(room, dtype=np.int64)
        cnt = np.empty(room, dtype=np.int32)
        nb = np.empty((room, N.MAX_NEIGHBORS), dtype=np.int64)
        fx = np.empty((room, 2))
        n = C.c_int64(0)
        N.check(self._lib.sc_download_neighbors(self._ctx, N.i64ptr(ids), N.i32ptr(cnt), N.i64ptr(nb), N.dptr(fx), room,
                                                C.byref(n)))
        k = n.value
        return ids[:k].copy(), cnt[:k].copy(), nb[:k].copy(), fx[:k].copy()

    def download_normals(self):
        room = self.capacity
        s = np.empty((room, 2))
        n = C.c_int64(0)
        N.check(self._lib.sc_download_normals(self._ctx, N.dptr(s), room, C.byref(n)))
        return s[:n.value].copy()

    # -- multi-GPU slabs (device pointers are plain integers, e.g. torch_tensor.data_ptr())
    def set_slab(self, col_lo: int, col_hi: int, halo: int, has_left: bool, has_right: bool) -> None:
        N.check(self._lib.sc_set_slab(self._ctx, int(col_lo), int(col_hi), int(halo), int(has_left), int(has_right)))

    def set_band_flag(self, on: bool) -> None:
        """Halo overlap with slabs of rows: one launch of the force kernel + a polling kernel on the side stream
        (sc_set_band_flag) instead of two launches."""
        N.check(self._lib.sc_set_band_flag(self._ctx, int(bool(on))))

    def set_slab_axis(self, axis: int) -> None:
        """0: slabs are ranges of columns floor(x / d) (the default); 1: of rows floor(y / d)."""
        N.check(self._lib.sc_set_slab_axis(self._ctx, int(axis)))

    def halo_pack(self, dev_left: int, dev_right: int, capacity_records: int) -> None:
        N.check(self._lib.sc_halo_pack(self._ctx, N._P(dev_left), N._P(dev_right), int(capacity_records)))

    def halo_sizes(self, capacity_records: int) -> tuple[int, int, int, int]:
        """-> records to (send left, receive from the left, send right, receive from the right) in the coming exchange."""
        a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        N.check(self._lib.sc_halo_sizes(self._ctx, int(capacity_records), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def halo_unpack(self, dev_from_left: int | None, left_records: int, dev_from_right: int | None,
                    right_records: int) -> None:
        N.check(self._lib.sc_halo_unpack(self._ctx, N._P(dev_from_left) if dev_from_left else None, int(left_records),
                                         N._P(dev_from_right) if dev_from_right else None, int(right_records)))

    def column_histogram(self, col0: int, n_columns: int) -> np.ndarray:
        hist = np.zeros(int(n_columns), dtype=np.int64)
        N.check(self._lib.sc_column_histogram(self._ctx, int(col0), int(n_columns), N.i64ptr(hist)))
        return hist

    # -- RCCL transport of the halo exchange (optional; see the header)
    @staticmethod
    def comm_available(rccl_path: str | None = None) -> bool:
        """True when librccl can be loaded here (sc_comm_available); never raises."""
        return N.load().sc_comm_available(rccl_path.encode() if rccl_path else None) == 0

    @staticmethod
    def comm_unique_id(rccl_path: str | None = None) -> bytes:
        buf = C.create_string_buffer(128)
        N.check(N.load().sc_comm_unique_id(rccl_path.encode() if rccl_path else None, C.cast(buf, N._P)))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int, rccl_path: str | None = None) -> None:
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        buf = C.create_string_buffer(unique_id, 128)
        N.check(self._lib.sc_comm_init(self._ctx, rccl_path.encode() if rccl_path else None, C.cast(buf, N._P),
                                       int(rank), int(world)))

    def comm_destroy(self) -> None:
        N.check(self._lib.sc_comm_destroy(self._ctx))

    def halo_exchange(self, send_left: int | None, recv_left: int | None, left_rank: int, send_right: int | None,
                      recv_right: int | None, right_rank: int, sizes: tuple[int, int, int, int]) -> None:
        """Device pointers as ints; a negative rank = no neighbor on that side; `sizes` as halo_sizes() gives them.
        Enqueued on the context's stream."""
        ptr = lambda a: N._P(a) if a else None  # noqa: E731
        sl, rl, sr, rr = (int(k) for k in sizes)
        N.check(self._lib.sc_halo_exchange(self._ctx, ptr(send_left), sl, ptr(recv_left), rl, int(left_rank),
                                           ptr(send_right), sr, ptr(recv_right), rr, int(right_rank)))

    # -- halo overlap: the exchange on the side stream, next to the interior blocks of the force kernel
    def set_halo_overlap(self, on: bool = True) -> None:
        N.check(self._lib.sc_set_halo_overlap(self._ctx, 1 if on else 0))

    def side_stream(self) -> int:
        s = N._P()
        N.check(self._lib.sc_side_stream(self._ctx, C.byref(s)))
        return int(s.value or 0)

    def halo_overlap_begin(self, peer: "Engine | None" = None) -> None:
        N.check(self._lib.sc_halo_overlap_begin(self._ctx, peer._ctx if peer is not None else None))

    def halo_overlap_end(self) -> None:
        N.check(self._lib.sc_halo_overlap_end(self._ctx))

    def owned_count(self) -> int:
        n = C.c_int64(0)
        N.check(self._lib.sc_owned_count(self._ctx, C.byref(n)))
        return n.value

    # -- force monitor
    def enable_force_monitor(self, on: bool = True) -> None:
        N.check(self._lib.sc_enable_force_monitor(self._ctx, 1 if on else 0))

    def force_monitor(self):
        """-> (sum of |dv| per phase (6,), particles summed) since the last call; synchronises."""
        sums = np.zeros(6)
        n = C.c_int64(0)
        N.check(self._lib.sc_get_force_monitor(self._ctx, N.dptr(sums), C.byref(n)))
        return sums, n.value

    # -- checkpoint
    def checkpoint_begin(self) -> None:
        N.check(self._lib.sc_checkpoint_begin(self._ctx))

    def checkpoint_finish(self, room: int | None = None):
        """-> dict(particles, velocities, ids, tick, next_id, rng) of the state sc_checkpoint_begin captured."""
        room = self.capacity if room is None else int(room)
        xy, vxy = np.empty((room, 2)), np.empty((room, 2))
        ids = np.empty(room, dtype=np.int64)
        n, tick, nid, pos = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int32(-1)
        key = np.zeros(624, dtype=np.uint32)
        N.check(self._lib.sc_checkpoint_finish(self._ctx, N.dptr(xy), N.dptr(vxy), N.i64ptr(ids), room, C.byref(n),
                                               C.byref(tick), C.byref(nid), key.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               C.byref(pos)))
        k = n.value
        return dict(particles=xy[:k].copy(), velocities=vxy[:k].copy(), ids=ids[:k].copy(), tick=tick.value,
                    next_id=nid.value, rng=(key, pos.value) if pos.value >= 0 else None)

    def restore_counters(self, tick: int, next_id: int) -> None:
        N.check(self._lib.sc_restore_counters(self._ctx, int(tick), int(next_id)))

    # -- NumPy's global MT19937 stream on the device
    def rng_set_state(self, key, pos: int) -> None:
        """Hand the stream of `np.random.get_state()` (624-word key, position) to the device."""
        k = np.ascontiguousarray(key, dtype=np.uint32).reshape(624)
        N.check(self._lib.sc_rng_set_state(self._ctx, k.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos)))

    def rng_get_state(self):
        """-> (key, position) of the device stream; synchronises."""
        k = np.zeros(624, dtype=np.uint32)
        pos = C.c_int32(0)
        N.check(self._lib.sc_rng_get_state(self._ctx, k.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos)))
        return k, pos.value

    def emit_particles(self, sources, dt: float, max_particles: int) -> None:
        """particle_source.py:17-24 for `sources` (objects with radius, position, velocity, flow, noise) on the device."""
        arr = (N.Source * max(len(sources), 1))()
        for k, s in enumerate(sources):
            arr[k] = N.Source(float(s.radius), float(s.position[0]), float(s.position[1]), float(s.velocity[0]),
                              float(s.velocity[1]), float(s.noise), int(s.flow))
        N.check(self._lib.sc_emit_particles(self._ctx, arr, len(sources), float(dt), int(max_particles)))

    # -- timing
    def enable_timing(self, on: bool = True) -> None:
        N.check(self._lib.sc_enable_timing(self._ctx, 1 if on else 0))

    def reset_timing(self) -> None:
        N.check(self._lib.sc_reset_timing(self._ctx))

    def timing(self) -> dict[str, tuple[float, int]]:
        """-> {kernel name: (total ms, launches)} since reset_timing()."""
        ms = np.zeros(N.NUM_KERNELS)
        cnt = np.zeros(N.NUM_KERNELS, dtype=np.int64)
        N.check(self._lib.sc_get_timing(self._ctx, N.dptr(ms), N.i64ptr(cnt)))
        return {self._lib.sc_kernel_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(N.NUM_KERNELS)}
