"""`Crate`: the reference's simulation object (``src/crate/crate.py:19-371``) with the per-timestep
particle update running on an MI355X.

Drop-in surface: ``Crate(world_config)``, ``physics_tick()``, and the
attributes the viewer reads or writes between ticks -- ``particles``, ``particle_velocities``,
``particles_pressure``, ``particle_radius``, ``segments``, ``gravity``, every YAML coefficient by
name, ``editable_coefficients()``, ``debug_arrows``, ``debug_prints``, ``tick``,
``particle_count``, ``diameter``, ``rigid_bodies``, ``particle_sources``.

What runs where
---------------
host   rigid-body motion (rigid_body.py) and pad_segments: O(S) per tick.  The particle sources: on the DEVICE
       (sc_emit_particles, below) with noise="host", where the device holds NumPy's stream; on the host
       (particle_source.py: `_create_new_particles`) in the modes that leave ``np.random`` to the host --
       "counter", "none" and "host-sync".
GPU    everything per particle: removal, wall contacts + hard wall fix, strip sort and neighbor
       lists, pressure / tension / gravity / viscosity / wall bounce / continuous collision,
       integration (sand_crate_amd/csrc/sc_kernels.h).  State stays on the device; the
       ``particles`` / ``particle_velocities`` / ``particles_pressure`` attributes download
       lazily, once per tick, when read.

Collider noise (crate.py:169) and particle sources (particle_source.py:17-24) draw from NumPy's global MT19937
stream in the reference.  Modes:
``"host"``    (default) that very stream, bit for bit, generated ON THE DEVICE: `Crate.__init__` seeds
              ``np.random`` like the reference (crate.py:22) and hands the state to the library
              (sc_rng_set_state); sources and noise are then drawn by kernels (sc_rng.h) and a tick costs no
              readback, no host draw and no upload.  `sync_host_rng()` returns the stream to ``np.random``
              for callers that draw from it themselves between ticks.  (Both branches of NumPy's legacy binomial are
              on the device -- inversion up to flow * dt = 30, BTPE beyond; only dt > 0.5 makes the crate fall back to
              the next mode.)
``"host-sync"`` the same numbers drawn by the host: one device->host count and one upload per tick;
``"counter"`` a counter-based hash on the device keyed by (seed, tick, particle id, slot): same
              distribution, different numbers, no host round trip (used for throughput runs);
``"none"``    no noise (what ``collider_noise_level = 0`` computes).

There is no CPU implementation behind this class: without libsandcrate_hip.so and a GPU it raises.
"""
from __future__ import annotations

import copy
import json
import time

import numpy as np
import yaml

from . import _native as N
from .engine import Engine
from .load_config import WorldConfig
from .particle_source import build_particle_sources
from .rigid_body import build_rigid_bodies
from .utils.geometry_utils import pad_segments

_NOISE_MODES = {"none": N.NOISE_NONE, "host": N.NOISE_HOST, "host-sync": N.NOISE_HOST, "counter": N.NOISE_COUNTER}
FORCE_PHASES = ("tension", "gravity", "pressure", "viscosity", "wall_bounce", "continuous_collision")  # crate.py:110-123
_TICK_COEFFICIENTS = ("dt", "particle_radius", "wall_collision_decay", "pressure_amplifier", "ignored_pressure",
                      "collider_noise_level", "viscosity", "surface_smoothing", "target_pressure")


def tick_geometry(rigid_bodies, particle_radius, cache):
    """-> (segments S x 2 x 2, padded 2S x 2 x 2, [(position, center_velocity, omega, n_segments)]) of one
    tick: `Crate.segments` (crate.py:69-71) and `pad_segments` of it (geometry_utils.py:146-172).  Padding
    is per segment, so bodies that did not move reuse their padded halves from `cache`."""
    if not rigid_bodies:
        return np.zeros((0, 2, 2)), np.zeros((0, 2, 2)), []
    plus, minus = [], []
    for body in rigid_bodies:
        hit = cache.get(id(body))
        # (a body that moves gets a new `segments` array every tick -- rigid_body.py: apply_velocity -- so the identity of
        # the array says whether the cached halves still belong to it; the contents are compared only for the same array,
        # which somebody may have edited in place)
        if hit is None or hit[0] != particle_radius or hit[1] is not body.segments or hit[2].tobytes() != body.segments.tobytes():
            pad = pad_segments(body.segments, particle_radius)
            hit = (particle_radius, body.segments, body.segments.copy(), pad[: len(body)], pad[len(body):])
            cache[id(body)] = hit
        plus.append(hit[3])
        minus.append(hit[4])
    seg = np.concatenate([body.segments for body in rigid_bodies])
    bodies = [(b.position, b.center_velocity, b.angular_clockwise_velocity, len(b)) for b in rigid_bodies]
    return seg, np.concatenate(plus + minus), bodies


class Crate:
    def __init__(self, world_config: WorldConfig, *, device: int = 0, noise: str = "host", noise_seed: int = 0,
                 capacity: int | None = None) -> None:
        if noise not in _NOISE_MODES:
            raise ValueError(f"noise must be one of {sorted(_NOISE_MODES)}")
        np.random.seed(0)  # the reference seeds the global legacy RNG here (crate.py:22)
        self.tick: int = 0
        self.debug_arrows: list = []
        self._debug_prints: str | None = ""
        self.world_config = world_config
        self.rigid_bodies = build_rigid_bodies(world_config.rigid_bodies)
        self.particle_sources = build_particle_sources(world_config.particle_sources)
        for name in self.editable_coefficients():
            setattr(self, name, world_config.coefficients[name])
        self.gravity = np.array(world_config.coefficients["gravity"])

        cap = capacity if capacity is not None else int(self.max_particles) + 1024
        self._engine = Engine(max(int(cap), 1), device=device)
        self._noise = noise
        self._noise_seed = noise_seed
        self._engine.set_noise_mode(_NOISE_MODES[noise], noise_seed)
        if noise == "host":        # the device takes over the global stream the reference draws from
            self._hand_rng_to_device()
        self._count = 0            # particles on the device after the last tick / upload
        self._count_known = True
        self._cache = None         # (particles, velocities, pressure) downloaded for this tick
        self._empty()
        self._tick_seconds = 0.0   # EMA of wall time per tick, for debug_prints
        self._pad_cache = {}
        self._hud_kernels = False
        self._kernel_seconds = {}
        self._hud_forces = False
        self._force_ema = {}
        self._pending_checkpoint = None
        self.last_stats = None

    # ------------------------------------------------------------------ reference accessors
    def editable_coefficients(self) -> list[str]:
        return list(self.world_config.coefficients.keys())

    @property
    def diameter(self) -> float:
        return self.particle_radius * 2

    @property
    def segments(self) -> np.ndarray:
        return np.vstack([body.segments for body in self.rigid_bodies])

    @property
    def particle_count(self) -> int:
        if not self._count_known:
            self._count = self._engine.count()
            self._count_known = True
        return self._count

    # ------------------------------------------------------------------ state attributes
    def _count_or_unknown(self) -> bool:
        return (not self._count_known) or self._count > 0

    def _empty(self) -> None:
        self._cache = (np.zeros((0, 2)), np.zeros((0, 2)), np.zeros((0,)))

    def _state(self):
        if self._cache is None:
            p, v, pr, _ = self._engine.download()
            self._cache = (p, v, pr)
            self._count, self._count_known = len(p), True
        return self._cache

    @property
    def particles(self) -> np.ndarray:
        return self._state()[0]

    @particles.setter
    def particles(self, value) -> None:
        value = np.array(value, dtype=np.float64).reshape(-1, 2)
        vel = self._state()[1]
        if len(vel) != len(value):
            vel = np.zeros_like(value)
        self._set_state(value, vel)

    @property
    def particle_velocities(self) -> np.ndarray:
        return self._state()[1]

    @particle_velocities.setter
    def particle_velocities(self, value) -> None:
        value = np.array(value, dtype=np.float64).reshape(-1, 2)
        pos = self._state()[0]
        if len(pos) != len(value):
            raise ValueError("set particles before particle_velocities when the particle count changes")
        self._set_state(pos, value)

    @property
    def particles_pressure(self) -> np.ndarray:
        return self._state()[2]

    def _set_state(self, particles: np.ndarray, velocities: np.ndarray) -> None:
        if len(particles) > self._engine.capacity:
            self._grow(len(particles))
        self._engine.upload(particles, velocities)
        self._cache = (particles, velocities, np.zeros(len(particles)))
        self._count, self._count_known = len(particles), True

    def _hand_rng_to_device(self) -> None:
        name, key, pos, _, _ = np.random.get_state()
        if name != "MT19937":
            raise RuntimeError("np.random is not the legacy MT19937 generator")
        self._engine.rng_set_state(key, pos)

    def sync_host_rng(self) -> None:
        """noise="host": bring ``np.random`` to where the device stream stands (synchronises).  The device keeps
        drawing from its own copy afterwards: call this before host code draws from the global generator, and
        `_hand_rng_to_device()` happens again on the next tick if the host state moved."""
        if self._noise == "host":
            key, pos = self._engine.rng_get_state()
            np.random.set_state(("MT19937", key, pos, 0, 0.0))
            self._host_rng_mark = (key.copy(), pos)

    def _fall_back_to_host_stream(self) -> None:
        """A particle source whose time step exceeds one half (binomial(n, p) with p > 0.5) is outside what the device
        draws of NumPy's legacy binomial (sc_rng.h: inversion and BTPE for 0 < p <= 0.5): from here on the host draws the
        stream -- same numbers, but two synchronisations per tick.  Said out loud, once."""
        import warnings
        warnings.warn("sand_crate_amd: a particle source draws binomial(n, p) with p > 0.5 -- not on the device; "
                      "physics_tick() falls back to noise='host-sync' (identical results, two host synchronisations per "
                      "tick) for the rest of this run", RuntimeWarning, stacklevel=3)
        self.sync_host_rng()
        self._noise = "host-sync"

    def _grow(self, needed: int) -> None:
        p, v, _ = self._state() if self._count_or_unknown() else (np.zeros((0, 2)), np.zeros((0, 2)), None)
        old = self._engine
        rng = old.rng_get_state() if self._noise == "host" else None
        self._engine = Engine(int(needed * 1.5) + 1024, device=old.device)
        self._engine.set_noise_mode(_NOISE_MODES[self._noise], self._noise_seed)
        if rng is not None:
            self._engine.rng_set_state(*rng)
        old.close()
        if len(p):
            self._engine.upload(p, v)

    # ------------------------------------------------------------------ the tick
    def physics_tick(self) -> None:
        t0 = time.perf_counter()
        self._create_new_particles()
        self.debug_arrows = []
        for body in self.rigid_bodies:  # crate.py:363-365
            body.apply_velocity(self.dt)
        eng = self._engine
        if self._noise == "host":    # the stream lives on the device: nothing comes back, nothing goes up --
            eng.tick(self._pack_tick_inputs())  # ... and the whole tick is one library call
            self._count_known = False
            self.last_stats = None
        elif self._noise == "host-sync":
            self._send_tick_inputs()
            eng.step_begin()
            stats = eng.step_stats()
            self.last_stats = stats
            # crate.py:165-170 draws rand(C_i, 2) particle by particle; one block is the same stream
            eng.set_noise_host(np.random.rand(stats.neighbor_slots, 2))
            eng.step_finish()
            self._count, self._count_known = stats.particles, True
        else:
            eng.tick(self._pack_tick_inputs())
            self._count_known = False
        self._accelerate_free_bodies()
        self._cache = None
        self.tick += 1
        if self._hud_kernels:  # the HUD's phase split (timer.py:37-48), from HIP events; synchronises the tick
            eng.synchronize()
            for name, (ms, launches) in eng.timing().items():
                if launches:
                    self._kernel_seconds[name] = 0.9 * self._kernel_seconds.get(name, 0.0) + 0.1 * ms / 1000.0
            eng.reset_timing()
        if self._hud_forces:  # force_monitor.py:27-33: EMA (0.80) of the mean |dv| of each force phase
            sums, count = eng.force_monitor()
            if count:
                for name, total in zip(FORCE_PHASES, sums):
                    self._force_ema[name] = 0.8 * self._force_ema.get(name, 0.0) + 0.2 * total / count
        dt_wall = time.perf_counter() - t0
        self._tick_seconds = 0.9 * self._tick_seconds + 0.1 * dt_wall
        self._debug_prints = None  # crate.py:129 formats the HUD text every tick; here it is formatted when read

    def show_kernel_times(self, on: bool = True) -> None:
        """The reference's HUD splits the frame into its Python phases (crate.py:93-125 under
        `debug_timer`, reported by timer.py:37-48).  The phases are fused into kernels here; with this on,
        `physics_tick` brackets every launch with HIP events and `debug_prints` shows the same
        `Timing: {name: "x ms (y%)"}` block per kernel.  Costs one synchronisation per tick."""
        self._hud_kernels = bool(on)
        self._kernel_seconds = {}
        self._engine.reset_timing()
        self._engine.enable_timing(self._hud_kernels)

    def show_forces(self, on: bool = True) -> None:
        """The `Forces` block of the reference's HUD (force_monitor.py:35-37, printed at crate.py:135): mean |dv| of
        tension, gravity, pressure, viscosity, wall_bounce and continuous_collision, EMA 0.80, times 1000.  The force
        kernel sums |dv| per phase on the side (results unchanged); costs one synchronisation per tick."""
        self._hud_forces = bool(on)
        self._force_ema = {}
        self._engine.enable_force_monitor(self._hud_forces)

    # ------------------------------------------------------------------ checkpoint (the reference's commented zarr dump,
    # playback.py:109-118, grown into something a run can resume from)
    def begin_checkpoint(self) -> None:
        """Capture the whole simulation state as of now.  Returns at once: the particle state is copied on the device
        and travels to pinned host memory on a side stream while later ticks run; `finish_checkpoint` collects it."""
        if self._pending_checkpoint is not None:
            raise RuntimeError("a checkpoint is already under way")
        self._engine.checkpoint_begin()
        bodies = []
        for body in self.rigid_bodies:
            bodies.append({"segments": body.segments.tolist(), "position": [float(x) for x in body.position],
                           "center_velocity": np.asarray(body.center_velocity, dtype=np.float64).tolist(),
                           "angular_clockwise_velocity": float(body.angular_clockwise_velocity),
                           "time_from_start": float(getattr(body, "time_from_start", 0.0))})
        coefficients = {}
        for name in self.editable_coefficients():
            value = getattr(self, name)
            coefficients[name] = value.tolist() if isinstance(value, np.ndarray) else value
        host_rng = None
        if self._noise != "host":  # the host owns the global stream (the device's copy travels with the particles)
            _, key, pos, has_gauss, cached = np.random.get_state()
            host_rng = {"key": key.tolist(), "pos": int(pos), "has_gauss": int(has_gauss), "cached_gaussian": float(cached)}
        wc = self.world_config
        self._pending_checkpoint = {
            "format": 1, "tick": int(self.tick), "noise": self._noise, "noise_seed": int(self._noise_seed),
            "world_config": {"rigid_bodies": copy.deepcopy(wc.rigid_bodies), "particle_sources": copy.deepcopy(wc.particle_sources),
                             "coefficients": coefficients},
            "bodies": bodies, "host_rng": host_rng, "pressure": self._cache[2] if self._cache is not None else None}

    def finish_checkpoint(self, path) -> None:
        """Wait for the transfer `begin_checkpoint` started (not for later ticks) and write the .npz file."""
        if self._pending_checkpoint is None:
            raise RuntimeError("begin_checkpoint first")
        meta, self._pending_checkpoint = self._pending_checkpoint, None
        snap = self._engine.checkpoint_finish()
        pressure = meta.pop("pressure")
        meta["next_id"] = int(snap["next_id"])
        meta["engine_tick"] = int(snap["tick"])
        arrays = {"particles": snap["particles"], "velocities": snap["velocities"], "ids": snap["ids"]}
        if pressure is not None and len(pressure) == len(snap["ids"]):
            arrays["pressure"] = pressure
        # the stream to restore is the one the run draws from: the device's in noise mode "host", else the host's (a crate
        # that fell back from "host" to "host-sync" still has a -- stale -- device state: it is not written)
        if snap["rng"] is not None and meta["noise"] == "host":
            arrays["rng_key"] = snap["rng"][0]
            meta["rng_pos"] = int(snap["rng"][1])
        np.savez(path, meta=np.array(json.dumps(meta)), **arrays)

    def save_checkpoint(self, path) -> None:
        self.begin_checkpoint()
        self.finish_checkpoint(path)

    @classmethod
    def from_checkpoint(cls, path, *, device: int = 0, capacity: int | None = None) -> "Crate":
        """A crate that continues the run `save_checkpoint` captured: same particles (and ids), velocities, tick,
        wall positions and motor clocks, coefficients as edited, and the random stream where it stood."""
        with np.load(path, allow_pickle=False) as z:
            meta = json.loads(str(z["meta"]))
            arrays = {k: z[k] for k in z.files if k != "meta"}
        wc = meta["world_config"]
        crate = cls(WorldConfig(rigid_bodies=wc["rigid_bodies"], particle_sources=wc["particle_sources"],
                                coefficients=wc["coefficients"]),
                    device=device, noise=meta["noise"], noise_seed=meta["noise_seed"], capacity=capacity)
        for body, saved in zip(crate.rigid_bodies, meta["bodies"]):
            body.segments = np.array(saved["segments"], dtype=np.float64)
            body.position = list(saved["position"])
            body.center_velocity = np.array(saved["center_velocity"], dtype=np.float64)
            body.angular_clockwise_velocity = saved["angular_clockwise_velocity"]
            if hasattr(body, "time_from_start"):
                body.time_from_start = saved["time_from_start"]
        n = len(arrays["ids"])
        if n > crate._engine.capacity:
            crate._grow(n)
        eng = crate._engine
        eng.upload_with_ids(arrays["particles"], arrays["velocities"], arrays["ids"])
        eng.restore_counters(meta["engine_tick"], meta["next_id"])
        crate.tick = meta["tick"]
        crate._count, crate._count_known = n, True
        crate._cache = (arrays["particles"], arrays["velocities"], arrays.get("pressure", np.zeros(n)))
        if meta["noise"] == "host" and "rng_key" in arrays:
            eng.rng_set_state(arrays["rng_key"], meta["rng_pos"])
        elif meta["host_rng"] is not None:
            h = meta["host_rng"]
            np.random.set_state(("MT19937", np.array(h["key"], dtype=np.uint32), h["pos"], h["has_gauss"], h["cached_gaussian"]))
        return crate

    def run(self, n_ticks: int) -> None:
        """`n_ticks` ticks back to back without touching the host state in between (no sources
        may be active, noise must not be "host"): the throughput path bench.py measures."""
        if self._noise in ("host", "host-sync"):
            raise RuntimeError("Crate.run needs noise='counter' or 'none'")
        if any(src.active_ticks > self.tick for src in self.particle_sources):
            raise RuntimeError("Crate.run cannot interleave particle sources; use physics_tick()")
        if n_ticks <= 0:
            return
        eng = self._engine
        # One library call per tick (sc_tick).  Nobody can edit coefficients inside run(), so every tick
        # also promises the next tick's inputs and its removal / wall pass rides on this tick's force
        # kernel (sc_set_next_inputs); each tick's inputs are computed and packed once.
        for body in self.rigid_bodies:
            body.apply_velocity(self.dt)
        now = self._pack_tick_inputs()
        for k in range(n_ticks):
            nxt = None
            self._accelerate_free_bodies()  # this tick's gravity step on free bodies precedes the next tick's motion
            if k + 1 < n_ticks:
                for body in self.rigid_bodies:
                    body.apply_velocity(self.dt)
                nxt = self._pack_tick_inputs()
            eng.tick(now, nxt)
            self.tick += 1
            now = nxt
        self._cache = None
        self._count_known = False

    def _accelerate_free_bodies(self) -> None:
        """crate.py:311-314: gravity accelerates bodies that are neither fixed nor motored.  (Each body owns its
        velocity here; the reference's default `center_velocity` is one array shared by all bodies,
        rigid_body.py:21 -- INTEGRATION.md, deviations.)"""
        for body in self.rigid_bodies:
            if body.moves and not body.driven:
                body.center_velocity = body.center_velocity + self.dt * self.gravity

    def synchronize(self) -> None:
        self._engine.synchronize()

    def _create_new_particles(self) -> None:
        """crate.py:138-147: sources append in order, each seeing the count the previous left."""
        if self._noise == "host":
            active = [s for s in self.particle_sources if s.active_ticks > self.tick]
            if not active:
                return
            mark = getattr(self, "_host_rng_mark", None)
            if mark is not None:  # sync_host_rng() was used: has the host drawn from the stream since?
                _, key, pos, _, _ = np.random.get_state()
                if pos != mark[1] or not np.array_equal(key, mark[0]):
                    self._hand_rng_to_device()
                self._host_rng_mark = None
            try:
                self._engine.emit_particles(active, self.dt, int(self.max_particles))
                self._cache = None
                self._count_known = False
                return
            except N.NativeError as err:
                if err.code == N.ERR_CAPACITY:
                    self._grow(max(self._engine.capacity, int(self.max_particles)) + 1024)
                    self._engine.emit_particles(active, self.dt, int(self.max_particles))
                    self._cache = None
                    self._count_known = False
                    return
                if err.code != N.ERR_DOMAIN:
                    raise
                self._fall_back_to_host_stream()  # binomial(n, p) with p > 0.5: the host draws from here on
        for source in self.particle_sources:
            if source.active_ticks <= self.tick:
                continue
            new_p, new_v = source.generate_particles(dt=self.dt, max_particles=self.max_particles - self.particle_count)
            if new_p is not None:
                if self._count + len(new_p) > self._engine.capacity:
                    self._grow(self._count + len(new_p))
                self._engine.append(new_p, new_v)
                self._count += len(new_p)
                self._cache = None

    def _pack_tick_inputs(self):
        coef = {name: getattr(self, name) for name in _TICK_COEFFICIENTS}
        seg, pad, bodies = tick_geometry(self.rigid_bodies, self.particle_radius, self._pad_cache)
        return self._engine.pack_inputs(coef, self.gravity, seg, pad, bodies)

    def _send_tick_inputs(self) -> None:
        coef = {name: getattr(self, name) for name in _TICK_COEFFICIENTS}
        self._engine.set_params(gravity=self.gravity, **coef)
        bodies = self.rigid_bodies
        if bodies:
            seg = self.segments
            self._engine.set_segments(
                seg, pad_segments(seg, self.particle_radius),
                [(b.position, b.center_velocity, b.angular_clockwise_velocity, len(b)) for b in bodies])
        else:
            self._engine.set_segments(np.zeros((0, 2, 2)), np.zeros((0, 2, 2)), [])

    # ------------------------------------------------------------------ HUD text (crate.py:131-136)
    @property
    def debug_prints(self) -> str:
        """The HUD text the viewer draws every frame (playback.py:81).  Two YAML dumps: formatted on first read after a
        tick rather than in every tick, which is most of a small scene's tick time."""
        if self._debug_prints is None:
            self.set_debug_prints()
        return self._debug_prints

    @debug_prints.setter
    def debug_prints(self, text: str) -> None:
        self._debug_prints = text

    def set_debug_prints(self) -> None:
        count = self._count if self._count_known else "?"
        self.debug_prints = f"Tick: {self.tick}\nParticles: {count}\n"
        frame = self._tick_seconds
        timing = {"tick (host wall, EMA)": f"{1000 * frame:.2f} ms"}
        if self._hud_kernels and frame > 0:
            for name, seconds in self._kernel_seconds.items():
                timing[name] = f"{1000 * seconds:.3f} ms ({100 * seconds / frame:.0f}%)"
        self.debug_prints += yaml.dump({"Timing": timing,
                                        "FPS": f"{int(1 / frame) if frame > 0 else 0} ({1000 * frame:.0f} ms)"})
        if self._hud_forces:  # force_monitor.py:35-37, in the place crate.py:135 gives it
            rounded = {name: float(f"{1000 * value:.1f}") for name, value in self._force_ema.items()}
            self.debug_prints += f"\n\n{yaml.dump({'Forces': rounded})}"
        self.debug_prints += f"\n\n{self.get_coefficient_debug()}"

    def get_coefficient_debug(self) -> str:
        rows = []
        for name in self.editable_coefficients():
            val = getattr(self, name)
            rows.append({name: val.tolist() if isinstance(val, np.ndarray) else val})
        return yaml.dump(rows)

    def kernel_timing(self):
        return self._engine.timing()

    @property
    def engine(self) -> Engine:
        return self._engine
