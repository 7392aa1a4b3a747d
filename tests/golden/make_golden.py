#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference); the fixtures it writes
are data (inputs + expected outputs) and are committed, the reference is not.

    python tests/golden/make_golden.py

The reference imports two packages that are absent here and that it uses only
for annotations (nptyping) and for one 2-D rotation at load time
(pygame.Vector2.rotate, rigid_body.py:38-39).  They are provided as in-memory
modules before the import; nothing of the
reference is edited or copied.
"""
from __future__ import annotations

import math
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def _install_import_stubs() -> None:
    npt = types.ModuleType("nptyping")
    npt.NDArray = np.ndarray
    sys.modules["nptyping"] = npt

    pg = types.ModuleType("pygame")

    class Vector2:
        def __init__(self, x=0.0, y=0.0):
            self.x, self.y = float(x), float(y)

        def rotate(self, angle):
            # pygame's algorithm: fold into [0, 360), exact right angles, else sin/cos of radians
            eps = 1e-6
            a = math.fmod(angle, 360.0)
            if a < 0:
                a += 360.0
            if math.fmod(a + eps, 90.0) < 2 * eps:
                q = int((a + eps) / 90.0) % 4
                x, y = [(self.x, self.y), (-self.y, self.x), (-self.x, -self.y), (self.y, -self.x)][q]
                return Vector2(x, y)
            r = a * math.pi / 180.0
            s, c = math.sin(r), math.cos(r)
            return Vector2(c * self.x - s * self.y, s * self.x + c * self.y)

        def __iter__(self):
            yield self.x
            yield self.y

        def __len__(self):
            return 2

        def __getitem__(self, i):
            return (self.x, self.y)[i]

    pg.Vector2 = Vector2
    sys.modules["pygame"] = pg


_install_import_stubs()
sys.path.insert(0, str(REF))
from src.crate.collision_detector import detect_particle_collisions, strip_sort_particles  # noqa: E402
from src.crate.crate import Crate  # noqa: E402
from src.crate.load_config import load_config  # noqa: E402
from src.crate.utils.geometry_utils import (pad_segments, points_to_segments_distance,  # noqa: E402
                                            segments_crossings)

MAXN = 20
COEF_KEYS = ["dt", "particle_radius", "wall_collision_decay", "pressure_amplifier", "ignored_pressure",
             "collider_noise_level", "viscosity", "surface_smoothing", "target_pressure"]


def pad_lists(lists):
    counts = np.array([len(x) for x in lists], dtype=np.int32)
    table = np.full((len(lists), MAXN), -1, dtype=np.int64)
    for i, x in enumerate(lists):
        table[i, : len(x)] = x
    return counts, table


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {path.stat().st_size / 1024:.1f} KiB")


# ---------------------------------------------------------------- G1/G2 neighbor search
def neighbor_case(name, particles, d):
    particles = np.asarray(particles, dtype=np.float64)
    _, y_floored, order = strip_sort_particles(particles=particles, diameter=d)
    counts, table = pad_lists(detect_particle_collisions(particles=particles, diameter=d))
    save(f"nbr_{name}", particles=particles, diameter=np.float64(d), y_floored=np.asarray(y_floored, np.int64),
         sorted_indices=np.asarray(order, np.int64), counts=counts, table=table)


def neighbor_cases():
    lattice = np.array([[i, j] for i in range(35) for j in range(35)], dtype=np.float64)
    row = np.array([[i, 0] for i in range(35)], dtype=np.float64)
    for d in (0.5, 1, 2):
        neighbor_case(f"lattice35_d{d}", lattice, d)
        neighbor_case(f"row35_d{d}", row, d)
    rs = np.random.RandomState(1234)
    n = 4096
    neighbor_case("uniform4096", rs.rand(n, 2), math.sqrt(12 / (math.pi * n)))
    # dense: ~60 candidates per particle, so the first-20 trim order is pinned
    rs = np.random.RandomState(7)
    neighbor_case("dense600", rs.rand(600, 2) * 0.2, 0.05)
    # duplicates and exact x ties (stable tie-break by original index)
    rs = np.random.RandomState(11)
    pts = np.round(rs.rand(500, 2) * 8) / 8
    neighbor_case("ties500", pts, 0.25)
    # negative coordinates and rows
    rs = np.random.RandomState(5)
    neighbor_case("negative800", rs.rand(800, 2) * 2 - 1, 0.07)
    neighbor_case("single", np.array([[0.3, 0.4]]), 0.01)
    neighbor_case("pair_at_d", np.array([[0.0, 0.0], [0.0, 0.01], [0.01, 0.0], [0.02, 0.0]]), 0.01)


# ---------------------------------------------------------------- geometry KATs
def geometry_cases():
    p = np.array([[i, 0] for i in range(35)], dtype=np.float64)
    seg = np.array([[[i, -1], [i, 1]] for i in range(5)], dtype=np.float64)
    near, dist = points_to_segments_distance(p, seg)
    save("dist_row", particles=p, segments=seg, nearest=near, distances=dist)
    rs = np.random.RandomState(3)
    cfg = load_config(REF / "config" / "wave_machine.yaml")
    crate = Crate(cfg.world_config)
    p = rs.rand(300, 2) * 1.1 - 0.05
    near, dist = points_to_segments_distance(p, crate.segments)
    save("dist_wave", particles=p, segments=crate.segments, nearest=near, distances=dist)
    save("pad_wave", segments=crate.segments, pad=np.float64(0.005), padded=pad_segments(crate.segments, 0.005))
    mv = np.concatenate((p[:, None], p[:, None] + (rs.rand(300, 1, 2) - 0.5) * 0.2), 1)
    save("cross_wave", movements=mv, padded=pad_segments(crate.segments, 0.005),
         crossings=segments_crossings(mv, pad_segments(crate.segments, 0.005)))


# ---------------------------------------------------------------- G3/G5 single ticks
class TickTap:
    """Wraps a reference Crate's phase methods to record the inputs of the tick core
    (state after new particles / removal / body motion) and per-phase velocities."""

    def __init__(self, crate):
        self.c = crate
        self.rec = {}
        for phase in ("calc_virtual_colliders", "populate_colliders", "apply_tension", "apply_pressure",
                      "apply_viscosity", "apply_wall_bounce", "apply_continuous_collision_velocity_fix"):
            self._wrap(phase)

    def _wrap(self, name):
        orig = getattr(self.c, name)

        def wrapped(*a, **k):
            c = self.c
            if name == "calc_virtual_colliders":
                self.rec = {"in_particles": c.particles.copy(), "in_velocities": c.particle_velocities.copy(),
                            "segments": c.segments.copy(),
                            "body_position": np.array([np.asarray(b.position, float) for b in c.rigid_bodies]),
                            "body_velocity": np.array([np.asarray(b.center_velocity, float) for b in c.rigid_bodies]),
                            "body_omega": np.array([float(b.angular_clockwise_velocity) for b in c.rigid_bodies]),
                            "body_nseg": np.array([len(b) for b in c.rigid_bodies], np.int64)}
            if name == "populate_colliders":
                self.rec["fixed_positions"] = c.particles.copy()
                state = np.random.get_state()
            out = orig(*a, **k)
            if name == "populate_colliders":
                total = sum(len(x) for x in c.colliders_indices)
                after = np.random.get_state()
                np.random.set_state(state)
                self.rec["eta_u01"] = np.random.rand(total, 2)
                np.random.set_state(after)
            if name in ("apply_tension", "apply_pressure", "apply_viscosity", "apply_wall_bounce"):
                self.rec["v_after_" + name.split("_", 1)[1].replace("wall_", "")] = c.particle_velocities.copy()
            return out

        setattr(self.c, name, wrapped)

    def finish(self):
        c = self.c
        rec = dict(self.rec)
        rec["out_particles"] = c.particles.copy()
        rec["out_velocities"] = c.particle_velocities.copy()
        rec["out_pressure"] = np.asarray(c.particles_pressure, dtype=np.float64).reshape(-1)
        counts, table = pad_lists(c.colliders_indices)
        rec["neighbor_counts"], rec["neighbor_table"] = counts, table
        P = c.particle_count
        s = np.zeros((P, 2))
        for i in range(P):  # surface normals from what the reference kept (virtual colliders have overlap 0)
            ov = c.collider_overlaps[i]
            if len(ov):
                s[i] = np.sum(((1 - ov) * ov)[:, None] * c.colliders[i], 0)
        rec["surface_normals"] = s
        rec["wall_count"] = np.array([len(v) for v in c.virtual_colliders], np.int64)
        for k in COEF_KEYS:
            rec["coef_" + k] = np.float64(getattr(c, k))
        rec["coef_gravity"] = np.asarray(c.gravity, dtype=np.float64)
        return rec


def synthetic_crate(n, seed, margin, vel_scale, noise_level, warm_ticks, yaml_name="wave_machine.yaml"):
    """SURVEY.md 8d M2 inputs on the wave_machine world: uniform particles, d for ~12 neighbors."""
    cfg = load_config(REF / "config" / yaml_name)
    d = math.sqrt(12 / (math.pi * n))
    co = cfg.world_config.coefficients
    co["particle_radius"] = d / 2
    co["max_particles"] = n
    co["dt"] = 0.002 * (d / 0.01)
    co["collider_noise_level"] = noise_level
    cfg.world_config.particle_sources = []
    crate = Crate(cfg.world_config)
    for _ in range(warm_ticks):  # moves the motored wall; there are no particles yet
        crate.physics_tick()
    rs = np.random.RandomState(seed)
    crate.particles = rs.rand(n, 2) * (1 - 2 * margin) + margin
    crate.particle_velocities = (rs.rand(n, 2) - 0.5) * vel_scale
    return crate


def tick_cases():
    specs = {
        "tick_synth2048_quiet": dict(n=2048, seed=1234, margin=0.02, vel_scale=0.1, noise_level=0.0, warm_ticks=0),
        "tick_synth2048_noise": dict(n=2048, seed=4321, margin=0.02, vel_scale=0.1, noise_level=0.1, warm_ticks=25),
        "tick_synth2048_walls": dict(n=2048, seed=99, margin=0.0, vel_scale=40.0, noise_level=0.1, warm_ticks=60),
        "tick_synth512_cup": dict(n=512, seed=17, margin=0.0, vel_scale=20.0, noise_level=0.1, warm_ticks=40,
                                  yaml_name="stirring_cup.yaml"),
    }
    for name, spec in specs.items():
        np.random.seed(2024)
        crate = synthetic_crate(**spec)
        tap = TickTap(crate)
        crate.physics_tick()
        save(name, **tap.finish())
    # mid-run states of the two scenes (sources active, walls moving, MT noise)
    for yaml_name, ticks in (("stirring_cup.yaml", (100, 300)), ("wave_machine.yaml", (100, 300))):
        cfg = load_config(REF / "config" / yaml_name)
        crate = Crate(cfg.world_config)
        tap = TickTap(crate)
        for t in range(max(ticks) + 1):
            crate.physics_tick()
            if t in ticks:
                save(f"tick_{yaml_name.split('.')[0]}_t{t}", **tap.finish())


# ---------------------------------------------------------------- G4 trajectories
# The scenes are chaotic: a 1e-17 difference in summation order grows ~10x every 6-8 ticks
# (measured: stirring_cup reaches 2e-3 by tick 94, wave_machine 1e-7 by tick 137), so the
# horizons below are where faithful float64 implementations (differing by a few ulp per pair term)
# still agree to well inside 1e-5 relative.
def trajectory_cases():
    for yaml_name, ticks in (("stirring_cup.yaml", (1, 10, 30, 60)), ("wave_machine.yaml", (1, 10, 50, 100))):
        cfg = load_config(REF / "config" / yaml_name)
        crate = Crate(cfg.world_config)
        out = {}
        for t in range(1, max(ticks) + 1):
            crate.physics_tick()
            if t in ticks:
                out[f"particles_t{t}"] = crate.particles.copy()
                out[f"velocities_t{t}"] = crate.particle_velocities.copy()
                out[f"pressure_t{t}"] = np.asarray(crate.particles_pressure, float).reshape(-1)
                out[f"segments_t{t}"] = crate.segments.copy()
        out["ticks"] = np.array(ticks)
        save(f"traj_{yaml_name.split('.')[0]}", **out)


if __name__ == "__main__":
    np.seterr(all="ignore")
    neighbor_cases()
    geometry_cases()
    tick_cases()
    trajectory_cases()
