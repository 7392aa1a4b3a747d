"""YAML-driven scene runner on top of ``oracle.tick``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``OracleCrate`` follows the
order of ``Crate.physics_tick`` (crate.py:91-129): new particles, removal, body
motion, then ``tick_core``.  It draws from the global legacy NumPy RNG in the
reference's order (crate.py:22 seeds it; particle_source.py:18-23 and
crate.py:169 consume it), so whole trajectories can be compared.
"""
from __future__ import annotations

import numpy as np

from .tick import BodyState, remove_outside, tick_core
from .world import World, build_bodies, build_sources


class OracleCrate:
    def __init__(self, world: World, tick_fn=tick_core):
        np.random.seed(0)                                        # crate.py:22
        self.tick = 0
        self.particles = np.zeros((0, 2))
        self.particle_velocities = np.zeros((0, 2))
        self.particles_pressure = np.zeros((0,))
        self.rigid_bodies = build_bodies(world.rigid_bodies)
        self.particle_sources = build_sources(world.particle_sources)
        self.coef = dict(world.coefficients)                     # crate.py:55-57
        self.coef["gravity"] = np.array(self.coef["gravity"], dtype=np.float64)
        self.last = None
        self._tick_fn = tick_fn

    @property
    def particle_count(self) -> int:
        return self.particles.shape[0]

    @property
    def segments(self) -> np.ndarray:
        return np.vstack([b.segments for b in self.rigid_bodies])  # crate.py:69-71

    def body_states(self):
        return [BodyState(np.asarray(b.position, dtype=np.float64), np.asarray(b.center_velocity, dtype=np.float64),
                          float(b.angular_clockwise_velocity), len(b)) for b in self.rigid_bodies]

    def physics_tick(self) -> None:
        c = self.coef
        for src in self.particle_sources:                        # crate.py:138-147
            if src.active_ticks <= self.tick:
                continue
            pos, vel = src.emit(c["dt"], c["max_particles"] - self.particle_count)
            if pos is not None:
                self.particles = np.vstack((self.particles, pos))
                self.particle_velocities = np.vstack((self.particle_velocities, vel))
        self.particles, self.particle_velocities = remove_outside(
            self.particles, self.particle_velocities, c["particle_radius"])
        for b in self.rigid_bodies:                              # crate.py:363-365
            b.advance(c["dt"])

        # crate.py:169 draws rand(C_i, 2) per particle in index order, after the neighbor search;
        # one rand(sum C_i, 2) is the same stream.
        out = self._tick_fn(self.particles, self.particle_velocities, self.segments, self.body_states(), c,
                            eta_u01=lambda total: np.random.rand(total, 2))
        self.particles = out["particles"]
        self.particle_velocities = out["velocities"]
        self.particles_pressure = out["pressure"]
        self.last = out
        self.tick += 1
