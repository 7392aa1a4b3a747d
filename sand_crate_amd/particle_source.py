"""Particle emitters: host-side, a handful of particles per tick.

Same behaviour as the reference's ``src/crate/particle_source.py:8-28``: per tick a source draws
``binomial(flow, dt)`` new particles from the GLOBAL legacy NumPy RNG (seeded by ``Crate``,
crate.py:22), then two ``rand(n, 2)`` blocks for position and velocity jitter, in that order.  The
draws stay on the host so the stream interleaves with the collider noise exactly as in the
reference."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class ParticleSource:
    radius: float
    position: list
    velocity: list
    flow: float
    active_ticks: int
    noise: float = 0.05

    def generate_particles(self, dt: float, max_particles: int) -> tuple[Optional[np.ndarray], Optional[np.ndarray]]:
        count = min(np.round(np.random.binomial(self.flow, dt)), max_particles)
        if count == 0:
            return None, None
        jitter = np.random.rand(count, 2)
        positions = (jitter - 0.5) * self.radius + np.array(self.position)
        velocities = np.ones_like(positions) * np.array(self.velocity)[None]
        velocities += (np.random.rand(count, 2) - 0.5) * self.noise
        return positions, velocities


def build_particle_sources(particle_source_configs) -> list[ParticleSource]:
    return [ParticleSource(**cfg) for cfg in (particle_source_configs or [])]
