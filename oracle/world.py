"""Host-side world state for the oracle: rigid bodies, particle sources, config.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates

* ``src/crate/rigid_body.py:19-91``   (bodies: placement, motion, point velocities)
* ``src/crate/particle_source.py:8-28`` (binomial emitter on the global NumPy RNG)
* ``src/crate/load_config.py:7-46``   (YAML -> world config)
* ``src/crate/utils/geometry_utils.py:146-179`` (pad_segments, cw90)

Everything here is O(S) or O(new particles) per tick and stays on the host in
the product too.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import yaml


def cw90(v: np.ndarray) -> np.ndarray:
    """(x, y) -> (y, -x) on the last axis.  geometry_utils.py:176-179."""
    return np.stack((v[..., 1], -v[..., 0]), axis=-1)


def rotate_deg(x: float, y: float, angle: float, eps: float = 1e-6):
    """Counter-clockwise rotation by ``angle`` degrees the way pygame's
    ``Vector2.rotate`` does it (rigid_body.py:38-39 calls that): the angle is
    first folded into [0, 360), right angles are exact, everything else goes
    through sin/cos of the folded angle in radians."""
    a = math.fmod(angle, 360.0)
    if a < 0:
        a += 360.0
    if math.fmod(a + eps, 90.0) < 2 * eps:
        q = int((a + eps) / 90.0) % 4
        if q == 0:
            return x, y
        if q == 1:
            return -y, x
        if q == 2:
            return -x, -y
        return y, -x
    r = a * math.pi / 180.0
    s, c = math.sin(r), math.cos(r)
    return c * x - s * y, s * x + c * y


class Body:
    """One rigid body.  kind is 'fixed', 'motored' or 'free' (rigid_body.py:91)."""

    def __init__(self, kind: str, segments, name: str = "", scale=(1.0, 1.0), position=(0.0, 0.0),
                 rotation: float = 0.0, velocity_func=None, angular_velocity_func=None):
        self.kind = kind
        self.name = name
        self.segments = np.array(segments, dtype=np.float64)  # S x 2 x 2
        self.position = np.array(position, dtype=np.float64)
        self.center_velocity = np.array([0.0, 0.0])
        self.angular_clockwise_velocity = 0.0
        self.velocity_func = velocity_func or (lambda t: np.array([0.0, 0.0]))
        self.angular_velocity_func = angular_velocity_func or (lambda t: 0)
        self.time_from_start = 0.0
        # place_in_world, rigid_body.py:36-40: scale, rotate, translate
        self.segments *= np.array(scale, dtype=np.float64)[None]
        for e in (0, 1):
            self.segments[:, e, :] = np.array([rotate_deg(px, py, rotation) for px, py in self.segments[:, e, :]])
        self.segments += self.position[None]

    def __len__(self):
        return len(self.segments)

    def points_velocity(self, pts: np.ndarray) -> np.ndarray:
        """rigid_body.py:28-34: v_c + cw90(pt - position) * omega (small-angle rotation)."""
        return self.center_velocity[None] + cw90(pts - self.position) * self.angular_clockwise_velocity

    def advance(self, dt: float) -> None:
        """rigid_body.py:42-46 (free), :54-55 (fixed: no-op), :64-68 (motored)."""
        if self.kind == "fixed":
            return
        if self.kind == "motored":
            self.time_from_start += dt
            self.center_velocity = self.velocity_func(self.time_from_start)
            self.angular_clockwise_velocity = self.angular_velocity_func(self.time_from_start)
        seg = self.segments.copy()
        seg[:, 0, :] += self.points_velocity(self.segments[:, 0, :]) * dt
        seg[:, 1, :] += self.points_velocity(self.segments[:, 1, :]) * dt
        self.segments = seg


_LAMBDA_SCOPE = {"np": np, "numpy": np, "math": math}


def build_bodies(body_configs) -> list[Body]:
    """rigid_body.py:71-88: one single-key mapping per body, lambda strings eval'd."""
    out = []
    for cfg in copy.deepcopy(body_configs or []):
        kind, kw = next(iter(cfg.items()))
        for key in ("velocity_func", "angular_velocity_func"):
            if key in kw:
                kw[key] = eval(kw[key], dict(_LAMBDA_SCOPE))  # noqa: S307 - same contract as the reference's YAML
        out.append(Body(kind, **kw))
    return out


def pad_segments(segments: np.ndarray, pad: float) -> np.ndarray:
    """geometry_utils.py:146-172: each segment (a,b) becomes (a+o, b+o) and (b-o, a-o),
    o = cw90(b-a)/|b-a| * pad.  Returns 2S x 2 x 2, the +o copies first."""
    a, b = segments[:, 0, :], segments[:, 1, :]
    n = cw90(b - a)
    o = n * pad / np.linalg.norm(n, axis=1)[:, None]
    first = np.stack((a + o, b + o), axis=1)
    second = np.stack((b - o, a - o), axis=1)
    return np.concatenate((first, second), axis=0)


@dataclass
class Source:
    """particle_source.py:8-24."""
    radius: float
    position: list
    velocity: list
    flow: float
    active_ticks: int
    noise: float = 0.05

    def emit(self, dt: float, room: int):
        n = min(np.round(np.random.binomial(self.flow, dt)), room)
        if n == 0:
            return None, None
        pos = (np.random.rand(n, 2) - 0.5) * self.radius + np.array(self.position)
        vel = np.ones_like(pos) * np.array(self.velocity)[None]
        vel += (np.random.rand(n, 2) - 0.5) * self.noise
        return pos, vel


def build_sources(cfgs) -> list[Source]:
    return [Source(**c) for c in (cfgs or [])]


@dataclass
class World:
    rigid_bodies: list
    particle_sources: list
    coefficients: dict


@dataclass
class SceneConfig:
    world: World
    playback: dict = field(default_factory=dict)


def load_scene(path) -> SceneConfig:
    """load_config.py:29-46 (the oracle keeps the playback block as a plain dict)."""
    with open(Path(path), "r") as f:
        raw = yaml.safe_load(f)
    w = raw["world"]
    return SceneConfig(
        world=World(rigid_bodies=w.get("rigid_bodies", []), particle_sources=w.get("particle_sources"),
                    coefficients=w.get("coefficients")),
        playback=raw.get("playback", {}),
    )
