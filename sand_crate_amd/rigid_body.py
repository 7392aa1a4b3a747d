"""Rigid bodies: the host-side part of the tick (O(S), S <= 16 wall segments).

Mirrors the behaviour of the reference's ``src/crate/rigid_body.py`` -- placement (:36-40),
per-tick motion (:42-46, :64-68), velocity of points on a body (:28-34), YAML construction with
``eval``'d lambda strings (:71-88) -- without pygame.  What a tick needs from here travels to the
GPU as kernel arguments: the stacked segment endpoints and, per body, position, centre velocity
and angular velocity (include/sandcrate_hip.h: sc_set_segments).
"""
from __future__ import annotations

import copy
import math

import numpy as np


_CW90 = np.array([1.0, -1.0])


def _rotate_ccw_degrees(points: np.ndarray, angle: float) -> np.ndarray:
    """Rotation as pygame.Vector2.rotate performs it (rigid_body.py:38-39): fold the angle into
    [0, 360), treat right angles exactly, otherwise sin/cos of the folded angle in radians."""
    eps = 1e-6
    a = math.fmod(angle, 360.0)
    if a < 0:
        a += 360.0
    x, y = points[:, 0].copy(), points[:, 1].copy()
    if math.fmod(a + eps, 90.0) < 2 * eps:
        quarter = int((a + eps) / 90.0) % 4
        rx, ry = [(x, y), (-y, x), (-x, -y), (y, -x)][quarter]
    else:
        rad = a * math.pi / 180.0
        s, c = math.sin(rad), math.cos(rad)
        rx, ry = c * x - s * y, s * x + c * y
    return np.stack((rx, ry), axis=1)


class RigidBody:
    """A free body: moves with its own centre and angular velocity; gravity accelerates it."""

    moves = True
    driven = False

    def __init__(self, segments, name: str = "", scale=(1.0, 1.0), position=(0.0, 0.0), rotation: float = 0.0,
                 center_velocity=(0.0, 0.0), angular_clockwise_velocity: float = 0.0):
        self.name = name
        self.scale = list(scale)
        self.position = list(position)
        self.rotation = rotation
        self.center_velocity = np.array(center_velocity, dtype=np.float64)
        self.angular_clockwise_velocity = angular_clockwise_velocity
        seg = np.array(segments, dtype=np.float64) * np.array(self.scale, dtype=np.float64)[None]
        for end in (0, 1):
            seg[:, end, :] = _rotate_ccw_degrees(seg[:, end, :], rotation)
        self.segments = seg + np.array(self.position, dtype=np.float64)[None]

    def __len__(self) -> int:
        return len(self.segments)

    def calc_body_points_velocities(self, body_points: np.ndarray) -> np.ndarray:
        rel = body_points - np.asarray(self.position, dtype=np.float64)
        tangent = np.stack((rel[:, 1], -rel[:, 0]), axis=1)  # clockwise quarter turn
        return np.asarray(self.center_velocity, dtype=np.float64)[None] + tangent * self.angular_clockwise_velocity

    def apply_velocity(self, dt: float) -> None:
        # rigid_body.py:42-46: every end point += (center_velocity + cw90(point - position) * omega) * dt,
        # both ends of all segments in one pass (the same float64 operations per element)
        # (few NumPy calls: this runs once per body and tick on the host, next to a GPU tick of tens of microseconds)
        seg = self.segments
        rel = seg - np.asarray(self.position, dtype=np.float64)
        tangent = rel[..., ::-1] * _CW90  # clockwise quarter turn (y, -x): the products by 1 and -1 are exact
        tangent *= self.angular_clockwise_velocity
        tangent += np.asarray(self.center_velocity, dtype=np.float64)  # = center_velocity + tangent * omega
        tangent *= dt
        self.segments = seg + tangent


class FixedRigidBody(RigidBody):
    moves = False

    def apply_velocity(self, dt: float) -> None:
        return None


class MotoredRigidBody(RigidBody):
    driven = True

    def __init__(self, *args, velocity_func=None, angular_velocity_func=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.velocity_func = velocity_func or (lambda t: np.array([0.0, 0.0]))
        self.angular_velocity_func = angular_velocity_func or (lambda t: 0)
        self.time_from_start = 0.0

    def apply_velocity(self, dt: float) -> None:
        self.time_from_start += dt
        self.center_velocity = self.velocity_func(self.time_from_start)
        self.angular_clockwise_velocity = self.angular_velocity_func(self.time_from_start)
        super().apply_velocity(dt)


BODY_TYPE_TO_CLASS = {"motored": MotoredRigidBody, "fixed": FixedRigidBody, "free": RigidBody}
_EVAL_SCOPE = {"np": np, "numpy": np, "math": math}


def build_rigid_bodies(body_configs: list) -> list[RigidBody]:
    bodies = []
    for entry in copy.deepcopy(body_configs or []):
        kind, kwargs = next(iter(entry.items()))
        for key in ("velocity_func", "angular_velocity_func"):
            if key in kwargs:
                kwargs[key] = eval(kwargs[key], dict(_EVAL_SCOPE))  # noqa: S307 - the scene file's contract
        bodies.append(BODY_TYPE_TO_CLASS[kind](**kwargs))
    return bodies
