"""Neighbor search entry points with the reference's signatures
(``src/crate/collision_detector.py:9-49`` and ``:124-128``), computed by the HIP path."""
from __future__ import annotations

import numpy as np

from . import _native as N

MAX_ALLOWED_NEIGHBORS = N.MAX_NEIGHBORS


def neighbor_search(particles, diameter: float, device: int = 0):
    """-> y_floored (P,), sorted_indices (P,), counts (P,), table (P,20; -1 padded)."""
    pts = N.f64(particles).reshape(-1, 2)
    n = len(pts)
    rows = np.empty(n, dtype=np.int64)
    order = np.empty(n, dtype=np.int64)
    counts = np.zeros(n, dtype=np.int32)
    table = np.full((n, N.MAX_NEIGHBORS), -1, dtype=np.int64)
    N.check(N.load().sc_neighbor_search(device, N.dptr(pts), n, float(diameter), N.i64ptr(rows), N.i64ptr(order),
                                        N.i32ptr(counts), N.i64ptr(table)))
    return rows, order, counts, table


def detect_particle_collisions(particles, diameter: float) -> list[list[int]]:
    """Per particle, the indices of the particles within ``diameter`` in the reference's order,
    at most 20 each."""
    _, _, counts, table = neighbor_search(particles, diameter)
    return [table[i, : counts[i]].tolist() for i in range(len(counts))]


def strip_sort_particles(particles, diameter: float):
    """-> (particles in strip order, their row index, the permutation), like the reference."""
    pts = N.f64(particles).reshape(-1, 2)
    rows, order, _, _ = neighbor_search(pts, diameter)
    return pts[order, :], rows, order
