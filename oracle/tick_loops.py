"""The tick written with the reference's cost structure: one Python-level loop iteration per
particle in every phase, small NumPy arrays inside (ragged per-particle lists, not padded tables).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``bench.py`` times this as the stand-in for
"the reference NumPy path" on the GPU box's host (the reference itself cannot travel there);
``tests/test_oracle_golden.py`` checks it against the same golden vectors as ``oracle.tick``.
It follows ``crate.py:97-125`` phase by phase and ``collision_detector.py:9-121`` for the search.
Single-threaded by construction, like the reference.
"""
from __future__ import annotations

import numpy as np

from .neighbors import MAX_NEIGHBORS
from .tick import closest_points_on_segments, continuous_collision_factors
from .world import cw90


def neighbor_lists_loops(pos: np.ndarray, d: float) -> list[list[int]]:
    """collision_detector.py:9-121: strip sort, per-particle searchsorted windows, distance filter,
    reverse edges, trim, back to original indices."""
    n = len(pos)
    if n == 0:
        return []
    row = np.floor(pos[:, 1] / d).astype(int)
    order = np.lexsort((pos[:, 0], row))
    sp, srow = pos[order], row[order]
    _, first = np.unique(srow, return_index=True)
    bounds = list(first) + [n, n]
    found: list[list[int]] = []
    for s, e, e2 in zip(bounds, bounds[1:], bounds[2:]):
        xs, nxt = sp[s:e, 0], sp[e:e2, 0]
        for k, x in enumerate(xs):
            hi = np.searchsorted(xs, x + d, side="right")
            lo2 = np.searchsorted(nxt, x - d, side="left")
            hi2 = np.searchsorted(nxt, x + d, side="right")
            cand = np.array(list(range(s + k + 1, s + hi)) + list(range(e + lo2, e + hi2)))
            if len(cand):
                gap = np.linalg.norm(sp[cand, :] - sp[s + k, :], axis=1)
                cand = cand[gap <= d]
            found.append(cand.tolist())
    for i in range(n - 1, -1, -1):
        for j in reversed(found[i]):
            found[j].append(i)
    found = [f[:MAX_NEIGHBORS] for f in found]
    back = np.argsort(order)
    return [[int(order[j]) for j in found[k]] for k in back]


def tick_loops(particles, velocities, segments, bodies, coef, eta_source=None):
    """Same contract as ``oracle.tick.tick_core`` (eta_source: callable total -> (total, 2) uniforms,
    or None for no noise).  Returns particles, velocities, pressure."""
    pos = np.array(particles, dtype=np.float64)
    vel = np.array(velocities, dtype=np.float64)
    P = len(pos)
    r, dt = coef["particle_radius"], coef["dt"]
    d = r * 2
    g = np.asarray(coef["gravity"], dtype=np.float64)

    # Virtual colliders + hard wall fix (crate.py:202-243, :73-85)
    wall_u, wall_v = [], []
    if P and len(segments):
        near, dist = closest_points_on_segments(pos, segments)
    for i in range(P):
        if not len(segments):
            wall_u.append(np.empty((0, 2)))
            wall_v.append(np.empty((0, 2)))
            continue
        touch = dist[i] <= r * 1.2
        contact = near[i, touch]
        if len(contact):
            wall_u.append((pos[i] - contact) * 2)
            cv = np.zeros_like(contact)
            seg0 = 0
            for b in bodies:
                nb = int(np.sum(touch[seg0:seg0 + b.n_segments]))
                seg0 += b.n_segments
                if nb:
                    cv[:nb] = b.center_velocity[None] + cw90(contact[:nb] - b.position) * b.omega
            wall_v.append(cv)
        else:
            wall_u.append(np.empty((0, 2)))
            wall_v.append(np.empty((0, 2)))
    for i in range(P):
        if len(wall_u[i]) == 0:
            continue
        rel = r / np.linalg.norm(wall_u[i], axis=1)
        rel[rel < 0.5] = 0.5
        pos[i] += np.sum(wall_u[i] * (rel[:, None] - 0.5), axis=0)

    # Collisions + colliders (crate.py:101-104, :161-175)
    nbrs = neighbor_lists_loops(pos, d)
    total = sum(len(x) for x in nbrs)
    eta = None if eta_source is None else np.asarray(eta_source(total))
    unit, gap, snap = [], [], []
    cursor = 0
    for i in range(P):
        idx = nbrs[i]
        other = pos[idx]
        if eta is not None:
            other = other + (eta[cursor:cursor + len(idx)] - 0.5) * d * coef["collider_noise_level"]
            cursor += len(idx)
        rel = pos[i] - other
        dist_i = np.linalg.norm(rel, axis=1) if len(idx) else np.zeros(0)
        gap.append(dist_i)
        unit.append(rel / dist_i[:, None] if len(idx) else np.empty((0, 2)))
        snap.append(vel[idx])

    # Pressure (crate.py:261-284)
    pressure = np.zeros(P)
    overlap = []
    for i in range(P):
        if len(nbrs[i]) == 0:
            overlap.append(np.array([]))
            continue
        w = 1 - np.clip(gap[i] / d, 0, 1)
        overlap.append(w)
        pressure[i] = np.maximum(0, np.sum(w, 0) - coef["ignored_pressure"])
    nb_pressure = [pressure[nbrs[i]] for i in range(P)]

    # tension (crate.py:335-353)
    normal = np.zeros((P, 2))
    for i in range(P):
        if len(nbrs[i]):
            normal[i] = np.sum(((1 - overlap[i]) * overlap[i])[:, None] * unit[i], 0)
    for i in range(P):
        if len(nbrs[i]) == 0:
            continue
        delta = normal[i][None] - normal[nbrs[i]]
        align = np.sum(delta * unit[i], 1) * coef["surface_smoothing"]
        fix = nb_pressure[i] + pressure[i] - 2 * coef["target_pressure"]
        vel[i] += dt * np.sum((align + fix)[:, None] * unit[i], 0)
    # gravity (crate.py:309-310)
    vel += dt * g[None]
    # pressure (crate.py:295-307; wall colliders joined with pressure 0, :286-293)
    for i in range(P):
        if len(nbrs[i]) + len(wall_u[i]) == 0:
            continue
        dirs = np.concatenate((unit[i], wall_u[i]))
        pj = np.append(nb_pressure[i], [0] * len(wall_u[i]))
        vel[i] += dt * coef["pressure_amplifier"] * np.sum(dirs * (pressure[i] + pj)[:, None], 0)
    # viscosity (crate.py:316-323)
    for i in range(P):
        vel[i] += dt * coef["viscosity"] * np.sum(snap[i] - vel[i], 0)
    # wall bounce (crate.py:245-259)
    for i in range(P):
        if len(wall_u[i]) == 0:
            continue
        nrm = np.mean(wall_u[i], 0)
        nrm = nrm / np.linalg.norm(nrm)
        q = np.dot(vel[i] - np.mean(wall_v[i], 0), nrm)
        if q < 0:
            push = -1 * q * nrm
            vel[i] += push
            vel[i] += push * coef["wall_collision_decay"]
    # continuous collision (crate.py:177-200) and integration (:360-361)
    vel *= continuous_collision_factors(pos, vel, segments, r, dt)[:, None]
    pos += dt * vel
    return {"particles": pos, "velocities": vel, "pressure": pressure}
