"""One physics tick, vectorised over padded P x 20 neighbor arrays.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates
``src/crate/crate.py:91-129`` (``Crate.physics_tick``) from the point where new
particles have been added and bodies have been advanced.  Every function cites
the reference lines it follows.  All arithmetic is float64, like the reference
(crate.py:24-26).

Decisions (which wall a particle touches, who is a neighbor, which row a
particle is in, which padded segment a movement crosses) are evaluated with the
reference's exact operation order, so they are bit-identical; sums over
neighbors may be associated differently (float-summation note).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .neighbors import MAX_NEIGHBORS, neighbor_lists
from .world import cw90, pad_segments


@dataclass
class BodyState:
    """What a tick needs from one rigid body after ``advance``: rigid_body.py:28-34."""
    position: np.ndarray         # (2,)
    center_velocity: np.ndarray  # (2,)
    omega: float                 # angular_clockwise_velocity
    n_segments: int


# Counter-based collider noise used when the stream is not the host's MT19937.
# The HIP kernels implement the same function (sand_crate_amd/csrc/sc_device.h: noise_base, noise_eta).
_M1 = np.uint64(0xFF51AFD7ED558CCD)
_M2 = np.uint64(0xC4CEB9FE1A85EC53)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_MIX = np.uint64(0xD6E8FEB86659FD93)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z ^ (z >> np.uint64(33))
    z = z * _M1
    z = z ^ (z >> np.uint64(33))
    z = z * _M2
    z = z ^ (z >> np.uint64(33))
    return z


def counter_noise_key(seed: int, tick: int) -> np.uint64:
    with np.errstate(over="ignore"):
        return _mix64(np.array([np.uint64(seed) + np.uint64(tick + 1) * _GOLD], dtype=np.uint64))[0]


def counter_noise_u01(ids: np.ndarray, key: np.uint64) -> np.ndarray:
    """Uniform [0,1) pairs for every (particle id, slot): returns (P, 20, 2)."""
    with np.errstate(over="ignore"):
        slot = np.arange(MAX_NEIGHBORS, dtype=np.uint64)[None, :]
        ctr = ids.astype(np.uint64)[:, None] * np.uint64(32) + slot
        h = ctr * _GOLD + key
        h = h ^ (h >> np.uint64(32))
        h = h * _MIX
        h = h ^ (h >> np.uint64(32))
    hi = (h >> np.uint64(32)).astype(np.float64)
    lo = (h & np.uint64(0xFFFFFFFF)).astype(np.float64)
    return np.stack((hi, lo), axis=-1) * (1.0 / 4294967296.0)


def remove_outside(particles, velocities, r, ids=None):
    """crate.py:149-159: drop any particle with a coordinate < -r or > 1 + r (NaN stays)."""
    gone = np.where((particles < -r) | (particles > 1 + r))[0]
    keep = np.ones(len(particles), dtype=bool)
    keep[gone] = False
    if ids is None:
        return particles[keep], velocities[keep]
    return particles[keep], velocities[keep], ids[keep]


def closest_points_on_segments(p: np.ndarray, segments: np.ndarray):
    """geometry_utils.py:7-39: nearest point on each segment (P x S x 2) and its distance (P x S)."""
    a = segments[:, 0, :]
    b = segments[:, 1, :]
    ab = (b - a)[None]
    ap = p[:, None] - a[None]
    t = (ap * ab).sum(2) / (ab * ab).sum(2)
    t = np.clip(t, 0, 1)
    c = ab * t[:, :, None] + a[None]
    pc = c - p[:, None]
    dist = np.sqrt(pc[..., 0] * pc[..., 0] + pc[..., 1] * pc[..., 1])
    return c, dist


def wall_contacts(p, segments, bodies, r):
    """crate.py:213-243 + :73-85.  Returns per particle, padded over S contact slots in
    segment order: count V (P,), u (P,S,2) = 2*(p - c), contact-point velocity (P,S,2)."""
    P, S = len(p), len(segments)
    if P == 0 or S == 0:
        return np.zeros(P, np.int64), np.zeros((P, S, 2)), np.zeros((P, S, 2))
    c, dist = closest_points_on_segments(p, segments)
    touch = dist <= r * 1.2                                  # crate.py:229
    V = touch.sum(1)
    slot = np.cumsum(touch, axis=1) - 1                      # contact slot of each touching segment
    u = np.zeros((P, S, 2))
    cp = np.zeros((P, S, 2))
    pi, si = np.nonzero(touch)
    u[pi, slot[pi, si]] = (p[pi] - c[pi, si]) * 2            # crate.py:234 (not normalised)
    cp[pi, slot[pi, si]] = c[pi, si]
    # crate.py:73-85: `calculated_points` never advances, so each body with n_b touching
    # segments writes slots [0:n_b] from contact points [0:n_b]; later bodies overwrite.
    vel = np.zeros((P, S, 2))
    seg0 = 0
    q = np.arange(S)[None, :]
    for body in bodies:
        n_b = touch[:, seg0:seg0 + body.n_segments].sum(1)
        seg0 += body.n_segments
        if not n_b.any():
            continue
        bv = body.center_velocity[None, None] + cw90(cp - body.position) * body.omega
        vel = np.where((q < n_b[:, None])[..., None], bv, vel)
    return V, u, vel


def hard_wall_fix(p, V, u, r):
    """crate.py:202-211: p += sum_k u_k * (max(r/|u_k|, 0.5) - 0.5), contacts in slot order."""
    out = p.copy()
    S = u.shape[1]
    corr = np.zeros_like(p)
    with np.errstate(divide="ignore", invalid="ignore"):
        for k in range(S):
            has = V > k
            if not has.any():
                break
            uk = u[:, k]
            rel = r / np.sqrt(uk[:, 0] * uk[:, 0] + uk[:, 1] * uk[:, 1])
            rel = np.where(rel < 0.5, 0.5, rel)
            term = uk * (rel[:, None] - 0.5)
            corr = np.where(has[:, None], corr + term, corr)
    touched = V > 0
    out[touched] = p[touched] + corr[touched]
    return out


def continuous_collision_factors(p, v, segments, r, dt):
    """crate.py:177-200 with geometry_utils.py:136-143, :182-222: per-particle velocity scale."""
    P = len(p)
    fac = np.ones(P)
    if P == 0 or len(segments) == 0:
        return fac
    pad = pad_segments(segments, r)
    a = p
    b = p + v * dt
    c = pad[:, 0, :]
    d = pad[:, 1, :]

    def orient(p_, q_, r_):
        return np.sign((q_[:, 1, None] - p_[:, 1, None]) * (r_[None, :, 0] - q_[:, 0, None])
                       - (q_[:, 0, None] - p_[:, 0, None]) * (r_[None, :, 1] - q_[:, 1, None]))

    opposite = np.sum(cw90(d - c)[None] * (b - a)[:, None], axis=2) < 0
    crossing = np.logical_and(orient(a, b, c) != orient(a, b, d), (orient(c, d, a) != orient(c, d, b)).T)
    pi, si = np.where(np.logical_and(crossing, opposite))
    if len(pi) == 0:
        return fac
    c1 = pad[si, 0, :]
    cd = pad[si, 1, :] - c1
    ab = v[pi] * dt
    ac = p[pi] - c1
    with np.errstate(divide="ignore", invalid="ignore"):
        f = (ac[:, 0] * cd[:, 1] - ac[:, 1] * cd[:, 0]) / (cd[:, 0] * ab[:, 1] - cd[:, 1] * ab[:, 0])
    np.fmin.at(fac, pi, f)
    return fac


def tick_core(particles, velocities, segments, bodies, coef, eta_u01=None, neighbor_fn=neighbor_lists):
    """One tick from "bodies advanced" to "positions integrated": crate.py:97-125.

    particles, velocities : (P,2) float64, original particle order (not modified)
    segments              : (S,2,2) float64, all bodies stacked (crate.py:69-71)
    bodies                : list[BodyState] in body order
    coef                  : dict with dt, particle_radius, wall_collision_decay, pressure_amplifier,
                            ignored_pressure, collider_noise_level, viscosity, surface_smoothing,
                            target_pressure, gravity
    eta_u01               : None (no noise), or uniform [0,1) numbers, either (sum C_i, 2) in
                            particle-major/slot-minor order -- what the reference draws per particle at
                            crate.py:169 -- or already padded (P, 20, 2), or a callable total -> (total, 2)
                            invoked once the neighbor counts are known
    Returns a dict: particles, velocities, pressure and the intermediates tests compare.
    """
    p0 = np.asarray(particles, dtype=np.float64)
    v0 = np.asarray(velocities, dtype=np.float64)
    P = len(p0)
    r = coef["particle_radius"]
    d = r * 2
    dt = coef["dt"]
    g = np.asarray(coef["gravity"], dtype=np.float64)

    # "Virtual Colliders": crate.py:97-99
    V, u, wall_vel = wall_contacts(p0, segments, bodies, r)
    p = hard_wall_fix(p0, V, u, r)

    # "Collisions": crate.py:101-102 (on post-fix positions)
    counts, table = neighbor_fn(p, d)
    K = table.shape[1] if P else MAX_NEIGHBORS
    valid = np.arange(K)[None, :] < counts[:, None]
    nb = np.where(valid, table, 0)

    # "Colliders": crate.py:161-175
    eta = np.zeros((P, K, 2))
    if callable(eta_u01):
        eta_u01 = eta_u01(int(counts.sum()))
    if eta_u01 is not None:
        eta_u01 = np.asarray(eta_u01, dtype=np.float64)
        if eta_u01.ndim == 2:
            eta[valid] = (eta_u01 - 0.5) * d * coef["collider_noise_level"]
        else:
            eta = np.where(valid[..., None], (eta_u01 - 0.5) * d * coef["collider_noise_level"], 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = p[:, None, :] - (p[nb] + eta)
        dist = np.sqrt(rel[..., 0] * rel[..., 0] + rel[..., 1] * rel[..., 1])
        n = rel / dist[..., None]
    n = np.where(valid[..., None], n, 0.0)
    v_snap = v0[nb]                                              # crate.py:175 (copy at tick start)

    # "Pressure": crate.py:261-284
    w = np.where(valid, 1 - np.clip(dist / d, 0, 1), 0.0)
    pressure = np.where(counts > 0, np.maximum(0, w.sum(1) - coef["ignored_pressure"]), 0.0)
    p_nb = np.where(valid, pressure[nb], 0.0)

    # "tension": crate.py:335-353
    s = (((1 - w) * w)[..., None] * n).sum(1)
    v = v0.copy()
    align = ((s[:, None, :] - s[nb]) * n).sum(2) * coef["surface_smoothing"]
    fix = p_nb + pressure[:, None] - 2 * coef["target_pressure"]
    v = v + dt * (np.where(valid, align + fix, 0.0)[..., None] * n).sum(1)
    v_after_tension = v.copy()

    # "gravity": crate.py:309-310
    v = v + dt * g[None]

    # "pressure": crate.py:295-307 (wall colliders joined the lists at :286-293 with pressure 0)
    Sn = u.shape[1]
    wall_u = (u * (np.arange(Sn)[None, :] < V[:, None])[..., None]).sum(1) if Sn else np.zeros((P, 2))
    push = ((pressure[:, None] + p_nb)[..., None] * n).sum(1) + pressure[:, None] * wall_u
    v = v + dt * coef["pressure_amplifier"] * push
    v_after_pressure = v.copy()

    # "viscosity": crate.py:316-323 (snapshot neighbors, current self)
    v = v + dt * coef["viscosity"] * np.where(valid[..., None], v_snap - v[:, None, :], 0.0).sum(1)
    v_after_viscosity = v.copy()

    # "wall_bounce": crate.py:245-259
    touched = V > 0
    if touched.any():
        Vt = V[touched].astype(np.float64)[:, None]
        in_slot = (np.arange(Sn)[None, :] < V[touched][:, None])[..., None]
        normal = (u[touched] * in_slot).sum(1) / Vt
        cv = (wall_vel[touched] * in_slot).sum(1) / Vt
        with np.errstate(divide="ignore", invalid="ignore"):
            nhat = normal / np.sqrt(normal[:, 0] * normal[:, 0] + normal[:, 1] * normal[:, 1])[:, None]
        relv = v[touched] - cv
        q = relv[:, 0] * nhat[:, 0] + relv[:, 1] * nhat[:, 1]
        counter = -1 * q[:, None] * nhat
        vt = v[touched]
        hit = q < 0
        vt[hit] = vt[hit] + counter[hit]
        vt[hit] = vt[hit] + counter[hit] * coef["wall_collision_decay"]
        v[touched] = vt
    v_after_bounce = v.copy()

    # "continuous_collision": crate.py:177-200
    fac = continuous_collision_factors(p, v, segments, r, dt)
    v = v * fac[:, None]

    # crate.py:360-361
    p_new = p + dt * v
    return {
        "particles": p_new, "velocities": v, "pressure": pressure,
        "fixed_positions": p, "wall_count": V, "wall_u": u, "wall_vel": wall_vel,
        "neighbor_counts": counts, "neighbor_table": table,
        "surface_normals": s, "v_after_tension": v_after_tension, "v_after_pressure": v_after_pressure,
        "v_after_viscosity": v_after_viscosity, "v_after_bounce": v_after_bounce, "ccd_factor": fac,
    }
