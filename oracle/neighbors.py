"""Strip sort and neighbor lists: restates ``src/crate/collision_detector.py:9-128``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Reference semantics (collision_detector.py):

* ``strip_sort_particles`` (:124-128): ``row = floor(y / d)`` in float64, cast to
  int; ``sorted_indices = lexsort((x, row))`` = stable sort by (row, x), ties by
  original index.
* strips are runs of equal row; the "next strip" of a strip is the next
  NON-EMPTY row (:34-40).
* forward candidates of the particle at sorted position i (:96-121): the rest of
  its strip while ``x_j <= x_i + d`` (searchsorted side="right"), then the next
  strip's ``x_i - d <= x_j <= x_i + d``; kept when ``norm(p_j - p_i) <= d`` (:75-80).
* reverse edges are appended walking particles, and each particle's forward
  list, backwards (:85-88); lists are cut to the first 20 (:6, :91-93) and
  mapped back to original indices (:46-48).

Net order of N(i): forward targets by ascending sorted position, then reverse
sources by descending sorted position, then ``[:20]``.
"""
from __future__ import annotations

import numpy as np

MAX_NEIGHBORS = 20  # collision_detector.py:6


def strip_sort(particles: np.ndarray, d: float):
    """-> (row of each sorted slot, sorted_indices), both int64.  collision_detector.py:124-128."""
    row = np.floor(particles[:, 1] / d).astype(np.int64)
    order = np.lexsort((particles[:, 0], row))
    return row[order], order.astype(np.int64)


def _forward_pairs(xs: np.ndarray, ys: np.ndarray, rows: np.ndarray, d: float):
    """All forward edges (i, j), i < j in sorted positions, that pass the distance filter."""
    n = len(xs)
    if n == 0:
        z = np.zeros(0, dtype=np.int64)
        return z, z
    starts = np.flatnonzero(np.r_[True, rows[1:] != rows[:-1]])
    bounds = np.r_[starts, n, n]  # the last strip's "next strip" is empty (:35-36)
    lo = np.empty(n, dtype=np.int64)   # same-strip candidates: (i, same_hi)
    same_hi = np.empty(n, dtype=np.int64)
    nxt_lo = np.empty(n, dtype=np.int64)
    nxt_hi = np.empty(n, dtype=np.int64)
    for k in range(len(starts)):
        s, e, e2 = bounds[k], bounds[k + 1], bounds[k + 2]
        sx = xs[s:e]
        same_hi[s:e] = s + np.searchsorted(sx, sx + d, side="right")
        nx = xs[e:e2]
        nxt_lo[s:e] = e + np.searchsorted(nx, sx - d, side="left")
        nxt_hi[s:e] = e + np.searchsorted(nx, sx + d, side="right")
    idx = np.arange(n, dtype=np.int64)
    lo[:] = idx + 1
    cnt_same = np.maximum(same_hi - lo, 0)
    cnt_next = np.maximum(nxt_hi - nxt_lo, 0)

    def expand(first, count):
        tot = int(count.sum())
        src = np.repeat(idx, count)
        off = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(count) - count, count)
        return src, np.repeat(first, count) + off

    i1, j1 = expand(lo, cnt_same)
    i2, j2 = expand(nxt_lo, cnt_next)
    i = np.concatenate((i1, i2))
    j = np.concatenate((j1, j2))
    dx = xs[j] - xs[i]
    dy = ys[j] - ys[i]
    keep = np.sqrt(dx * dx + dy * dy) <= d  # np.linalg.norm(axis=1) of a 2-vector
    return i[keep], j[keep]


def neighbor_lists_sorted(xs, ys, rows, d: float, max_neighbors: int = MAX_NEIGHBORS):
    """Neighbor lists in SORTED index space: (counts int32[n], table int64[n, max] padded with -1)."""
    n = len(xs)
    fi, fj = _forward_pairs(xs, ys, rows, d)
    src = np.concatenate((fi, fj))
    dst = np.concatenate((fj, fi))
    kind = np.concatenate((np.zeros(len(fi), np.int64), np.ones(len(fj), np.int64)))
    within = np.where(kind == 0, dst, -dst)  # forward: ascending target; reverse: descending source
    order = np.lexsort((within, kind, src))
    src, dst = src[order], dst[order]
    total = np.bincount(src, minlength=n).astype(np.int64)
    first = np.cumsum(total) - total
    rank = np.arange(len(src), dtype=np.int64) - first[src]
    keep = rank < max_neighbors
    table = np.full((n, max_neighbors), -1, dtype=np.int64)
    table[src[keep], rank[keep]] = dst[keep]
    return np.minimum(total, max_neighbors).astype(np.int32), table


def neighbor_lists(particles: np.ndarray, d: float, max_neighbors: int = MAX_NEIGHBORS):
    """Neighbor lists in ORIGINAL index space, as ``detect_particle_collisions`` returns them
    (collision_detector.py:9-49) but padded: (counts int32[P], table int64[P, max], -1 padded)."""
    particles = np.asarray(particles, dtype=np.float64)
    n = len(particles)
    if n == 0:
        return np.zeros(0, np.int32), np.full((0, max_neighbors), -1, np.int64)
    rows, order = strip_sort(particles, d)
    xs = particles[order, 0]
    ys = particles[order, 1]
    c_s, t_s = neighbor_lists_sorted(xs, ys, rows, d, max_neighbors)
    counts = np.empty(n, dtype=np.int32)
    table = np.full((n, max_neighbors), -1, dtype=np.int64)
    counts[order] = c_s
    mapped = np.where(t_s >= 0, order[np.maximum(t_s, 0)], -1)
    table[order] = mapped
    return counts, table


def as_python_lists(counts: np.ndarray, table: np.ndarray) -> list[list[int]]:
    """Padded table -> the reference's list-of-lists shape."""
    return [table[i, : counts[i]].tolist() for i in range(len(counts))]
