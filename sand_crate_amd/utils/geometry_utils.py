"""Geometry helpers of the tick.

* ``pad_segments`` / ``rotate_vectors_clockwise_90_deg`` run on the host: they are O(S) and
  their result is a kernel argument (geometry_utils.py:146-179 in the reference).
* ``points_to_segments_distance`` (geometry_utils.py:7-39) runs on the GPU through
  ``sc_points_to_segments``; the reference's own test pins it (tests/test_distance.py:16-25).
"""
from __future__ import annotations

import numpy as np

from .. import _native as N


def rotate_vectors_clockwise_90_deg(vectors: np.ndarray) -> np.ndarray:
    return np.stack((vectors[:, 1], -vectors[:, 0]), axis=1)


def pad_segments(segments: np.ndarray, pad_distance: float) -> np.ndarray:
    """Two parallel copies of every segment at +-pad_distance: first all (a+o, b+o), then all
    (b-o, a-o), with o = cw90(b - a) * pad_distance / |b - a| -- computed by the library on the host
    (sc_pad_segments: the reference's operations in its order; `pad_segments_numpy` below is the same in NumPy,
    kept as the check of it)."""
    seg = N.f64(segments).reshape(-1, 2, 2)
    out = np.empty((2 * len(seg), 2, 2))
    N.check(N.load().sc_pad_segments(N.dptr(seg), len(seg), float(pad_distance), N.dptr(out)))
    return out


def pad_segments_numpy(segments: np.ndarray, pad_distance: float) -> np.ndarray:
    segments = np.asarray(segments, dtype=np.float64)
    n = len(segments)
    start, end = segments[:, 0, :], segments[:, 1, :]
    along = end - start
    nx, ny = along[:, 1], -along[:, 0]            # clockwise quarter turn of (end - start)
    norm = np.sqrt(nx * nx + ny * ny)             # = np.linalg.norm(normal, axis=1) for two components
    ox, oy = nx * pad_distance / norm, ny * pad_distance / norm
    out = np.empty((2 * n, 2, 2))
    out[:n, :, 0] = segments[:, :, 0] + ox[:, None]
    out[:n, :, 1] = segments[:, :, 1] + oy[:, None]
    out[n:, 0, 0] = end[:, 0] - ox
    out[n:, 0, 1] = end[:, 1] - oy
    out[n:, 1, 0] = start[:, 0] - ox
    out[n:, 1, 1] = start[:, 1] - oy
    return out


def points_to_segments_distance(p, segments, device: int = 0):
    """-> (nearest point on each segment, P x S x 2; distance, P x S), computed on the GPU."""
    pts = N.f64(p).reshape(-1, 2)
    seg = N.f64(segments).reshape(-1, 2, 2)
    nearest = np.empty((len(pts), len(seg), 2))
    dist = np.empty((len(pts), len(seg)))
    N.check(N.load().sc_points_to_segments(device, N.dptr(pts), len(pts), N.dptr(seg), len(seg), N.dptr(nearest),
                                           N.dptr(dist)))
    return nearest, dist
