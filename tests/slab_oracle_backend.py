"""A NumPy stand-in for `sand_crate_amd.slab.HipSlabBackend`, built on the CPU oracle, so that the
N > 1 host logic of `SlabCrate` (cuts, halo exchange over torch.distributed, migration, ownership)
runs without a GPU under gloo.  TEST INFRASTRUCTURE: lives in tests/, never imported by the product."""
import numpy as np
import torch

from oracle.tick import BodyState, counter_noise_key, counter_noise_u01, remove_outside, tick_core
from sand_crate_amd.slab import HALO_FIELDS, column_of


class OracleSlabBackend:
    engine = None

    def __init__(self, halo_capacity: int, noise: str, noise_seed: int):
        self.cap = int(halo_capacity)
        self.noise, self.seed = noise, noise_seed
        shape = ((self.cap + 1) * HALO_FIELDS,)
        self.send_left = torch.zeros(shape, dtype=torch.float64)
        self.send_right = torch.zeros(shape, dtype=torch.float64)
        self.recv_left = torch.zeros(shape, dtype=torch.float64)
        self.recv_right = torch.zeros(shape, dtype=torch.float64)
        self.tick = 0
        self.pressure = np.zeros(0)

    def load(self, particles, velocities, ids):
        self.p, self.v, self.ids = particles.copy().reshape(-1, 2), velocities.copy().reshape(-1, 2), ids.copy()
        self.pressure = np.zeros(len(ids))

    def append(self, particles, velocities, ids):
        self.p = np.vstack((self.p, particles))
        self.v = np.vstack((self.v, velocities))
        self.ids = np.concatenate((self.ids, np.asarray(ids, dtype=np.int64)))

    def set_axis(self, axis):
        self.axis = int(axis)

    def set_slab(self, lo, hi, halo, has_left, has_right):
        self.lo, self.hi, self.halo, self.has_left, self.has_right = lo, hi, halo, has_left, has_right

    def set_tick_inputs(self, coef, gravity, segments, padded, bodies):
        self.coef = dict(coef, gravity=np.asarray(gravity, dtype=np.float64))
        self.segments = np.asarray(segments, dtype=np.float64)
        self.bodies = [BodyState(np.asarray(p, float), np.asarray(v, float), float(w), int(n)) for p, v, w, n in bodies]

    def _fill(self, tensor, mask):
        rec = np.column_stack((self.p[mask], self.v[mask], self.ids[mask].astype(np.float64)))
        assert len(rec) <= self.cap, "halo buffer too small"
        buf = np.zeros((self.cap + 1, HALO_FIELDS))
        buf[0, 0] = len(rec)
        buf[1:1 + len(rec)] = rec
        tensor.copy_(torch.from_numpy(buf.reshape(-1)))

    def pack(self):
        col = column_of(self.p[:, self.axis], 2 * self.coef["particle_radius"])
        self._fill(self.send_left, (col < self.lo + self.halo) if self.has_left else np.zeros(len(col), bool))
        self._fill(self.send_right, (col >= self.hi - self.halo) if self.has_right else np.zeros(len(col), bool))

    def message_sizes(self, whole=False):
        return (self.cap,) * 4  # the lagged sizes are the library's business (sc_halo_sizes); here: whole buffers

    def column_histogram(self, col0, n_columns):
        col = column_of(self.p[:, self.axis], 2 * self.coef["particle_radius"])
        return np.bincount(np.clip(col - col0, 0, n_columns - 1), minlength=n_columns).astype(np.int64)

    def unpack(self, from_left, from_right, sizes=None):
        for use, tensor in ((from_left, self.recv_left), (from_right, self.recv_right)):
            if not use:
                continue
            buf = tensor.numpy().reshape(-1, HALO_FIELDS)
            n = int(buf[0, 0])
            rec = buf[1:1 + n]
            self.p = np.vstack((self.p, rec[:, 0:2]))
            self.v = np.vstack((self.v, rec[:, 2:4]))
            self.ids = np.concatenate((self.ids, rec[:, 4].astype(np.int64)))

    def step(self, next_inputs=None):  # no look-ahead here: SlabCrate packs explicitly every tick
        c = self.coef
        p, v, ids = remove_outside(self.p, self.v, c["particle_radius"], self.ids)
        col = column_of(p[:, self.axis], 2 * c["particle_radius"])
        own = (col >= self.lo) & (col < self.hi)
        keep = own | ((col >= self.lo - self.halo) & (col < self.hi + self.halo))
        p, v, ids, own = p[keep], v[keep], ids[keep], own[keep]
        # the reference breaks ties of equal x by array position (stable lexsort, collision_detector.py:127), which in
        # the single domain is the particle id: keep the local arrays in id order (the HIP path sorts by id explicitly)
        order = np.argsort(ids, kind="stable")
        p, v, ids, own = p[order], v[order], ids[order], own[order]
        eta = None if self.noise == "none" else counter_noise_u01(ids, counter_noise_key(self.seed, self.tick))
        out = tick_core(p, v, self.segments, self.bodies, c, eta_u01=eta)
        self.p, self.v, self.ids = out["particles"][own], out["velocities"][own], ids[own]
        self.pressure = out["pressure"][own]
        self.tick += 1

    def synchronize(self):
        pass

    def owned_count(self):
        return len(self.ids)

    def download_owned(self):
        order = np.argsort(self.ids, kind="stable")
        return self.p[order], self.v[order], self.pressure[order], self.ids[order]
