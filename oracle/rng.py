"""TEST INFRASTRUCTURE.  The slice of NumPy's legacy global generator the path consumes, restated in plain Python:
MT19937 (state = NumPy's 624-word key + position, `np.random.get_state()`), `random_sample` doubles and the
inversion branch of `binomial` -- what `ParticleSource.generate_particles` (particle_source.py:17-24) and
`populate_colliders` (crate.py:169) draw.  The HIP generator (sand_crate_amd/csrc/sc_rng.h) is checked against
this, and this against `np.random` itself (tests/test_oracle_rng.py).

Algorithms (NumPy 1.17+ `_legacy` distributions, unchanged since): a double is (a >> 5, b >> 6) of two 32-bit
outputs, (a * 2^26 + b) / 2^53; binomial(n, p) for p <= 0.5 and n p <= 30 is sequential inversion with the
restart bound min(n, n p + 10 sqrt(n p q + 1)) and q^n taken as exp(n log q)."""
from __future__ import annotations

import math

import numpy as np

N, M = 624, 397
_UPPER, _LOWER, _A = 0x80000000, 0x7FFFFFFF, 0x9908B0DF


class MT19937:
    def __init__(self, key, pos: int):
        self.mt = [int(k) for k in key]
        self.pos = int(pos)

    @classmethod
    def from_numpy(cls):
        name, key, pos, _, _ = np.random.get_state()
        assert name == "MT19937"
        return cls(key, pos)

    def to_numpy(self) -> None:
        np.random.set_state(("MT19937", np.array(self.mt, dtype=np.uint32), self.pos, 0, 0.0))

    def _refill(self) -> None:
        mt = self.mt
        for kk in range(N):
            y = (mt[kk] & _UPPER) | (mt[(kk + 1) % N] & _LOWER)
            mt[kk] = mt[(kk + M) % N] ^ (y >> 1) ^ (_A if y & 1 else 0)
        self.pos = 0

    def next_u32(self) -> int:
        if self.pos >= N:
            self._refill()
        y = self.mt[self.pos]
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def next_double(self) -> float:
        a, b = self.next_u32() >> 5, self.next_u32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    def rand(self, *shape) -> np.ndarray:
        out = np.array([self.next_double() for _ in range(int(np.prod(shape)))])
        return out.reshape(shape)


def binomial_setup(n: int, p: float):
    """-> (q, q^n, restart bound): the constants the inversion loop needs (host libm, as NumPy computes them)."""
    q = 1.0 - p
    qn = math.exp(n * math.log(q))
    npq = n * p
    bound = int(min(n, npq + 10.0 * math.sqrt(npq * q + 1)))
    return q, qn, bound


def binomial_inversion_ok(n: int, p: float) -> bool:
    return 0.0 < p <= 0.5 and n * p <= 30.0 and n > 0


def binomial(rng: MT19937, n: int, p: float) -> int:
    assert binomial_inversion_ok(n, p)
    q, qn, bound = binomial_setup(n, p)
    x, px, u = 0, qn, rng.next_double()
    while u > px:
        x += 1
        if x > bound:
            x, px, u = 0, qn, rng.next_double()
        else:
            u -= px
            px = ((n - x + 1) * p * px) / (x * q)
    return x


def generate_particles(rng: MT19937, source, dt: float, room: int):
    """particle_source.py:17-24 on the restated stream: -> (positions, velocities) or (None, None)."""
    count = min(binomial(rng, int(source.flow), dt), room)
    if count == 0:
        return None, None
    jitter = rng.rand(count, 2)
    positions = (jitter - 0.5) * source.radius + np.array(source.position)
    velocities = np.ones_like(positions) * np.array(source.velocity)[None]
    velocities += (rng.rand(count, 2) - 0.5) * source.noise
    return positions, velocities
