"""TEST INFRASTRUCTURE.  The slice of NumPy's legacy global generator the path consumes, restated in plain Python:
MT19937 (state = NumPy's 624-word key + position, `np.random.get_state()`), `random_sample` doubles and
`binomial` (its inversion and BTPE branches, p <= 0.5) -- what `ParticleSource.generate_particles` (particle_source.py:17-24) and
`populate_colliders` (crate.py:169) draw.  The HIP generator (sand_crate_amd/csrc/sc_rng.h) is checked against
this, and this against `np.random` itself (tests/test_oracle_rng.py).

Algorithms (NumPy 1.17+ `_legacy` distributions, unchanged since): a double is (a >> 5, b >> 6) of two 32-bit
outputs, (a * 2^26 + b) / 2^53; binomial(n, p) for p <= 0.5 and n p <= 30 is sequential inversion with the
restart bound min(n, n p + 10 sqrt(n p q + 1)) and q^n taken as exp(n log q); beyond n p = 30 it is randomkit's BTPE."""
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


def binomial_btpe_setup(n: int, p: float) -> dict:
    """The constants of the BTPE branch (randomkit `rk_binomial_btpe`, which NumPy's legacy `binomial` takes for
    p <= 0.5 and n p > 30: Kachitvichyanukul & Schmeiser's triangle / parallelogram / exponential tails), host libm."""
    r = min(p, 1.0 - p)
    q = 1.0 - r
    fm = n * r + r
    m = int(math.floor(fm))
    p1 = math.floor(2.195 * math.sqrt(n * r * q) - 4.6 * q) + 0.5
    xm = m + 0.5
    xl, xr = xm - p1, xm + p1
    c = 0.134 + 20.5 / (15.3 + m)
    a = (fm - xl) / (fm - xl * r)
    laml = a * (1.0 + a / 2.0)
    a = (xr - fm) / (xr * q)
    lamr = a * (1.0 + a / 2.0)
    p2 = p1 * (1.0 + 2.0 * c)
    p3 = p2 + c / laml
    p4 = p3 + c / lamr
    return dict(r=r, q=q, m=m, p1=p1, xm=xm, xl=xl, xr=xr, c=c, laml=laml, lamr=lamr, p2=p2, p3=p3, p4=p4, nrq=n * r * q)


def binomial_btpe(rng: MT19937, n: int, p: float) -> int:
    """NumPy's legacy `binomial` for p <= 0.5 and n p > 30, draw for draw (two doubles per attempt)."""
    k_ = binomial_btpe_setup(n, p)
    r, q, m, p1, xm, xl, xr, c = (k_[x] for x in ("r", "q", "m", "p1", "xm", "xl", "xr", "c"))
    laml, lamr, p2, p3, p4, nrq = (k_[x] for x in ("laml", "lamr", "p2", "p3", "p4", "nrq"))
    while True:
        u = rng.next_double() * p4
        v = rng.next_double()
        if u <= p1:                                   # the triangle: accepted as it is
            return int(math.floor(xm - p1 * v + u))
        if u <= p2:                                   # the parallelograms
            x = xl + (u - p1) / c
            v = v * c + 1.0 - abs(m - x + 0.5) / p1
            if v > 1.0:
                continue
            y = int(math.floor(x))
        elif u <= p3:                                 # the left exponential tail
            if v == 0.0:
                continue
            y = int(math.floor(xl + math.log(v) / laml))
            if y < 0:
                continue
            v = v * (u - p2) * laml
        else:                                         # the right one
            if v == 0.0:
                continue
            y = int(math.floor(xr - math.log(v) / lamr))
            if y > n:
                continue
            v = v * (u - p3) * lamr
        k = abs(y - m)
        if k > 20 and k < nrq / 2.0 - 1:              # the squeeze, then Stirling's bound
            rho = (k / nrq) * ((k * (k / 3.0 + 0.625) + 0.16666666666666666) / nrq + 0.5)
            t = -k * k / (2 * nrq)
            big_a = math.log(v)
            if big_a < t - rho:
                return y
            if big_a > t + rho:
                continue
            x1, f1, z, w = y + 1, m + 1, n + 1 - m, n - y + 1
            x2, f2, z2, w2 = x1 * x1, f1 * f1, z * z, w * w
            if big_a > (xm * math.log(f1 / x1) + (n - m + 0.5) * math.log(z / w) + (y - m) * math.log(w * r / (x1 * q))
                        + (13680. - (462. - (132. - (99. - 140. / f2) / f2) / f2) / f2) / f1 / 166320.
                        + (13680. - (462. - (132. - (99. - 140. / z2) / z2) / z2) / z2) / z / 166320.
                        + (13680. - (462. - (132. - (99. - 140. / x2) / x2) / x2) / x2) / x1 / 166320.
                        + (13680. - (462. - (132. - (99. - 140. / w2) / w2) / w2) / w2) / w / 166320.):
                continue
            return y
        s = r / q                                     # the explicit ratio f(y) / f(m)
        a = s * (n + 1)
        f = 1.0
        if m < y:
            for i in range(m + 1, y + 1):
                f *= a / i - s
        elif m > y:
            for i in range(y + 1, m + 1):
                f /= a / i - s
        if v > f:
            continue
        return y


def binomial(rng: MT19937, n: int, p: float) -> int:
    """NumPy's legacy `binomial(n, p)` for 0 < p <= 0.5 (a particle source's p is the time step): inversion up to
    n p = 30, BTPE beyond."""
    assert 0.0 < p <= 0.5 and n > 0
    if n * p > 30.0:
        return binomial_btpe(rng, n, p)
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
