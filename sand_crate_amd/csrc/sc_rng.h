// NumPy's legacy global generator on the device: MT19937 with NumPy's state layout
// (624-word key + position), `random_sample` doubles, and what the tick draws from it --
//   k_rng_emit   ParticleSource.generate_particles (particle_source.py:17-24) for every active source: the legacy
//                binomial (inversion up to flow dt = 30, BTPE beyond), then rand(n, 2) for the position jitter and rand(n, 2)
//                for the velocity noise, appended to the storage arrays (crate.py:138-147);
//   k_rng_noise  the tick's collider noise (crate.py:169): one block rand(sum C_i, 2), which is what the reference
//                draws particle by particle (`rand(a, 2)` then `rand(b, 2)` is `rand(a + b, 2)` split).
// The stream is NumPy's bit for bit (oracle/rng.py restates it and is checked against np.random), so the default
// noise="host" mode of `Crate` needs no per-tick count readback, host draw and upload any more.  Included once by
// sandcrate_hip.hip after sc_kernels.h.  Built with -ffp-contract=off like everything else: the position and
// velocity arithmetic is NumPy's, operation for operation.
#pragma once
#include "sc_kernels.h"

namespace sc {

constexpr int kMtN = 624, kMtM = 397;
constexpr uint32_t kMtUpper = 0x80000000u, kMtLower = 0x7FFFFFFFu, kMtA = 0x9908B0DFu;

struct RngState {
  uint32_t mt[kMtN];
  int pos;  // next unused word of mt; kMtN: the block is used up
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9D2C5680u;
  y ^= (y << 15) & 0xEFC60000u;
  y ^= y >> 18;
  return y;
}

__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t next, uint32_t far) {
  const uint32_t y = (cur & kMtUpper) | (next & kMtLower);
  return far ^ (y >> 1) ^ ((y & 1u) ? kMtA : 0u);
}

__device__ __forceinline__ double mt_double(uint32_t a, uint32_t b) {  // NumPy: (a >> 5, b >> 6)
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// one thread, state staged in LDS: the few dozen draws of the particle sources
struct RngSerial {
  uint32_t* mt;  // 624 words in LDS
  int pos;
  bool refilled;
  __device__ uint32_t next_u32() {
    if (pos >= kMtN) {
      for (int kk = 0; kk < kMtN; ++kk) mt[kk] = mt_twist(mt[kk], mt[(kk + 1) % kMtN], mt[(kk + kMtM) % kMtN]);
      pos = 0;
      refilled = true;
    }
    return mt_temper(mt[pos++]);
  }
  __device__ double next_double() {
    const uint32_t a = next_u32(), b = next_u32();
    return mt_double(a, b);
  }
};

struct SourceK {
  double radius, px, py, vx, vy, noise;
  double p, q, qn;          // binomial(flow, p): p = dt, q = 1 - p, qn = q^flow as exp(flow log q) (host libm, like NumPy)
  long long flow, bound;    // restart bound min(flow, flow p + 10 sqrt(flow p q + 1))
  // flow p > 30: the BTPE branch (randomkit rk_binomial_btpe) and its constants, taken on the host like NumPy takes them
  int btpe;
  long long m;
  double p1, xm, xl, xr, c, laml, lamr, p2, p3, p4, nrq;
};

// NumPy's legacy binomial for p <= 0.5 and n p > 30 (Kachitvichyanukul & Schmeiser's BTPE as randomkit has it; oracle/rng.py:
// binomial_btpe is the same, pinned against np.random): two doubles per attempt -- a point under the triangle is accepted as
// it is, the parallelograms and the exponential tails go through the squeeze and, near the mode, the explicit ratio.
// (The logarithms are the device library's: a result decided differently from glibc's needs a draw within an ulp of an
// integer boundary.)
template <class Rng>
__device__ long long binomial_btpe(Rng& rng, const SourceK& s) {
  const long long n = s.flow, m = s.m;
  const double r = s.p, q = s.q;
  for (;;) {
    const double u = rng.next_double() * s.p4;
    double v = rng.next_double();
    long long y;
    if (u <= s.p1) return (long long)floor(s.xm - s.p1 * v + u);
    if (u <= s.p2) {
      const double x = s.xl + (u - s.p1) / s.c;
      v = v * s.c + 1.0 - fabs((double)m - x + 0.5) / s.p1;
      if (v > 1.0) continue;
      y = (long long)floor(x);
    } else if (u <= s.p3) {
      if (v == 0.0) continue;
      const double yy = floor(s.xl + log(v) / s.laml);
      if (yy < 0.0) continue;
      y = (long long)yy;
      v = v * (u - s.p2) * s.laml;
    } else {
      if (v == 0.0) continue;
      const double yy = floor(s.xr - log(v) / s.lamr);
      if (yy > (double)n) continue;
      y = (long long)yy;
      v = v * (u - s.p3) * s.lamr;
    }
    const long long k = y > m ? y - m : m - y;
    if (k > 20 && (double)k < s.nrq / 2.0 - 1) {
      const double kd = (double)k;
      const double rho = (kd / s.nrq) * ((kd * (kd / 3.0 + 0.625) + 0.16666666666666666) / s.nrq + 0.5);
      const double t = -kd * kd / (2 * s.nrq);
      const double A = log(v);
      if (A < t - rho) return y;
      if (A > t + rho) continue;
      const double x1 = (double)(y + 1), f1 = (double)(m + 1), z = (double)(n + 1 - m), w = (double)(n - y + 1);
      const double x2 = x1 * x1, f2 = f1 * f1, z2 = z * z, w2 = w * w;
      if (A > (s.xm * log(f1 / x1) + ((double)(n - m) + 0.5) * log(z / w) + (double)(y - m) * log(w * r / (x1 * q)) +
               (13680. - (462. - (132. - (99. - 140. / f2) / f2) / f2) / f2) / f1 / 166320. +
               (13680. - (462. - (132. - (99. - 140. / z2) / z2) / z2) / z2) / z / 166320. +
               (13680. - (462. - (132. - (99. - 140. / x2) / x2) / x2) / x2) / x1 / 166320. +
               (13680. - (462. - (132. - (99. - 140. / w2) / w2) / w2) / w2) / w / 166320.))
        continue;
      return y;
    }
    const double sq = r / q, a = sq * (double)(n + 1);
    double F = 1.0;
    if (m < y) {
      for (long long i = m + 1; i <= y; ++i) F *= (a / (double)i - sq);
    } else if (m > y) {
      for (long long i = y + 1; i <= m; ++i) F /= (a / (double)i - sq);
    }
    if (v > F) continue;
    return y;
  }
}
constexpr int kMaxSources = 8;
struct SourcesK {
  SourceK src[kMaxSources];
  int n;
};

// crate.py:138-147 + particle_source.py:17-24.  One thread: a tick emits a handful of particles.
__global__ void k_rng_emit(SourcesK srcs, long long max_particles, RngState* __restrict__ state, int* __restrict__ counters,
                           double* __restrict__ x, double* __restrict__ y, double* __restrict__ vx,
                           double* __restrict__ vy, int* __restrict__ id, int cap) {
  __shared__ uint32_t mt[kMtN];
  for (int k = threadIdx.x; k < kMtN; k += blockDim.x) mt[k] = state->mt[k];  // one coalesced read instead of a
  __syncthreads();                                                             // dependent global load per draw
  if (threadIdx.x != 0) return;
  RngSerial rng{mt, state->pos, false};
  int stored = counters[C_NS];
  int next_id = counters[C_NEXT_ID];
  for (int k = 0; k < srcs.n; ++k) {
    const SourceK s = srcs.src[k];
    long long X = 0;
    if (s.btpe) {  // legacy random_binomial_btpe (flow p > 30)
      X = binomial_btpe(rng, s);
      if (X > s.bound) {  // (ten standard deviations out: the host sized its bounds for less; say so rather than overrun them)
        atomicOr(&counters[C_FLAGS], F_CAPACITY);
        X = s.bound;
      }
    } else {  // legacy random_binomial_inversion
      double px = s.qn, U = rng.next_double();
      while (U > px) {
        ++X;
        if (X > s.bound) {
          X = 0;
          px = s.qn;
          U = rng.next_double();
        } else {
          U -= px;
          px = ((double)(s.flow - X + 1) * s.p * px) / ((double)X * s.q);
        }
      }
    }
    long long count = X;  // np.round of an integer
    const long long room = max_particles - stored;  // crate.py:142: the count the previous source left
    if (count > room) count = room;
    if (count <= 0) continue;  // particle_source.py:19-20 (a negative room draws nothing more either)
    if (stored + count > cap) {
      atomicOr(&counters[C_FLAGS], F_CAPACITY);
      count = cap - stored;
    }
    for (long long j = 0; j < count; ++j) {  // jitter = rand(count, 2); (jitter - 0.5) * radius + position
      const double jx = rng.next_double(), jy = rng.next_double();
      x[stored + j] = (jx - 0.5) * s.radius + s.px;
      y[stored + j] = (jy - 0.5) * s.radius + s.py;
    }
    for (long long j = 0; j < count; ++j) {  // ones * velocity += (rand(count, 2) - 0.5) * noise
      const double nx = rng.next_double(), ny = rng.next_double();
      vx[stored + j] = s.vx + (nx - 0.5) * s.noise;
      vy[stored + j] = s.vy + (ny - 0.5) * s.noise;
      id[stored + j] = next_id + (int)j;
    }
    stored += (int)count;
    next_id += (int)count;
  }
  counters[C_NS] = stored;
  counters[C_NEXT_ID] = next_id;
  if (rng.refilled)
    for (int k = 0; k < kMtN; ++k) state->mt[k] = mt[k];
  state->pos = rng.pos;
}

// The collider noise of one tick: 2 * pairs doubles from the stream into eta, in order.  FOUR WAVES: the recurrence
// x[n + 624] = f(x[n], x[n + 1], x[n + 397]) reaches 227 words back, so a state block of 624 words is regenerated in three
// dependent phases of at most 227 independent words -- a word per thread, a workgroup barrier between the reads and the
// writes of a phase (thread k reads what thread k + 1 overwrites) and one behind the writes --; the block is then
// tempered and turned into doubles by all 256 threads; a double whose two words straddle a block boundary is finished
// with the carried word.  (Measured on the viewer's scene, 77 blocks per tick: 1,024 threads -- sixteen waves at every
// barrier -- 100 us; ONE wave without any barrier 150 us: a lone wave issues an instruction every 5-9 clocks and the
// tempering is ~35 instructions per double; four waves: see DESIGN.md.)
constexpr int kRngBlock = 256;
// (the first kRngBlock threads of a workgroup call this -- all of them --; `mt`: 624 words of LDS; `bar`: a barrier
// those threads, and only they, meet: the workgroup's own for a workgroup of kRngBlock threads)
// A workgroup barrier that orders LDS accesses only: __syncthreads() also waits for the thread's outstanding GLOBAL stores
// (s_waitcnt vmcnt(0)), and the loop below stores a block's doubles to global memory in every round -- each barrier then
// cost a store's round trip to memory.
struct LdsBarrier {
  __device__ __forceinline__ void operator()() const { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
};

template <class Barrier>
__device__ __forceinline__ void rng_noise_block(uint32_t* mt2, RngState* __restrict__ state, long long pairs, double* __restrict__ eta,
                                                long long eta_pairs_room, int* __restrict__ counters, Barrier bar) {
  // `mt2`: TWO state blocks of LDS.  A regeneration reads the current block and writes the other one, so a phase needs
  // no barrier between its reads and its writes -- three barriers per block, one behind each phase's writes (nine with
  // one block in place; the kernel is a chain of barriers and LDS round trips, ~0.1 us each at the clocks a one-workgroup
  // kernel runs at).
  uint32_t* cur = mt2;
  uint32_t* nxt = mt2 + kMtN;
  const int tid = threadIdx.x;
  for (int k = tid; k < kMtN; k += kRngBlock) cur[k] = state->mt[k];
  int pos = state->pos;
  if (pairs > eta_pairs_room) {
    if (tid == 0) atomicOr(&counters[C_FLAGS], F_CAPACITY);
    pairs = eta_pairs_room;
  }
  const long long need = 2 * pairs;  // doubles
  long long w = 0;
  bool have_carry = false;
  uint32_t carry = 0;
  constexpr int kSpan = kMtN - kMtM;  // 227: the words of a phase
  static_assert(kSpan <= kRngBlock, "a phase is one word per thread");
  bar();
  while (w < need) {  // every variable that steers this loop is uniform over the threads
    if (pos >= kMtN) {
      // word kk of the new block from words kk, kk + 1 of the current one and word kk + 397 -- of the current block up to
      // kk = 226, of the NEW block (kk - 227) beyond: hence the phases [0, 227), [227, 454), [454, 624); the last word
      // takes the new word 0 as its successor
#pragma unroll
      for (int phase = 0; phase < 3; ++phase) {
        const int kk = phase * kSpan + tid;
        if (tid < kSpan && kk < kMtN) {
          const uint32_t succ = kk + 1 < kMtN ? cur[kk + 1] : nxt[0];
          const uint32_t far = kk + kMtM < kMtN ? cur[kk + kMtM] : nxt[kk + kMtM - kMtN];
          nxt[kk] = mt_twist(cur[kk], succ, far);
        }
        bar();
      }
      uint32_t* t = cur;
      cur = nxt;
      nxt = t;
      pos = 0;
    }
    int start = pos;
    if (have_carry) {
      if (tid == 0) eta[w] = mt_double(carry, mt_temper(cur[start]));
      w += 1;
      start += 1;
      have_carry = false;
    }
    const long long left = need - w;
    const int nd = (int)min((long long)((kMtN - start) / 2), left);
    for (int k = tid; k < nd; k += kRngBlock)
      eta[w + k] = mt_double(mt_temper(cur[start + 2 * k]), mt_temper(cur[start + 2 * k + 1]));
    w += nd;
    start += 2 * nd;
    if (w < need && start == kMtN - 1) {
      carry = mt_temper(cur[kMtN - 1]);
      have_carry = true;
      start = kMtN;
    }
    pos = start;
  }
  bar();
  for (int k = tid; k < kMtN; k += kRngBlock) state->mt[k] = cur[k];
  if (tid == 0) state->pos = pos;
}

__global__ void __launch_bounds__(kRngBlock)
    k_rng_noise(RngState* __restrict__ state, const int* __restrict__ pairs_ptr, double* __restrict__ eta,
                long long eta_pairs_room, int* __restrict__ counters) {
  __shared__ uint32_t mt[2 * kMtN];
  rng_noise_block(mt, state, *pairs_ptr, eta, eta_pairs_room, counters, LdsBarrier{});
}

// A small world's noise in ONE launch (the viewer's scenes hold a few thousand particles: their tick is a dozen launches of
// a few microseconds each, and five of them were these): the neighbor counts by particle id (zero, scatter), their
// exclusive scan -- the offsets into the tick's rand(sum C_i, 2) block, crate.py:165-170 draws particle by particle in id
// order -- and the block itself, by one workgroup.  `ids`: the host's bound of the ids handed out (at most kSmallIds).
constexpr int kSmallIds = 1 << 16;
constexpr int kSmallBlock = kRngBlock;  // (the scan of the ids and the stream share the workgroup)
__global__ void __launch_bounds__(kSmallBlock)
    k_rng_noise_small(RngState* __restrict__ state, const int* __restrict__ id, const unsigned int* __restrict__ rows, int ids,
                      int* __restrict__ cntById, int* __restrict__ offById, double* __restrict__ eta, long long eta_pairs_room,
                      int* __restrict__ counters) {
  __shared__ uint32_t mt[2 * kMtN];
  __shared__ int waveTot[kSmallBlock / 64];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = counters[C_NT];
  for (int k = tid; k < ids; k += kSmallBlock) cntById[k] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += kSmallBlock) cntById[id[i]] = row_count_of(rows, (size_t)i);
  if (tid == 0) carry_s = 0;
  __syncthreads();
  // exclusive scan over the ids, every thread a run of consecutive ids: its loads are all in flight together (one
  // round trip -- an id per thread and step was a chain of dependent global loads, ~2 us each), one scan of the
  // threads' totals across the workgroup, then the run is walked again for the offsets
  {
    const int per = (ids + kSmallBlock - 1) / kSmallBlock, k0 = tid * per, k1 = min(k0 + per, ids);
    int sum = 0;
    for (int k = k0; k < k1; ++k) sum += cntById[k];
    const int incl = wave_scan_add(sum);
    if (lane == 63) waveTot[wv] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < kSmallBlock / 64; ++w) {
      if (w < wv) before += waveTot[w];
      total += waveTot[w];
    }
    int run = before + incl - sum;
    for (int k = k0; k < k1; ++k) {
      offById[k] = run;
      run += cntById[k];
    }
    if (tid == 0) carry_s = total;
    __syncthreads();
  }
  const long long pairs = carry_s;
  if (tid == 0) offById[ids] = (int)pairs;
  rng_noise_block(mt, state, pairs, eta, eta_pairs_room, counters, LdsBarrier{});
}

}  // namespace sc
