// HIP kernels of the SandCrate particle update for gfx950 (MI355X).  Included once by
// sandcrate_hip.hip.  Built with -ffp-contract=off: every decision the reference makes in
// float64 (row index, wall contact, neighbor predicate, segment crossing) is evaluated with the
// reference's operation order and without fused multiply-add, so it comes out bit-identical.
#pragma once
#include "sc_device.h"

namespace sc {

// np.clip(t, 0, 1): NaN passes through, like NumPy's minimum/maximum.
__device__ __forceinline__ double clip01(double t) { return t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t); }

// Closest point of segment s to (px,py) and the squared distance to it, evaluated exactly like
// points_to_segments_distance (geometry_utils.py:26-38): separate multiplies and adds, IEEE divide.
__device__ __forceinline__ double closest_on_segment(const Seg& s, double px, double py, double& cx, double& cy) {
  double abx = s.bx - s.ax, aby = s.by - s.ay;
  double apx = px - s.ax, apy = py - s.ay;
  double t = clip01((apx * abx + apy * aby) / (abx * abx + aby * aby));
  cx = abx * t + s.ax;
  cy = aby * t + s.ay;
  double dx = cx - px, dy = cy - py;
  return dx * dx + dy * dy;
}

// ------------------------------------------------------------------------------------------
// K0  append: crate.py:138-147 (create_new_particles).  Host arrays are P x 2 interleaved.
// ------------------------------------------------------------------------------------------
__global__ void k_append(const double* __restrict__ xy, const double* __restrict__ vxy, int m, int first_id,
                         const int* __restrict__ ids, int* __restrict__ counters, double* __restrict__ x, double* __restrict__ y,
                         double* __restrict__ vx, double* __restrict__ vy, int* __restrict__ id, int reset, int cap) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  int base = reset ? 0 : counters[C_NS];
  // (the host checks its own bound of the stored count; a slab also stores ghosts the host does not count)
  if (k < m && base + k >= cap) atomicOr(&counters[C_FLAGS], F_CAPACITY);
  if (k < m && base + k < cap) {
    x[base + k] = xy[2 * k];
    y[base + k] = xy[2 * k + 1];
    vx[base + k] = vxy[2 * k];
    vy[base + k] = vxy[2 * k + 1];
    id[base + k] = ids ? ids[k] : first_id + k;
  }
  // every block reads `base` before any block may bump the counter: the bump happens in a
  // separate one-thread launch (k_bump) ordered after this kernel on the stream.
}

__global__ void k_bump(int* counters, int m, int reset, int next_id, int cap) {
  counters[C_NS] = min((reset ? 0 : counters[C_NS]) + m, cap);
  counters[C_NEXT_ID] = next_id;  // the host's id counter after this append (an upper bound once the device emits)
}

// ------------------------------------------------------------------------------------------
// K1  wall + bin.  One thread per stored particle.
//   remove_particles            crate.py:149-159
//   calc_virtual_colliders      crate.py:213-243  (points_to_segments_distance geometry_utils.py:7-39,
//                               rigid_bodies_points_velocities crate.py:73-85 incl. its slot bug)
//   apply_hard_wall_fix         crate.py:202-211
//   row/column of the fixed position: collision_detector.py:126
// Writes the fixed position in place, the cell index, a wall-record slot, and counts the cell.
// A wall record is (sum_k u_k, sum_k vel_k, V): all that apply_pressure (:295-307) and
// apply_wall_bounce (:245-259) need later.
// ------------------------------------------------------------------------------------------
// The per-particle part of K1 on a position held in registers: returns the particle's packed cell
// or -1 (removed), the wall-record slot in `wslot`, and the position after the hard wall fix in
// (px, py).  W is World or WallInputs.  A particle in contact with a wall leaves its record at `rec` in the
// tick's record buffer -- the caller passes the particle's own storage index, so that no shared counter
// hands out slots (one returning atomic per wave on one address was the cost of a compact list).
// floor(p / d) as the reference takes it (collision_detector.py:126: an IEEE division, then floor) without the division
// wherever that is safe: p * (1/d) is within 3e-16 |q| of the quotient q, and so is the rounded division, so both have the
// quotient's floor unless the product lies within that of an integer -- only then (one particle in 10^9) is the division
// taken.  (Two float64 divisions per particle were a fifth of the vector instructions of the fused wall pass.)
__device__ __forceinline__ double floor_div(double p, double d, double inv_d) {
  const double q = p * inv_d;
  double f = floor(q);
  const double t = q - f, band = fabs(q) * 1e-12 + 1e-300;
  if (!(t > band && t < 1.0 - band)) f = floor(p / d);  // also NaN / infinity
  return f;
}

// `segs`: the segments to look at, a bit per segment, the same for every lane of the wave -- all of them, or (pass B's
// epilogue) those near the block's particles; a segment outside the mask must be farther than far_box from this particle.
template <class W>
__device__ __forceinline__ int wall_and_cell(const W& w, double& px, double& py, int& wslot, int* __restrict__ counters,
                                             int rec, double* __restrict__ wrec, unsigned segs = ~0u) {
  wslot = -1;
  if (px < w.lo || px > w.hi || py < w.lo || py > w.hi) return -1;  // crate.py:152 (dead ghosts carry x = +inf)
  bool ghost = false;
  if (w.slab) {
    long long col = (long long)floor((w.slab_axis ? py : px) / w.d);
    ghost = col < w.own_lo || col >= w.own_hi;
  }
  // bounding-box reject (exact-safe: the boxes are inflated far beyond rounding error)
  unsigned cand = 0;
  bool far = true;
  for (unsigned left = segs & (w.nseg >= 32 ? ~0u : (1u << w.nseg) - 1u); left; left &= left - 1) {
    const int k = __builtin_ctz(left);
    Seg s = w.seg[k];
    double ox = fmax(fmax(fmin(s.ax, s.bx) - px, px - fmax(s.ax, s.bx)), 0.0);
    double oy = fmax(fmax(fmin(s.ay, s.by) - py, py - fmax(s.ay, s.by)), 0.0);
    if (ox <= w.far_box && oy <= w.far_box) far = false;
    if (ox <= w.touch_box && oy <= w.touch_box) cand |= 1u << k;
  }
  wslot = far ? -1 : -2;
  if (cand) {
    // pass 1: which segments touch (geometry_utils.py:26-38 with the reference's operation order)
    unsigned touch = 0;
    double cx1 = 0, cy1 = 0;  // the contact point on the last touching segment
    for (int k = 0; k < w.nseg; ++k) {
      if (!(cand >> k & 1u)) continue;
      double cx, cy;
      if (closest_on_segment(w.seg[k], px, py, cx, cy) <= w.t_wall) {  // crate.py:229
        touch |= 1u << k;
        cx1 = cx;
        cy1 = cy;
      }
    }
    // one touching segment per lane, in every lane that touches at all (the rule away from corners): its contact point is
    // known already (one float64 division less per contact)
    const bool one_each = __ballot(touch & (touch - 1)) == 0;
    if (touch) {
      // crate.py:73-85: a body with n_b touching segments overwrites contact slots [0, n_b) -- of ALL
      // contacts, not of its own -- so slot q ends up with the velocity law of the LAST body whose
      // n_b exceeds q, evaluated at contact point q; slots beyond every n_b keep velocity 0.
      int nbv[kMaxBody];
      {
        int seg0 = 0;
        for (int b = 0; b < w.nbody; ++b) {
          int ns = w.body[b].nseg;
          unsigned mask = (ns >= 32 ? 0xFFFFFFFFu : ((1u << ns) - 1u)) << seg0;
          nbv[b] = __popc(touch & mask);
          seg0 += ns;
        }
      }
      double Ux = 0, Uy = 0, Cx = 0, Cy = 0, fx = 0, fy = 0;
      int q = 0;
      for (int k = 0; k < w.nseg; ++k) {
        if (!(touch >> k & 1u)) continue;
        double cx = cx1, cy = cy1;
        if (!one_each) closest_on_segment(w.seg[k], px, py, cx, cy);
        double ukx = (px - cx) * 2, uky = (py - cy) * 2;  // crate.py:234
        double vkx = 0.0, vky = 0.0;
        for (int b = 0; b < w.nbody; ++b) {
          if (nbv[b] > q) {
            BodyK bd = w.body[b];
            vkx = bd.vx + (cy - bd.py) * bd.omega;  // rigid_body.py:28-34
            vky = bd.vy + (-(cx - bd.px)) * bd.omega;
          }
        }
        Ux += ukx;
        Uy += uky;
        Cx += vkx;
        Cy += vky;
        double rel = w.r / sqrt(ukx * ukx + uky * uky);  // crate.py:206-208
        if (rel < 0.5) rel = 0.5;
        fx += ukx * (rel - 0.5);
        fy += uky * (rel - 0.5);
        ++q;
      }
      px += fx;  // crate.py:211
      py += fy;
      wslot = rec;
      double* out = wrec + 5 * (size_t)rec;
      out[0] = Ux;
      out[1] = Uy;
      out[2] = Cx;
      out[3] = Cy;
      out[4] = (double)q;
    }
  }
  if (!(px == px) || !(py == py)) {  // crate.py:206: distance 0 to a wall gives NaN
    atomicOr(&counters[C_FLAGS], F_NAN);
    return -1;
  }
  const double fr = floor_div(py, w.d, w.inv_d), fc = floor_div(px, w.d, w.inv_d);  // collision_detector.py:126
  // row / column in the grid, taken in float64 (integers far below 2^53: exact) -- 64-bit integer conversions and
  // compares are several instructions each; a NaN or an infinity fails the range test like any row outside the grid
  const double lr = fr - w.row0d, lc = fc - w.col0d;
  if (!(lr >= 1.0 && lr <= (double)(w.nrows - 2) && lc >= 1.0 && lc <= (double)(w.ncols - 2))) {
    if (!ghost) atomicOr(&counters[C_FLAGS], F_OUT_OF_GRID);  // a ghost beyond the local grid is simply not needed
    return -1;
  }
  return ((int)lr * w.ncols + (int)lc) | (ghost ? kGhostBit : 0);
}

// Bucket count of a wave's particles: one atomic per run of equal cells (lanes without a cell get
// distinct negative keys).  Every lane of the wave must call it.
// A wave whose particles no longer sit in their old cells (many short runs: a pile-up, where a cell of thousands
// would otherwise take thousands of atomics on one address -- they serialise at ~15 ns each and the kernel ends
// when the last one has landed) first groups its lanes by cell whatever their order, one atomic per cell, for up
// to kMatchRounds cells.
constexpr int kScrambledHeads = 28;  // more runs of equal cells than this in a wave: group the lanes by cell
constexpr int kMatchRounds = 12;
// GROUP is a launch-time choice (the host launches the grouping variants while the scans report big buckets):
// inlined into the fused epilogue of pass B the masks of the grouping loop cost that kernel eight more scalar
// registers than it has -- +2.6 us per tick in the uniform regime for a path that regime never takes.
// The same grouping over a whole WORKGROUP, in an LDS hash table of (cell, count): every particle finds or claims its
// cell's slot (linear probing; at most blockDim.x <= kCellTabSlots / 2 distinct cells) and adds itself; one global atomic
// per used slot follows.  Four times fewer atomics on a pile's counters than the waves' own groups send.
constexpr int kCellTabSlots = 512;
__device__ __forceinline__ void cell_tab_clear(int* tkey, int* tcnt) {
  for (int s = threadIdx.x; s < kCellTabSlots; s += blockDim.x) {
    tkey[s] = -1;
    tcnt[s] = 0;
  }
}
__device__ __forceinline__ int cell_tab_insert(int* tkey, int cell) {  // cell >= 0; -> its slot
  int s = (int)(((unsigned)cell * 2654435761u) >> 23);  // the product's top 9 bits
  static_assert(kCellTabSlots == 1 << 9, "the hash keeps 9 bits");
  for (;;) {
    const int prev = atomicCAS(&tkey[s], -1, cell);
    if (prev == -1 || prev == cell) return s;
    s = (s + 1) & (kCellTabSlots - 1);
  }
}

template <bool GROUP>
__device__ __forceinline__ void count_cells(int c, int* __restrict__ cellCount) {
  const int lane = threadIdx.x & 63;
  int key = c >= 0 ? (c & kCellMask) : -1 - lane;
  const LaneRun run = lane_run(key);
  if constexpr (GROUP) {
    if (__popcll(__ballot(run.is_head && c >= 0)) > kScrambledHeads) {  // wave-uniform
      unsigned long long todo = __ballot(key >= 0);
      for (int it = 0; it < kMatchRounds && todo; ++it) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll(todo) - 1);
        const int kc = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long grp = __ballot(key == kc);
        if (lane == leader) atomicAdd(&cellCount[kc], (int)__popcll(grp));
        if (key == kc) key = -1;  // counted
        todo &= ~grp;
      }
      if (key >= 0) atomicAdd(&cellCount[key], 1);  // cells beyond the rounds: one by one
      return;
    }
  }
  if (run.is_head && key >= 0) atomicAdd(&cellCount[key], run.len);
}

__global__ void __launch_bounds__(kBlock) k_wall_bin(World w, int* __restrict__ counters, double* __restrict__ x,
                                                     double* __restrict__ y, int* __restrict__ cellS,
                                                     int* __restrict__ wslotS, int* __restrict__ cellCount,
                                                     double* __restrict__ wrec, int cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int ic = min(i, cap - 1);  // the position is requested before the stored count is waited for
  double px = x[ic], py = y[ic];
  const double px0 = px, py0 = py;
  int c = -1;
  if (i < counters[C_NS]) {
    int wslot;
    c = wall_and_cell(w, px, py, wslot, counters, i, wrec);
    cellS[i] = c;
    if (c >= 0) {
      wslotS[i] = wslot;
      if (px != px0 || py != py0) {  // moved by the hard wall fix
        x[i] = px;
        y[i] = py;
      }
    }
  }
  count_cells<true>(c, cellCount);
}

// ------------------------------------------------------------------------------------------
// K2  exclusive prefix sum of the cell counts ("cell buckets").  Two launches:
//   k_scan_local: each workgroup scans 2048 counts (8 per lane: lane-serial, wave shuffle scan,
//                 4 wave totals through LDS) and writes its total;
//   k_scan_fix:   each workgroup sums the totals of the workgroups before it and adds that.
// ------------------------------------------------------------------------------------------
constexpr int kScanPerThread = 8;
constexpr int kScanPerBlock = kBlock * kScanPerThread;

__global__ void __launch_bounds__(kBlock) k_scan_local(const int* __restrict__ in, int* __restrict__ out, int n,
                                                       int* __restrict__ blockSums) {
  __shared__ int waveTot[kBlock / 64];
  int base = blockIdx.x * kScanPerBlock + threadIdx.x * kScanPerThread;
  int v[kScanPerThread];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k) {
    int e = base + k < n ? in[base + k] : 0;
    v[k] = sum;
    sum += e;
  }
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int incl = wave_scan_add(sum);
  if (lane == 63) waveTot[wv] = incl;
  __syncthreads();
  int wbase = 0;
  for (int k = 0; k < wv; ++k) wbase += waveTot[k];
  int excl = wbase + incl - sum;
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k)
    if (base + k < n) out[base + k] = excl + v[k];
  if (threadIdx.x == kBlock - 1) blockSums[blockIdx.x] = excl + sum;
}

__global__ void __launch_bounds__(kBlock) k_scan_fix(int* __restrict__ out, int n, const int* __restrict__ blockSums,
                                                     int nblocks, int* __restrict__ total_out) {
  __shared__ int waveTot[kBlock / 64];
  int acc = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) acc += blockSums[b];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) waveTot[threadIdx.x >> 6] = acc;
  __syncthreads();
  int off = 0;
  for (int k = 0; k < kBlock / 64; ++k) off += waveTot[k];
  int base = blockIdx.x * kScanPerBlock + threadIdx.x * kScanPerThread;
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k)
    if (base + k < n) out[base + k] += off;
  if ((int)blockIdx.x == nblocks - 1 && threadIdx.x == 0) {
    int tot = off + blockSums[nblocks - 1];
    out[n] = tot;  // one-past-the-end entry: cellStart[ncells]
    if (total_out) *total_out = tot;
  }
}

// Bucket starts in ONE launch and one level: a single-pass scan with decoupled look-back.  Every workgroup scans its
// 2048 counts, publishes their total (a descriptor: launch stamp, state, value in one 64-bit word, stored past the XCD's
// L2), looks back over the descriptors of the workgroups before it -- 64 at a time, down to the nearest one that already
// knows its own prefix -- publishes its prefix and writes global bucket starts.  (Round 2 kept two levels -- a start inside
// the block and block offsets that the last workgroup to draw a ticket scanned alone on the GPU: a returning atomic, a
// fence and two more round trips on the critical path of a 7.5 us kernel, and a second load in every bucket lookup.)
// The stamp changes with every launch, so the descriptors are never cleared.  A workgroup waits for lower block indices
// only, which the dispatcher starts first; the wait is bounded all the same (`max_polls`: sc_set_scan_patience).  A
// workgroup that gives up raises F_SCAN_TIMEOUT -- and with that flag up every later kernel of the tick returns at its
// first instruction (tick_abandoned): the bucket starts are not to be trusted, so the tick is SKIPPED, the storage
// arrays keep the state the tick started from, and the error reaches the caller with that state intact.
constexpr int kSortThreshold = 96;  // buckets above this many particles are listed for k_sort_big
#ifndef SC_SORT_BLOCK
#define SC_SORT_BLOCK 512
#define SC_SORT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
constexpr int kSortBlock = SC_SORT_BLOCK;
#ifndef SC_SORT_WAVES_ATTR
#define SC_SORT_WAVES_ATTR
#endif  // threads of a sorting task
#ifndef SC_SORT_CHUNK
#define SC_SORT_CHUNK 1024
#endif
constexpr int kSortChunk = SC_SORT_CHUNK;  // slots per sorting task (12 B of LDS per slot for the keys)
#ifndef SC_SORT_GRID_PER_CU
#define SC_SORT_GRID_PER_CU 4
#endif
constexpr int kSortGridPerCu = SC_SORT_GRID_PER_CU;  // k_sort_big's workgroups per CU (each takes every grid-th task)
constexpr int kSortBins = 256;         // bins of a chunk (by sampled splitters)
#ifndef SC_MAX_SORT_TASKS
#define SC_MAX_SORT_TASKS 16384
#endif
constexpr int kMaxSortTasks = SC_MAX_SORT_TASKS;   // room in k_sort_big's task list (a bucket that does not fit is ranked in K4 by counting)
static_assert(kSortChunk <= 2048, "a task packs its chunk's length - 1 into 11 bits");

// A bucket slot's sort key: the particle's x, its id (the tie-break) and its storage index, one 16-byte record -- written by
// the scatter in one store, moved by k_sort_big in one, probed by K4's searches in one load (three arrays before: a
// scrambled wave's scatter wrote 4 x 64 sectors, and a probe that hit an exact tie in x went back for the id).
struct alignas(16) SortKey {
  double x;
  int id;
  int src;
};

// A tick whose bucket scan gave up is abandoned: its later kernels do nothing (k_scan_cells).
__device__ __forceinline__ bool tick_abandoned(const int* __restrict__ counters) {
  return (__hip_atomic_load(&counters[C_FLAGS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & F_SCAN_TIMEOUT) != 0;
}

struct Buckets {
  const int* __restrict__ start;
  __device__ __forceinline__ int operator()(int c) const { return start[c]; }
};

constexpr unsigned kDescTotal = 1, kDescPrefix = 2;  // a descriptor's state: the block's own total / the total of all blocks up to it
constexpr int kScanMaxPolls = 1 << 22;
__device__ __forceinline__ unsigned long long scan_desc(unsigned stamp, unsigned state, int value) {
  return ((unsigned long long)((stamp << 2) | state) << 32) | (unsigned)value;
}

__global__ void __launch_bounds__(kBlock) k_scan_cells(const int* __restrict__ in, int* __restrict__ out, int n,
                                                       unsigned long long* __restrict__ desc, unsigned stamp,
                                                       int* __restrict__ counters, int2* __restrict__ sortTasks, int max_polls) {
  SC_TIMELINE_SCAN();
  __shared__ int waveTot[kBlock / 64], waveTasks[kBlock / 64], waveBig[kBlock / 64];
  __shared__ int blockBase, taskBase;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, blk = blockIdx.x;
  const int base = blk * kScanPerBlock + threadIdx.x * kScanPerThread;
  int v[kScanPerThread];
  int sum = 0, my_big = 0, my_tasks = 0;
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k) {
    int e = base + k < n ? in[base + k] : 0;
    v[k] = sum;
    sum += e;
    if (e > kSortThreshold) {  // rare: a bucket worth sorting properly
      ++my_big;
      my_tasks += (e + kSortChunk - 1) / kSortChunk;
    }
  }
  // Such buckets are cut into k_sort_big's tasks right here -- one task per chunk of kSortChunk slots, listed in whatever
  // order the atomics hand out (the tasks are independent).  All buckets of the workgroup's 2048 cells share ONE 64-bit
  // atomic (buckets in the low word, tasks in the high one: C_NBIG, C_NTASKS), in flight during the look-back: in the
  // pile-up regime nearly every wave has such a bucket, and an atomic per wave on that one address (11 ns each) was 6 us.
  // (The tasks used to be laid out by the last workgroup, alone on the GPU: 8 of the scan's 17 us there.)
  const bool wave_has_big = __ballot(my_big > 0) != 0;
  const int incl_t = wave_has_big ? wave_scan_add(my_tasks) : 0, wave_big = wave_has_big ? wave_scan_add(my_big) : 0;  // lane 63: the wave's
  const int incl = wave_scan_add(sum);
  if (lane == 63) {
    waveTot[wv] = incl;
    waveTasks[wv] = incl_t;
    waveBig[wv] = wave_big;
  }
  __syncthreads();
  int wbase = 0, tot = 0, tbase = 0, all_tasks = 0, all_big = 0;
  for (int k = 0; k < kBlock / 64; ++k) {
    if (k < wv) {
      wbase += waveTot[k];
      tbase += waveTasks[k];
    }
    tot += waveTot[k];
    all_tasks += waveTasks[k];
    all_big += waveBig[k];
  }
  const int excl = wbase + incl - sum;
  if (wv == 0) {  // the look-back is one wave's work
    unsigned long long got = 0;
    if (all_big > 0 && lane == 0)
      got = __hip_atomic_fetch_add((unsigned long long*)&counters[C_NBIG], (unsigned long long)all_big | ((unsigned long long)all_tasks << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int prefix = 0;
    if (blk == 0) {
      if (lane == 0) __hip_atomic_store(&desc[0], scan_desc(stamp, kDescPrefix, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&desc[blk], scan_desc(stamp, kDescTotal, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int j = blk - 1;; j -= 64) {  // wave-uniform: 64 predecessors per step, nearest first
        const int idx = j - lane;
        unsigned long long d = scan_desc(stamp, kDescPrefix, 0);  // before block 0: nothing, and known
        int polls = 0;
        bool late;
        do {
          if (idx >= 0) d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          late = (unsigned)(d >> 34) != stamp;
          if (__ballot(late) == 0) break;
          __builtin_amdgcn_s_sleep(2);
        } while (++polls < max_polls);
        if (max_polls < 0) late = true;  // (sc_set_scan_patience(-1), for tests: give up without having looked)
        if (__ballot(late)) {  // never seen: a predecessor that was not started -- give up loudly instead of hanging
          if (lane == 0) atomicOr(&counters[C_FLAGS], F_SCAN_TIMEOUT);
          break;
        }
        const unsigned long long known = __ballot(((unsigned)(d >> 32) & 3u) == kDescPrefix);
        const int stop = known ? __ffsll(known) - 1 : 63;  // the nearest predecessor that knows its prefix ends the walk
        prefix += wave_sum(lane <= stop ? (int)(unsigned)d : 0);
        if (known) break;
      }
      if (lane == 0) __hip_atomic_store(&desc[blk], scan_desc(stamp, kDescPrefix, prefix + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      blockBase = prefix;
      taskBase = (int)(got >> 32);
    }
  }
  __syncthreads();
  const int start0 = blockBase + excl;
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k)
    if (base + k <= n) out[base + k] = start0 + v[k];  // index n: one-past-the-end entry
  if (my_big > 0) {
    int first = taskBase + tbase + incl_t - my_tasks;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
      const int e = (k + 1 < kScanPerThread ? v[k + 1] : sum) - v[k];
      if (e > kSortThreshold) {
        const int nt = (e + kSortChunk - 1) / kSortChunk;
        const bool fits = first + nt <= kMaxSortTasks;
        for (int j = 0; j < nt && first + j < kMaxSortTasks; ++j)  // a bucket cut off by the list's end: no-op tasks, ranked in K4
          sortTasks[first + j] = make_int2(fits ? base + k : -1, j | ((min(kSortChunk, e - j * kSortChunk) - 1) << 20));
        first += nt;
      }
    }
  }
  if (blk == (int)gridDim.x - 1 && threadIdx.x == 0) counters[C_NT] = blockBase + tot;  // live particles = entries of the sorted arrays
}

// ------------------------------------------------------------------------------------------
// K3b  big buckets are SORTED before K4 ranks them.  Ranking a bucket of k particles by counting costs k^2
// compares; a pile of thousands of particles in one cell (stopped against a wall by the continuous-collision fix,
// many with exactly equal x) made that the longest kernel of the tick -- in the contract workload's pile-up
// regime half of all particles sit in such buckets.  Here every bucket the scan listed is cut into chunks of
// kSortChunk slots and every chunk is sorted by (x, id) by one workgroup -- a sample sort: 256 of its keys, sorted by a
// bitonic network (lane shuffles inside a wave, LDS across waves), split it into 256 bins (the keys are compared in their
// total order, so exact ties in x cannot unbalance the bins), a key finds its bin by binary search and its rank inside the
// bin by counting over the keys there -- and written back in place, the storage index travelling along: a bucket segment
// is in arrival order anyway, any permutation of it is as good.  A task's time is a chain of latencies (less than one wave
// per SIMD is resident) that grows with the keys per thread -- the bins' sizes are spread like an exponential, a wave
// counts as long as its largest bin, about five times the mean, for every key of a thread -- so the chunk is 1024 slots,
// not 2048: k_sort_big 56 -> 28 us at 1,048,576 particles in the pile-up regime (26 with the network on the DPP path),
// K4 32 -> 40 us for the extra searches.
// (A bitonic network over a whole 2048-slot chunk took 84 us; bins by x range instead of by samples 196: the piles crowd
// against their wall.)  The bucket
// is stamped; K4 then takes a particle's rank as its position inside its chunk plus, for buckets of several chunks, a
// binary search in each of the other chunks.
// Launched only when the previous tick saw big buckets; a bucket that is not stamped is ranked inside K4 by counting.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ bool key_less(double xa, int ia, double xb, int ib) { return xa < xb || (xa == xb && ia < ib); }

__global__ void __launch_bounds__(kSortBlock) SC_SORT_WAVES_ATTR
    k_sort_big(const int* __restrict__ counters, const int2* __restrict__ sortTasks,
               Buckets bk, SortKey* __restrict__ keys, int* __restrict__ sortedStamp, int stamp) {
  SC_TIMELINE_KERNEL(5);
  __shared__ double ox[kSortChunk];         // the chunk's keys in bin order
  __shared__ int oid[kSortChunk];
  __shared__ int hist[kSortBins + 1];       // bin sizes, then bin starts
  __shared__ double spx[kSortBins];  // the samples, sorted: splitters
  __shared__ int spi[kSortBins];
  __shared__ int waveTot[kSortBins / 64];
  static_assert(kSortBins == 256 && kSortBlock >= kSortBins && kSortChunk % kSortBlock == 0, "the first 256 threads hold one sample / one bin each");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tick_abandoned(counters)) return;
  const int total = min(__hip_atomic_load(&counters[C_NTASKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), kMaxSortTasks);
  for (int task = blockIdx.x; task < total; task += gridDim.x) {
    const int2 tk = sortTasks[task];  // the scan's list: (cell, chunk of its bucket | length of the chunk - 1)
    const int cell = tk.x, local = tk.y & 0xFFFFF, len = (tk.y >> 20) + 1;
    if (cell < 0) continue;  // workgroup-uniform: a bucket the list had no room for
    const int b = bk(cell) + local * kSortChunk;
    SC_STAMP_VALUE(2, 8, len);
    SC_STAMP(2, 0);
    // 1. the chunk's keys, eight per thread
    constexpr int kPerT = kSortChunk / kSortBlock;
    double x[kPerT];
    int id[kPerT], pm[kPerT], bin[kPerT], pos[kPerT];
#pragma unroll
    for (int u = 0; u < kPerT; ++u) {
      const int e = tid + u * kSortBlock;
      if (e < len) {
        const SortKey kk = keys[b + e];
        x[u] = kk.x;
        id[u] = kk.id;
        pm[u] = kk.src;
      }
    }
    // (and the thread's sample, below: every fourth key of the chunk)
    constexpr int kSampleStride = kSortChunk / kSortBins;
    const bool has_sample = tid < kSortBins && tid * kSampleStride < len;
    SortKey smp{0.0, 0, 0};
    if (has_sample) smp = keys[b + tid * kSampleStride];
    SC_STAMP(2, 1);
    // 2. splitters: 256 samples of the chunk are sorted (bitonic network, keys (x, id) in their total order -- exact ties in
    // x cannot unbalance the bins) and the first 255 of them split the chunk into 256 bins of ~4 keys
    __syncthreads();  // the previous task is done with the shared arrays
    {
      // thread t < 256 contributes key 4 t of the chunk as a sample (a chunk shorter than that has fewer samples, the rest
      // sort behind every key): samples from all over the chunk, evenly spaced.  The chunk's order is the scatter's arrival
      // order, i.e. stretches of the storage order -- of the previous tick's sorted order --, so neither a sample of its head
      // nor one of two of its quarters (rounds 2-3: keys t and 512 + t) knows the whole x range once the scatter groups a
      // workgroup's particles: k_sort_big 22 -> 37 us with bins of hundreds.
      double sx = has_sample ? smp.x : __builtin_huge_val();
      int si = has_sample ? smp.id : 0x7FFFFFFF;
      // the network: a stage whose partners sit in the same wave exchanges lane to lane -- on the DPP path when they are
      // less than 16 lanes apart (26 of the 36 stages), through the LDS crossbar for 16 and 32 (7) --, only the three
      // stages across waves go through LDS and a barrier (all 36 that way took 8 us of a 25 us task)
      auto stage = [&](int k, int j, double px, int pi) {
        const bool keep_min = ((tid & j) == 0) == ((tid & k) == 0);
        const bool partner_less = key_less(px, pi, sx, si);
        if (keep_min ? partner_less : !partner_less) {
          sx = px;
          si = pi;
        }
      };
#pragma unroll
      for (int k = 2; k <= kSortBins; k <<= 1) {
        if (k >= 256) {
          __syncthreads();
          if (tid < kSortBins) {
            spx[tid] = sx;
            spi[tid] = si;
          }
          __syncthreads();
          stage(k, 128, spx[(tid ^ 128) & (kSortBins - 1)], spi[(tid ^ 128) & (kSortBins - 1)]);
        }
        if (k >= 128) {
          __syncthreads();
          if (tid < kSortBins) {
            spx[tid] = sx;
            spi[tid] = si;
          }
          __syncthreads();
          stage(k, 64, spx[(tid ^ 64) & (kSortBins - 1)], spi[(tid ^ 64) & (kSortBins - 1)]);
        }
        if (k >= 64) stage(k, 32, xor_lane<32>(sx), xor_lane<32>(si));
        if (k >= 32) stage(k, 16, xor_lane<16>(sx), xor_lane<16>(si));
        if (k >= 16) stage(k, 8, xor_lane<8>(sx), xor_lane<8>(si));
        if (k >= 8) stage(k, 4, xor_lane<4>(sx), xor_lane<4>(si));
        if (k >= 4) stage(k, 2, xor_lane<2>(sx), xor_lane<2>(si));
        stage(k, 1, xor_lane<1>(sx), xor_lane<1>(si));
      }
      __syncthreads();
      if (tid < kSortBins) {
        spx[tid] = sx;
        spi[tid] = si;
        hist[tid] = 0;
      }
    }
    __syncthreads();
    SC_STAMP(2, 2);
    // a key's bin: the splitters (0 .. 254) below it -- a lower bound in log2(bins) fixed steps, the searches of the
    // thread's keys side by side (one after the other they were a chain of 8 x 8 dependent LDS round trips); then its
    // arrival number inside the bin
#pragma unroll
    for (int u = 0; u < kPerT; ++u) bin[u] = 0;
#pragma unroll
    for (int h = kSortBins / 2; h >= 1; h >>= 1) {
#pragma unroll
      for (int u = 0; u < kPerT; ++u) {
        const int m = bin[u] + h - 1;
        if (key_less(spx[m], spi[m], x[u], id[u])) bin[u] += h;
      }
    }
#pragma unroll
    for (int u = 0; u < kPerT; ++u)
      if (tid + u * kSortBlock < len) pos[u] = atomicAdd(&hist[bin[u]], 1);
    SC_STAMP(2, 3);
    __syncthreads();
    // 3. bin starts (exclusive scan of the sizes, one bin per thread of the first four waves)
    {
      const int v = tid < kSortBins ? hist[tid] : 0;
      const int incl = wave_scan_add(v);
      if (lane == 63 && tid < kSortBins) waveTot[wv] = incl;
      __syncthreads();
      if (tid < kSortBins) {
        int wbase = 0;
        for (int k = 0; k < wv; ++k) wbase += waveTot[k];
        hist[tid] = wbase + incl - v;
        if (tid == kSortBins - 1) hist[kSortBins] = wbase + incl;
      }
    }
    __syncthreads();
    SC_STAMP(2, 4);
    // 4. keys to their bins
#pragma unroll
    for (int u = 0; u < kPerT; ++u) {
      if (tid + u * kSortBlock < len) {
        const int p = hist[bin[u]] + pos[u];
        ox[p] = x[u];
        oid[p] = id[u];
      }
    }
    __syncthreads();
    SC_STAMP(2, 5);
    // 5. rank inside the bin by counting (a bin holds 8 keys on average; a bin of exact ties holds what it holds),
    // and the record goes to its final slot of the chunk
#pragma unroll
    for (int u = 0; u < kPerT; ++u) {
      if (tid + u * kSortBlock < len) {
        const int s0 = hist[bin[u]], s1 = hist[bin[u] + 1];
        int r = 0;
        for (int m = s0; m < s1; m += 4) {  // four entries per step, their LDS reads issued together
          double qx[4];
          int qi[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int mk = min(m + k, s1 - 1);
            qx[k] = ox[mk];
            qi[k] = oid[mk];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) r += (m + k < s1 && key_less(qx[k], qi[k], x[u], id[u])) ? 1 : 0;
        }
        keys[b + s0 + r] = SortKey{x[u], id[u], pm[u]};
      }
    }
    SC_STAMP_LARGEST_BIN();
    SC_STAMP(2, 6);
    if (local == 0 && tid == 0) sortedStamp[cell] = stamp;
  }
}

// ------------------------------------------------------------------------------------------
// K3  scatter: a slot inside the particle's cell bucket, in arrival order.  The returning
// atomic counts the bucket back down to zero, so cellCount needs no clearing for the next tick.
// The bucket slot receives the sort key and the particle's storage index (one SortKey record) and the packed cell,
// so that K4 ranks over CONTIGUOUS keys instead of chasing storage index -> x.
// ------------------------------------------------------------------------------------------
// XCD-aware block -> chunk mapping (same idea as sc_tiled.h: tile_of_block): workgroups are dealt round-robin over
// the 8 XCDs, so giving every XCD one contiguous run of chunks keeps neighboring chunks -- whose gathers and
// scattered stores fall into the same cache lines -- inside one L2.  Placement is a speed matter only.
// `live_hint`: the particles expected to be live (World::live_hint): only the blocks that hold them are dealt into
// runs, the rest of a grid sized by capacity keeps its own index.
__device__ __forceinline__ int chunk_of_block(int live_hint) {
#ifdef SC_NO_XCD_CHUNKS
  return blockIdx.x;
#else
  const int nb = min((int)gridDim.x, (int)((live_hint + blockDim.x - 1) / blockDim.x)), b = blockIdx.x;
  if (b >= nb) return b;
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  return xcd * q + min(xcd, r) + (b >> 3);
#endif
}

// The same runs walked from both ends towards the middle (like the search's tiles, sc_tiled.h): blocks start in index
// order and a kernel ends with its slowest block -- in a pile-up the blocks of the piles along floor and ceiling, the first
// of the first XCD's run and the last of the last one's.  (The scatter; K4 gains nothing from it: its slowest waves are
// the ones of the first XCD's first blocks either way.)
__device__ __forceinline__ int chunk_of_block_ends_first(int live_hint) {
#ifdef SC_NO_ENDS_FIRST
  return chunk_of_block(live_hint);
#else
  const int nb = min((int)gridDim.x, (int)((live_hint + blockDim.x - 1) / blockDim.x)), b = blockIdx.x;
  if (b >= nb) return b;
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  const int start = xcd * q + min(xcd, r), len = q + (xcd < r ? 1 : 0), l = b >> 3;
  return (l & 1) ? start + len - 1 - (l >> 1) : start + (l >> 1);
#endif
}

template <bool GROUP>
__global__ void __launch_bounds__(kBlock) k_scatter(const int* __restrict__ counters, const int* __restrict__ cellS,
                                                    const double* __restrict__ xS, const int* __restrict__ idS,
                                                    Buckets bk, int* __restrict__ cellCount,
                                                    SortKey* __restrict__ keys, int* __restrict__ keyCell, int cap,
                                                    int live_hint) {
  SC_TIMELINE_KERNEL(3);
  int i = chunk_of_block_ends_first(live_hint) * blockDim.x + threadIdx.x;  // (pile-up regime: 33.6 -> 30.0 us; uniform: the same)
  const int ic = min(i, cap - 1);  // loads that do not depend on the stored count go out first
  int c = cellS[ic];
  const int cpacked = c;
  const double xi = xS[ic];
  const int idi = idS[ic];
  SC_STAMP(4, 0);
  if (i >= counters[C_NS] || tick_abandoned(counters)) c = -1;  // (an abandoned tick scatters nothing: every lane stays idle)
  if (c >= 0) c &= kCellMask;
  SC_STAMP(4, 1);
  const int lane = threadIdx.x & 63;
  // Storage order is the previous tick's sorted order, so the lanes of a run share a cell -- as long as particles
  // stay near their cells: one returning atomic per run.  In a pile-up they do not (the contract workload's particles cross
  // more than a cell per tick by then): runs shrink to one or two lanes and a cell of thousands takes thousands of returning
  // atomics on one address.  GROUP (launched while the scans report big buckets): the WORKGROUP's particles are grouped by
  // cell in an LDS table (cell_tab_*), one returning atomic per cell and workgroup -- a quarter of what groups inside each
  // wave (rounds 2-3) sent to the hot counters: 30.3 -> 18.2 us.
  int pos = -1;
  if constexpr (GROUP) {
    __shared__ int tkey[kCellTabSlots], tcnt[kCellTabSlots], tbase[kCellTabSlots];
    cell_tab_clear(tkey, tcnt);
    __syncthreads();
    int slot = 0, rank = 0;
    if (c >= 0) {
      slot = cell_tab_insert(tkey, c);
      rank = atomicAdd(&tcnt[slot], 1);
    }
    __syncthreads();
    SC_STAMP(4, 2);
    for (int s = threadIdx.x; s < kCellTabSlots; s += blockDim.x) {
      const int k = tkey[s];
      if (k >= 0) {
        const int len = tcnt[s];
        tbase[s] = bk(k) + atomicSub(&cellCount[k], len) - len;
      }
    }
    __syncthreads();
    pos = tbase[slot] + rank;
  } else {
    const LaneRun run = lane_run(c >= 0 ? c : -1 - lane);
    int base = 0;
    if (run.is_head && c >= 0) base = bk(c) + atomicSub(&cellCount[c], run.len) - run.len;
    base = __shfl(base, run.head, 64);
    pos = base + (lane - run.head);
  }
  SC_STAMP(4, 3);
  if (c >= 0) {
    keys[pos] = SortKey{xi, idi, i};
    keyCell[pos] = cpacked;  // K4 finds the ends of a small bucket from its neighbors' cells instead of looking them up
  }
  SC_STAMP(4, 4);
}

// ------------------------------------------------------------------------------------------
// K4  reorder: final slot = bucket start + rank of (x, id) inside the bucket, which makes the
// whole array sorted by (row, x, id) = np.lexsort((x, y_floored)) with its stable tie-break
// (collision_detector.py:127).  Moves the particle's state to the sorted arrays.  Keys and ids
// of a bucket are contiguous, so even a bucket of thousands of exact-x ties (particles stopped
// on a wall by the continuous-collision fix) ranks from cached, coalesced reads.
// ------------------------------------------------------------------------------------------
constexpr int kBigBucket = 24;    // buckets above this size are ranked cooperatively
constexpr int kRankChunk = 256;   // keys streamed through LDS per step (3 KB per one-wave workgroup: the kernel is a chain
                                  // of gathers and needs the waves; with 1024 the LDS held it to 13 waves per CU, 27.6 -> 23.7 us)
constexpr int kReorderBlock = 64; // one wave per workgroup: a big bucket is shared by 4x more CUs

constexpr int kRankWindow = 12;   // slots either side of a particle in which a small bucket's ends are looked for
#ifndef SC_RANK_SIDE
#define SC_RANK_SIDE 4
#endif
constexpr int kRankSide = SC_RANK_SIDE;  // searches of a sorted bucket's other chunks that advance together (2: the same; 8: registers, 53 us)

__global__ void __launch_bounds__(kReorderBlock)
    k_reorder(const int* __restrict__ counters, const SortKey* __restrict__ keys, const int* __restrict__ keyCell, const int* __restrict__ cellS, Buckets bk,
              const int* __restrict__ wslotS, const double* __restrict__ yS, const double* __restrict__ vxS,
              const double* __restrict__ vyS, XY* __restrict__ xyT, XY* __restrict__ vvT, int* __restrict__ idT,
              int* __restrict__ cellT,
              int* __restrict__ wslotT, const int* __restrict__ sortedStamp, int stamp, int ncols,
              int* __restrict__ tileBounds, int live_hint, volatile int* __restrict__ bigHint) {
  SC_TIMELINE_KERNEL(4);
  // a hint for the host, in host-mapped memory: were there big buckets?  It is read without any synchronisation when a
  // later tick is enqueued and only decides whether k_sort_big and the grouping kernel variants are launched
  if (blockIdx.x == 0 && threadIdx.x == 0) bigHint[0] = counters[C_NBIG];
  __shared__ SortKey ck[kRankChunk];
  __shared__ int wcell[kReorderBlock + 2 * kRankWindow];
  __shared__ int pick;
  static_assert(kReorderBlock + 2 * kRankWindow <= kRankChunk && 2 * kRankWindow - 1 <= kBigBucket, "window fits, and what it resolves is a small bucket");
  const int s = chunk_of_block(live_hint) * blockDim.x + threadIdx.x;
  SC_STAMP(5, 0);
  const int nlive = counters[C_NT];
  if (s - (int)threadIdx.x >= nlive || tick_abandoned(counters)) return;  // a block beyond the live particles (a slab's grid covers its capacity)
  const bool live = s < nlive;
  int i = 0, idi = 0, cpacked = 0, c = 0, wsi = 0, b = 0, e = 0;
  double xi = 0, yi = 0, vxi = 0, vyi = 0;
  // The kernel is a chain of dependent round trips at full occupancy, so its time is the chain's length.  The bucket of
  // a particle used to cost three of them (storage index -> cell -> bucket starts -> the bucket's keys); now the scatter
  // leaves the cell next to the key and the workgroup reads the keys and cells of its 64 slots and kRankWindow slots
  // either side in one coalesced sweep: a bucket whose two ends show inside the window (nearly all: a cell holds 4
  // particles on average) is ranked from LDS while the gathers by storage index are still in flight.
  if (live) {  // the thread's own key first: the gathers by storage index hang on it
    const SortKey own = keys[s];
    xi = own.x;
    idi = own.id;
    i = own.src;
  }
  {
    const int s0 = s - (int)threadIdx.x;
    for (int w = threadIdx.x; w < kReorderBlock + 2 * kRankWindow; w += kReorderBlock) {
      const int slot = s0 - kRankWindow + w;
      const bool ok = slot >= 0 && slot < nlive;
      const int sl = ok ? slot : s0;
      const int cw = keyCell[sl];
      ck[w] = keys[sl];
      wcell[w] = ok ? cw : -1;  // packed (ghost bit and all); -1: no slot
    }
  }
  if (live) {
    yi = yS[i];
    vxi = vxS[i];
    vyi = vyS[i];
    wsi = wslotS[i];
  }
  __syncthreads();
  SC_STAMP(5, 1);
  int rank = 0;
  bool resolved = false;
  if (live) {
    const int w0 = threadIdx.x + kRankWindow;
    cpacked = wcell[w0];
    c = cpacked & kCellMask;
    // the slots of the same cell next to this one: all 24 cells of the window are read at once and the two runs are counted
    // in registers, then the bucket's keys four per step (three loops of dependent LDS reads, one per slot, were a third
    // of this kernel's wave life)
    unsigned ml = 0, mr = 0;
#pragma unroll
    for (int k = 0; k < kRankWindow; ++k) {
      const int vl = wcell[w0 - 1 - k], vr = wcell[w0 + 1 + k];
      ml |= (unsigned)(vl >= 0 && (vl & kCellMask) == c) << k;
      mr |= (unsigned)(vr >= 0 && (vr & kCellMask) == c) << k;
    }
    const int nl = __builtin_ctz(~ml), nr = __builtin_ctz(~mr);  // at most kRankWindow: bit 12 of ~m is set
    resolved = nl < kRankWindow && nr < kRankWindow;
    if (resolved) {
      b = s - nl;
      e = s + nr + 1;
      const int wl = w0 + nr;
      for (int w = w0 - nl; w <= wl; w += 4) {
        SortKey q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = ck[min(w + k, wl)];
#pragma unroll
        for (int k = 0; k < 4; ++k) rank += (w + k <= wl) && key_less(q[k].x, q[k].id, xi, idi);
      }
    } else {
      cpacked = cellS[i];  // k_sort_big moves keys and storage indices inside a bucket, not the cells next to them
      b = bk(c);
      e = bk(c + 1);
    }
  }
  SC_STAMP(5, 2);
  __syncthreads();  // the window is read; ck serves the big buckets below
  // a bucket k_sort_big has sorted this tick, chunk by chunk: the rank is the position inside the particle's chunk
  // plus the keys below (x, id) in each of the bucket's other chunks
  const bool presorted = live && !resolved && (e - b) > kSortThreshold && sortedStamp[c] == stamp;
  if (presorted) {
    const int mine = (s - b) / kSortChunk, nch = (e - b + kSortChunk - 1) / kSortChunk;
    rank = (s - b) - mine * kSortChunk;
    // the searches of up to four other chunks advance together, key and id of a probe requested at once: a step is ONE
    // round trip whatever the number of chunks and of exact ties in x (one chunk after the other, the id fetched only on a
    // tie -- the rule in a pile stopped on a wall -- a bucket of 4000 was a chain of 60 round trips: the kernel's tail)
    constexpr int kSide = kRankSide;
    for (int r0 = 0; r0 < nch; r0 += kSide) {
      int lo[kSide], hi[kSide];
#pragma unroll
      for (int u = 0; u < kSide; ++u) {
        const int r = r0 + u, cb = b + r * kSortChunk;
        lo[u] = cb;
        hi[u] = (r < nch && r != mine) ? min(cb + kSortChunk, e) : cb;  // nothing to search: an empty range
      }
      auto searching = [&]() {
        bool any = false;
#pragma unroll
        for (int u = 0; u < kSide; ++u) any |= lo[u] < hi[u];
        return any;
      };
      while (searching()) {
        double xm[kSide];
        int im[kSide], mid[kSide];
#pragma unroll
        for (int u = 0; u < kSide; ++u) {
          mid[u] = (lo[u] + hi[u]) >> 1;
          xm[u] = 0.0;
          im[u] = 0;
          if (lo[u] < hi[u]) {
            const SortKey kk = keys[mid[u]];
            xm[u] = kk.x;
            im[u] = kk.id;
          }
        }
#pragma unroll
        for (int u = 0; u < kSide; ++u) {
          if (lo[u] < hi[u]) {  // first key of the chunk that is not below (x, id)
            if (key_less(xm[u], im[u], xi, idi)) lo[u] = mid[u] + 1; else hi[u] = mid[u];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kSide; ++u) rank += lo[u] - (b + (r0 + u) * kSortChunk);
    }
  }
  SC_STAMP(5, 3);
  const bool big = live && !resolved && !presorted && (e - b) > kBigBucket;
  if (live && !resolved && !big && !presorted) {
    for (int t = b; t < e; t += 4) {  // four keys in flight per round trip
      const SortKey own{xi, idi, 0};  // a slot past the bucket's end counts as the particle itself: not below it
      const SortKey k0 = keys[t], k1 = t + 1 < e ? keys[t + 1] : own, k2 = t + 2 < e ? keys[t + 2] : own,
                    k3 = t + 3 < e ? keys[t + 3] : own;
      rank += key_less(k0.x, k0.id, xi, idi);
      rank += key_less(k1.x, k1.id, xi, idi);
      rank += key_less(k2.x, k2.id, xi, idi);
      rank += key_less(k3.x, k3.id, xi, idi);
    }
  }
  // Buckets of thousands (particles piled up against a wall, many with exactly equal x) would cost
  // bucket_size global reads per thread on the few CUs that own them.  The workgroup ranks such
  // buckets together instead: their keys stream through LDS in coalesced chunks and every thread of
  // the bucket compares against the chunk with broadcast LDS reads.  A workgroup holds consecutive
  // slots, so it sees at most a handful of distinct buckets, taken one at a time.
  SC_STAMP(5, 4);
  bool pending = big;
  while (true) {
    __syncthreads();
    if (threadIdx.x == 0) pick = -1;
    __syncthreads();
    if (pending) pick = c;  // any pending bucket will do (benign race: all writers hold valid values)
    __syncthreads();
    const int cur = pick;
    if (cur < 0) break;
    const int cb = bk(cur), ce = bk(cur + 1);
    const bool mine = pending && c == cur;
    for (int base = cb; base < ce; base += kRankChunk) {
      const int len = min(kRankChunk, ce - base);
      __syncthreads();
      for (int k = threadIdx.x; k < len; k += kReorderBlock) ck[k] = keys[base + k];
      __syncthreads();
      if (mine) {
        for (int k = 0; k < len; ++k) {
          const SortKey kk = ck[k];
          rank += (kk.x < xi) || (kk.x == xi && kk.id < idi);
        }
      }
    }
    if (mine) pending = false;
  }
  SC_STAMP(5, 5);
  if (!live) return;
  const int dst = b + rank;
  // The candidates of a block of SC_TILE_W consecutive sorted particles lie in three index ranges (sc_tiled.h);
  // the block's first and last particle know them from their cells.  Published here, one kernel ahead of the
  // tiled passes, so that those can stage their tile without waiting for bucket lookups of their own.
  if (dst % SC_TILE_W == 0) {
    int* tb = tileBounds + 6 * (dst / SC_TILE_W);
    tb[0] = bk(c - 1);
    tb[2] = bk(c + ncols - 1);
    tb[4] = bk(c - ncols - 1);
  }
  if (dst % SC_TILE_W == SC_TILE_W - 1 || dst == nlive - 1) {
    int* tb = tileBounds + 6 * (dst / SC_TILE_W);
    tb[1] = bk(c + 2);
    tb[3] = bk(c + ncols + 2);
    tb[5] = bk(c - ncols + 2);
  }
  xyT[dst] = XY{xi, yi};  // the sorted arrays hold pairs: one 16-byte store and, in the passes that stage them, one load
  vvT[dst] = XY{vxi, vyi};
  idT[dst] = idi;
  cellT[dst] = cpacked;
  wslotT[dst] = wsi;
  SC_STAMP(5, 6);
}

// K5-K7 (neighbor search, pass A, pass B) are the LDS-tiled kernels of sc_tiled.h.

// a neighbor-table row (sc_tiled.h: NbrRow) in 32-bit words; the count sits in the top five bits of the last one
constexpr int kRowWords = 8, kRowCountWord = 7, kRowCountShift = 27;
__device__ __host__ __forceinline__ int row_count_of(const unsigned int* rows, size_t i) {
  return (int)(rows[i * kRowWords + kRowCountWord] >> kRowCountShift);
}

// Sum and maximum of the neighbor counts, on demand (sc_step_stats).  Kept out of the search kernel:
// one atomic per workgroup on a single address serialises at ~12 ns each and dominated it.
__global__ void __launch_bounds__(kBlock) k_count_stats(int* __restrict__ counters, const unsigned int* __restrict__ rows,
                                                        const int* __restrict__ wslot) {
  __shared__ int ssum[kBlock / 64], smax[kBlock / 64], swall[kBlock / 64];
  int n = counters[C_NT];
  long long sum = 0;
  int mx = 0, walls = 0;
  for (int i = threadIdx.x; i < n; i += kBlock) {
    int c = row_count_of(rows, (size_t)i);  // (the count of particle i's table row: sc_tiled.h, NbrRow)
    sum += c;
    mx = max(mx, c);
    walls += wslot[i] >= 0;  // particles with a wall record (crate.py:229: V_i not empty)
  }
  // counts are <= 20, so a lane's partial sum fits 32 bits for n < 1e8
  int s32 = wave_sum((int)sum), m32 = wave_max(mx), w32 = wave_sum(walls);
  if ((threadIdx.x & 63) == 0) {
    ssum[threadIdx.x >> 6] = s32;
    smax[threadIdx.x >> 6] = m32;
    swall[threadIdx.x >> 6] = w32;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tot = 0;
    int m = 0, wl = 0;
    for (int k = 0; k < kBlock / 64; ++k) {
      tot += (unsigned)ssum[k];
      m = max(m, smax[k]);
      wl += swall[k];
    }
    counters[C_WREC] = wl;
    counters[C_SUMC] = (int)(unsigned)(tot & 0xFFFFFFFFull);
    counters[C_SUMC_HI] = (int)(tot >> 32);
    counters[C_MAXC] = m;
  }
}

// host-noise mode: exclusive scan of C_i in particle-id order gives each particle's offset into
// the host's rand(sum C_i, 2) block (crate.py:165-170 draws particle by particle in index order).
__global__ void k_count_by_id(const int* __restrict__ counters, const int* __restrict__ id,
                              const unsigned int* __restrict__ rows, int* __restrict__ cntById) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < counters[C_NT]) cntById[id[i]] = row_count_of(rows, (size_t)i);
}

// Collider offset eta_ij of crate.py:169 for slot `slot` of a particle.  `z` is the particle's
// noise_base plus slot * GOLD (counter mode), `off` its offset into the host's block (host mode).
template <int NOISE>
__device__ __forceinline__ void collider_noise(const World& w, uint64_t z, int slot, const double* __restrict__ eta,
                                               int off, double& ex, double& ey) {
  if (NOISE == SC_NOISE_NONE) {
    ex = ey = 0.0;
  } else if (NOISE == SC_NOISE_HOST) {
    const double ux = eta[2 * ((size_t)off + slot)], uy = eta[2 * ((size_t)off + slot) + 1];
    ex = (ux - 0.5) * w.d * w.level;  // crate.py:169, operation for operation
    ey = (uy - 0.5) * w.d * w.level;
  } else {
    noise_eta(z, w.eta_scale, ex, ey);
  }
}

// r = p_i - (p_j + eta) of crate.py:167-171 from the difference (dx, dy) = o_i - p_j, o_i = pair_origin(p_i).  Float-tolerance
// math: no decision is taken on it.  Counter mode: eta = (hi32 - 2^31) * eta_scale, folded into one convert and one fused
// multiply-add per component: r = ((p_i + 2^31 eta_scale) - p_j) - u32 * eta_scale, the bracket's first sum taken once per
// particle (pair_origin) instead of once per pair.
template <int NOISE>
__device__ __forceinline__ double pair_origin(const World& w, double p) {
  return NOISE == SC_NOISE_COUNTER ? p + w.eta_half : p;
}
// The hash's two 64-bit constants held in VECTOR registers (pass B's unrolled pair loop: the kernel has vector registers
// to spare -- four waves per SIMD by its LDS -- and no scalar ones, so the compiler re-materialised both constants with
// four s_mov per pair)
struct NoiseRegs {
  uint64_t gold, mix;
};
__device__ __forceinline__ NoiseRegs noise_regs() {
  uint64_t g = kGold, m = kMix;
  asm volatile("" : "+v"(g), "+v"(m));
  return NoiseRegs{g, m};
}
template <int NOISE>
__device__ __forceinline__ void pair_offset(const World& w, uint64_t z, int slot, const double* __restrict__ eta, int off,
                                            double dx, double dy, double& rx, double& ry, uint64_t mix = kMix) {
  if (NOISE == SC_NOISE_COUNTER) {
    z ^= z >> 32;
    z *= mix;
    z ^= z >> 32;
    rx = fma((double)(uint32_t)(z >> 32), -w.eta_scale, dx);
    ry = fma((double)(uint32_t)z, -w.eta_scale, dy);
  } else {
    double ex, ey;
    collider_noise<NOISE>(w, z, slot, eta, off, ex, ey);
    rx = dx - ex;
    ry = dy - ey;
  }
}

// ------------------------------------------------------------------------------------------
// Halo exchange for x-slabs (no reference counterpart).  Records are
// (x, y, vx, vy, id) as five doubles; record 0 of a buffer is a header whose first 32-bit word is the
// record count -- the very counter the packing kernel increments, so one fixed-size message per
// direction carries everything, nothing has to be published after the last record, and the host
// never needs to know the count.  The receiver's unpack re-arms the local send headers (in stream
// order the sends are over by then).
//   halo_pack_one  a particle within `halo` columns of a slab edge -- or already beyond it (a migrant) --
//                  goes to that neighbor; one atomic per wave and direction.  Called from k_halo_pack
//                  and, when the next tick's inputs are promised, from pass B's epilogue, so that the
//                  steady-state tick has no packing launch.
//   k_halo_unpack  appends the received buffers to the storage arrays; whether a record is owned or
//                  a ghost here is decided by its column in K1, not by the sender.  FUSED: K1 of the
//                  coming tick for the appended particles as well (pass B did it for the stored ones).
// ------------------------------------------------------------------------------------------
// every lane of the wave calls this; lanes that `want` get a record index in `buf`
__device__ __forceinline__ int halo_slot(bool want, double* __restrict__ buf) {
  const unsigned long long m = __ballot(want);
  if (!m) return -1;
  const int lane = threadIdx.x & 63, leader = __ffsll(m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(reinterpret_cast<int*>(buf), (int)__popcll(m));
  base = __shfl(base, leader, 64);
  return want ? base + (int)__popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// `d`, own_lo .. has_right: the slab of the COMING tick.  `on`: this lane holds a stored, live particle.
// `late`: the message has already left (halo overlap: this is an interior tile); a particle that belongs in it
// after all moved further than the band margin allows -- flagged, never silently dropped.
// -> this lane wrote a record
__device__ __forceinline__ bool halo_pack_one(bool on, double px, double py, double pvx, double pvy, int pid, double d,
                                              int axis, long long own_lo, long long own_hi, int halo, int has_left,
                                              int has_right, double* __restrict__ left, double* __restrict__ right,
                                              int cap, int* __restrict__ counters, bool late = false) {
  bool toL = false, toR = false;
  if (on && fabs(px) < 1e300) {  // not a dead ghost copy (x = +inf), not NaN
    const long long col = (long long)floor((axis ? py : px) / d);
    toL = has_left && col < own_lo + halo;
    toR = has_right && col >= own_hi - halo;
  }
  if (late) {
    if (toL || toR) atomicOr(&counters[C_FLAGS], F_HALO_LATE);
    return false;
  }
  const int kl = halo_slot(toL, left), kr = halo_slot(toR, right);
  if (kl >= cap || kr >= cap) atomicOr(&counters[C_FLAGS], F_HALO_OVERFLOW);
  if (kl >= 0 && kl < cap) {
    double* r = left + (size_t)kHaloFields * (kl + 1);
    r[0] = px; r[1] = py; r[2] = pvx; r[3] = pvy; r[4] = (double)pid;
  }
  if (kr >= 0 && kr < cap) {
    double* r = right + (size_t)kHaloFields * (kr + 1);
    r[0] = px; r[1] = py; r[2] = pvx; r[3] = pvy; r[4] = (double)pid;
  }
  return toL || toR;
}

__global__ void __launch_bounds__(kBlock)
    k_halo_pack(World w, int* __restrict__ counters, const double* __restrict__ x, const double* __restrict__ y,
                const double* __restrict__ vx, const double* __restrict__ vy, const int* __restrict__ id,
                double* __restrict__ left, double* __restrict__ right, int cap, int capS) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int ic = min(i, capS - 1);
  const bool on = i < counters[C_NS];
  halo_pack_one(on, x[ic], y[ic], vx[ic], vy[ic], id[ic], w.d, w.slab_axis, w.own_lo, w.own_hi, w.halo, w.has_left, w.has_right, left,
                right, cap, counters);
}

// Appends the received buffers (either may be null) to the storage arrays and re-arms the local
// send headers.  Plain: every workgroup reads the old stored count first, the last one to finish
// (ticket) adds the two record counts to it.  FUSED (pass B of the previous tick ran K1 for the stored
// particles): the stored count is the sorted count of that tick, nobody changes it, and the appended
// particles get their K1 here.
// `capL` / `capR`: records the transport actually moved from the left / right (the agreed message sizes of
// sc_halo_sizes).  A header that announces more means records were cut off: F_HALO_OVERFLOW.  `ring`: the four
// counts of this tick (sent left, sent right, received from the left, from the right) are published in
// host-mapped memory, slot tick % kHaloRing, for the message sizes of a later tick.
constexpr int kHaloRing = 8;
constexpr int kProgressInts = 4 + 4 * kHaloRing;  // [big buckets, ticks finished, live count, -] + the ring

template <bool FUSED>
__global__ void __launch_bounds__(kBlock)
    k_halo_unpack(const double* __restrict__ bufL, const double* __restrict__ bufR, int capL, int capR,
                  int* __restrict__ counters, double* __restrict__ x, double* __restrict__ y, double* __restrict__ vx,
                  double* __restrict__ vy, int* __restrict__ id, int capS, double* __restrict__ sendL,
                  double* __restrict__ sendR, WallInputs wn, int* __restrict__ cellS, int* __restrict__ wslotS,
                  int* __restrict__ cellCount, double* __restrict__ wrec_next, volatile int* __restrict__ ring) {
  const int hl = bufL ? *reinterpret_cast<const int*>(bufL) : 0, hr = bufR ? *reinterpret_cast<const int*>(bufR) : 0;
  const int nl = min(hl, capL), nr = min(hr, capR);
  const int base = FUSED ? counters[C_NT] : __hip_atomic_load(&counters[C_NS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  int cnext = -1;
  if (k < nl + nr) {
    const double* r = k < nl ? bufL + (size_t)kHaloFields * (k + 1) : bufR + (size_t)kHaloFields * (k - nl + 1);
    if (base + k >= capS) {
      atomicOr(&counters[C_FLAGS], F_CAPACITY);
    } else {
      double px = r[0], py = r[1];
      if (FUSED) {
        int wsn = -1;
        cnext = wall_and_cell(wn, px, py, wsn, counters, base + k, wrec_next);
        cellS[base + k] = cnext;
        if (cnext >= 0) wslotS[base + k] = wsn;
      }
      x[base + k] = px;
      y[base + k] = py;
      vx[base + k] = r[2];
      vy[base + k] = r[3];
      id[base + k] = (int)r[4];
    }
  }
  auto finish = [&]() {  // one thread, after every workgroup has read what it needs
    counters[C_NS] = min(base + nl + nr, capS);
    if (hl > capL || hr > capR) atomicOr(&counters[C_FLAGS], F_HALO_OVERFLOW);
    ring[0] = sendL ? *reinterpret_cast<const int*>(sendL) : 0;
    ring[1] = sendR ? *reinterpret_cast<const int*>(sendR) : 0;
    ring[2] = hl;
    ring[3] = hr;
    if (sendL) *reinterpret_cast<int*>(sendL) = 0;
    if (sendR) *reinterpret_cast<int*>(sendR) = 0;
  };
  if (FUSED) {
    count_cells<true>(cnext, cellCount);  // every lane of the wave takes part
    if (blockIdx.x == 0 && threadIdx.x == 0) finish();
    return;
  }
  __syncthreads();  // every thread of this workgroup has read `base`
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&counters[C_TICKET], 1) == (int)gridDim.x - 1) {
      finish();
      counters[C_TICKET] = 0;
    }
  }
}

// Stored live particles per grid column, clamped into [col0, col0 + ncols): across the ranks every particle is
// stored live exactly once, so the sum of the ranks' histograms is the global one (slab re-balancing).
__global__ void __launch_bounds__(kBlock)
    k_column_histogram(const int* __restrict__ counters, const double* __restrict__ x, const double* __restrict__ coord,
                       double d, long long col0, int ncols, int* __restrict__ hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counters[C_NS]) return;
  if (!(fabs(x[i]) < 1e300)) return;  // a dead ghost copy: its owner counts the particle
  const long long col = (long long)floor(coord[i] / d);  // coord: x (slabs of columns) or y (slabs of rows)
  const long long k = col - col0;
  atomicAdd(&hist[k < 0 ? 0 : (k >= ncols ? ncols - 1 : (int)k)], 1);
}

// live particles in the storage arrays: everything but the dead ghost copies (x = +inf) a slab tick
// leaves behind.  Across ranks every particle is stored live exactly once.
__global__ void __launch_bounds__(kBlock) k_owned_count(const int* __restrict__ counters, const double* __restrict__ x,
                                                        int* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int own = 0;
  if (i < counters[C_NS]) {
    own = fabs(x[i]) < 1e300;
  }
  int s = wave_sum(own);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// geometry_utils.py:7-39 as a stand-alone kernel (the reference's tests pin it): one thread per
// (point, segment).
__global__ void k_points_to_segments(const double* __restrict__ xy, int n, const double* __restrict__ seg, int ns,
                                     double* __restrict__ nearest, double* __restrict__ dist) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * ns) return;
  int p = (int)(t / ns), k = (int)(t % ns);
  double px = xy[2 * p], py = xy[2 * p + 1];
  double ax = seg[4 * k], ay = seg[4 * k + 1], bx = seg[4 * k + 2], by = seg[4 * k + 3];
  double abx = bx - ax, aby = by - ay;
  double apx = px - ax, apy = py - ay;
  double tt = (apx * abx + apy * aby) / (abx * abx + aby * aby);
  tt = clip01(tt);
  double cx = abx * tt + ax, cy = aby * tt + ay;
  double dx = cx - px, dy = cy - py;
  nearest[2 * t] = cx;
  nearest[2 * t + 1] = cy;
  dist[t] = sqrt(dx * dx + dy * dy);
}

}  // namespace sc
