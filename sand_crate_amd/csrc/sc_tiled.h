// LDS-tiled pass A (neighbor search + density + surface normals) and pass B (forces + integration)
// for gfx950.  Included once by sandcrate_hip.hip after sc_kernels.h.
//
// Geometry.  The sorted arrays are in (row, x, id) order and cells are numbered row-major with a
// ring of empty cells, so for a block of kTileW consecutive particles (cells c_first..c_last) the
// candidates of ALL its particles lie in three contiguous index ranges:
//     same rows      [cellStart[c_first - 1],         cellStart[c_last + 2])
//     next rows      [cellStart[c_first + ncols - 1], cellStart[c_last + ncols + 2])
//     previous rows  [cellStart[c_first - ncols - 1], cellStart[c_last - ncols + 2])
// (about 3 x (kTileW + 12) particles at 3.8 particles per cell).  A workgroup copies these ranges
// into LDS once with coalesced loads; every later read of a neighbor is a ds_read.  Positions in
// the concatenation of the three ranges are "tile slots"; the neighbor table stores tile slots, so
// pass B, which stages the same three ranges (pass A publishes them per block), indexes its tile
// directly.  A tile that does not fit the LDS budget (a block inside a very dense region) is read
// from global memory by a second instantiation of the same body behind one workgroup-uniform branch
// -- two instantiations rather than one body with a run-time flag, because a run-time choice
// between an LDS and a global address compiles to flat loads.
//
// What bounds these kernels (rocprofv3 counters, profiles/): with 262,144 particles every kernel
// starts with a cold L2 (multi-XCD coherence at kernel boundaries) and a wave's time is a chain of
// dependent cold misses, ~2 us each, plus fp64 VALU issue (wave64 fp64 instructions take 4 cycles);
// from ~1M particles on it is fp64 VALU issue alone.  Hence: loads that do not depend on each other
// are issued together ahead of the first wait, one thread scans its candidates serially from LDS
// (about 12 instructions per candidate), and the pair math uses v_rsq_f64 + Newton steps instead of
// an IEEE sqrt and divide.
#pragma once
#include <climits>
#include <type_traits>

#include "sc_kernels.h"

namespace sc {

// Tile geometry (measured, profiles/README.md): 256 particles per workgroup beat 128 by 2-3 % (less halo per
// particle: 3 x (256 + 12) entries for 256 particles) and 64 lose 3-6 %.
#ifndef SC_SCAN_BATCH
#define SC_SCAN_BATCH 4
#endif
#ifndef SC_COOP
#define SC_COOP 2
#endif
#ifndef SC_LEAD_IN
#define SC_LEAD_IN 1
#endif
#ifndef SC_COOP_FAST
#define SC_COOP_FAST 1
#endif
#ifndef SC_B_LEAN_LOOP
#define SC_B_LEAN_LOOP 1
#endif
#ifndef SC_SERIAL_ONCE
#define SC_SERIAL_ONCE 1
#endif
#ifndef SC_CAP_A
#define SC_CAP_A 1024
#define SC_CAP_AW 1536
#define SC_CAP_B 960
#endif
constexpr int kTileW = SC_TILE_W;      // particles (= threads) per workgroup
// pass A tile, (x, y) records: 16 KiB when many workgroups share a CU (more waves in flight), 24 KiB when the
// whole grid is resident anyway (fewer tiles fall out of LDS); the launcher picks (measured: profiles/)
constexpr int kTileCapA = SC_CAP_A, kTileCapAWide = SC_CAP_AW;
constexpr int kTileCapB = SC_CAP_B;   // pass B tile: (x, y), (sx, sy), P of the three ranges: 40 B per entry, 37.5 KiB
#ifndef SC_DENSE_TILE
#define SC_DENSE_TILE (SC_TILE_W * 43 / 10)
#endif
constexpr int kDenseTile = SC_DENSE_TILE;  // 1100 entries for 256 particles (the usual tile has ~800)
constexpr int kSlotMax = 65535;    // lists are staged as u16 tile slots in pass A
constexpr int kRowSlotMax = 4095;  // ... and stored as 12-bit slots of the tile pass B will stage

// The neighbor table: one 32-byte row per sorted particle -- twenty entries of 12 bits (slots of the tile published in
// tileBoundsT, below: entry s in bits 12 s .. 12 s + 11) and, in the top five bits, the count -- written by pass A in two
// 16-byte stores and read by pass B in two 16-byte loads.  Neither pass has bandwidth to spare (pass B without its pair
// loop still takes 41 of its 55 us, pass A without search and pair math 29 of 54: what they move at the rate the fabric
// sustains), and the table was a third of what pass A writes.  A tile whose published ranges hold more than 4,095 entries
// -- a block beside a pile whose lists reach across thousands -- keeps its entries in the 32-bit table instead (`nbr`).
// (Round 4 began with twenty 16-bit entries and the count in 48 bytes; rounds 1-3 kept the table slot-major, entry s of all
// particles contiguous: twenty 2-byte loads per particle in pass B and up to twenty 2-byte stores in pass A.)
struct alignas(16) NbrRow {
  unsigned int w[kRowWords];
};
static_assert(sizeof(NbrRow) == 32 && 12 * kMaxNbr <= 32 * kRowCountWord + kRowCountShift && kMaxNbr < 32, "twenty 12-bit entries and a 5-bit count");
__device__ __host__ __forceinline__ int row_entry(const NbrRow& r, int s) {
  const int b = 12 * s, k = b >> 5, sh = b & 31;
  unsigned int v = r.w[k] >> sh;
  if (sh > 20) v |= r.w[k + 1] << (32 - sh);  // (the entry straddles two words; the last entry ends inside word 7)
  return (int)(v & 0xFFFu);
}
__device__ __host__ __forceinline__ int row_count(const NbrRow& r) { return (int)(r.w[kRowCountWord] >> kRowCountShift); }
// a thread's list as pass A holds it -- ten words of two 16-bit slots -- into a row: every pair of slots becomes 24 bits,
// four pairs fill three words
__device__ __forceinline__ NbrRow row_pack(const unsigned int (&lw)[kMaxNbr / 2], int count) {
  unsigned int p[kMaxNbr / 2];
#pragma unroll
  for (int k = 0; k < kMaxNbr / 2; ++k) p[k] = (lw[k] & 0xFFFu) | ((lw[k] >> 4) & 0xFFF000u);
  NbrRow r;
  r.w[0] = p[0] | (p[1] << 24);
  r.w[1] = (p[1] >> 8) | (p[2] << 16);
  r.w[2] = (p[2] >> 16) | (p[3] << 8);
  r.w[3] = p[4] | (p[5] << 24);
  r.w[4] = (p[5] >> 8) | (p[6] << 16);
  r.w[5] = (p[6] >> 16) | (p[7] << 8);
  r.w[6] = p[8] | (p[9] << 24);
  r.w[7] = (p[9] >> 8) | ((unsigned int)count << kRowCountShift);
  return r;
}

// A thread's list while it is being built, in LDS: kMaxNbr entries of 16 bits and SC_SCAN_BATCH spare ones (the writes of a
// full list, below), thread-major -- the row is then copied to the table as it is.  The stride is an odd number of
// 32-bit words, so that the rows of consecutive lanes start in different banks.
constexpr int kListWords = ((kMaxNbr + SC_SCAN_BATCH) / 2) | 1;
constexpr int kListStride = 4 * kListWords;
struct Lists {
  char* base;
  __device__ __forceinline__ unsigned short& operator()(int s, int t) const {
    return *(unsigned short*)(base + t * kListStride + 2 * s);
  }
  __device__ __forceinline__ unsigned int& word(int k, int t) const { return *(unsigned int*)(base + t * kListStride + 4 * k); }
};

struct Tile {
  int a0, n0;  // same rows
  int a1, n1;  // next rows
  int a2, n2;  // previous rows
};

__device__ __host__ __forceinline__ int tile_index(const Tile& t, int slot) {  // tile slot -> sorted index
  if (slot < t.n0) return t.a0 + slot;
  slot -= t.n0;
  if (slot < t.n1) return t.a1 + slot;
  return t.a2 + (slot - t.n1);
}

// A neighbor-table entry is a tile slot (>= 0) or, for a tile too large for u16 slots, -(index+1).
__device__ __host__ __forceinline__ int entry_index(const Tile& t, int e) { return e >= 0 ? tile_index(t, e) : -e - 1; }

// 1/sqrt(s) to ~4e-15 relative (explicit fma: no decision is taken on it; the forces it enters are float-tolerance math,
// contract 1e-5, tests 1e-9).  s = 0 gives NaN downstream, like the reference's 0/0 (crate.py:174).
__device__ __forceinline__ double rsqrt_nr(double s) {
  // v_rsq_f64 is good to ~2^-24 (measured 5e-8); one Newton step, y (1 + e/2) with e = 1 - s y^2 (|e| <= 1e-7), leaves
  // 3/8 e^2 = 4e-15 in four fp64 operations (the third-order step y (1 + e/2 + 3e^2/8) reached 2e-16 in five: one
  // instruction per pair and pass more, 0.7 % of the tick, for digits nothing downstream resolves)
  const double y = __builtin_amdgcn_rsq(s);
  const double e = fma(-(s * y), y, 1.0);
  return fma(y * 0.5, e, y);
}


// XCD-aware block -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and
// b + 8 share an L2), while a tile overlaps its neighbors in the sorted order (same rows) and the
// tiles one grid row away (next / previous rows).  Giving every XCD one contiguous run of tiles
// keeps those overlaps inside one L2 instead of fetching them once per XCD (measured with
// FETCH_SIZE: profiles/).  Placement is a speed matter only; nothing depends on it.
// `tiles`: the blocks expected to hold particles.  A slab's grid is sized by its capacity; dealing ALL of it into
// runs would give the last XCDs nothing but empty blocks (measured on a slab with 30 % slack: pass A +15 %, pass B
// +13 %).  Blocks beyond `tiles` keep their own index.
__device__ __forceinline__ int tile_of_block(int tiles) {
  const int nb = min((int)gridDim.x, tiles), b = blockIdx.x;
  if (b >= nb) return b;
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  return xcd * q + min(xcd, r) + (b >> 3);
}
__device__ __forceinline__ int tiles_expected(const World& w) { return (w.live_hint + kTileW - 1) / kTileW; }

// The same runs, walked from both ends towards the middle (the search).  Blocks start in index order and the
// kernel ends with its slowest block: in a pile-up those are the blocks beside the piles along the floor and the
// ceiling -- the first tiles of the first XCD's run and the last tiles of the last one's, which in plain order start
// when everything else is nearly done.  Two contiguous fronts per XCD keep the overlaps in its L2 as before.
__device__ __forceinline__ int tile_of_block_ends_first(int tiles) {
  const int nb = min((int)gridDim.x, tiles), b = blockIdx.x;
  if (b >= nb) return b;
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  const int start = xcd * q + min(xcd, r), len = q + (xcd < r ? 1 : 0), l = b >> 3;
  return (l & 1) ? start + len - 1 - (l >> 1) : start + (l >> 1);
}

// Phases 3-5 of pass A for one particle.  LDS: where the tile is (compile time, see the header).
template <int NOISE, bool ENUM, bool DENS, bool LDS, int CAP, bool STAGE = false>
__device__ __forceinline__ void pass_a_body(const World& w, const Tile& tl, const int total, XY* txy,
                                            const Lists list, int* wkey, const int t, const int i,
                                            const bool live, const int idi, const int e0, const int b0, const int b1,
                                            const int e1, const int bm, const int em, const XY* __restrict__ sxy,
                                            int* __restrict__ nbr,
                                            NbrRow* __restrict__ rows, const int cap,
                                            const double* __restrict__ eta, const int* __restrict__ offById,
                                            double* __restrict__ P, XY* __restrict__ snn,
                                            const int tile_id, int* __restrict__ tileBoundsT) {
  auto load_xy = [&](int slot) -> XY {
    if constexpr (LDS) {
      return txy[slot];
    } else {
      const int j = tile_index(tl, slot);
      return sxy[j];
    }
  };
  // (a launch that builds the lists numbers the tile's slots in 16 bits; one that reads them back from the table -- the
  // density pass of SC_NOISE_HOST on the published tile -- finds them there only if the row's 12 bits could hold them)
  const bool slots_fit = total <= (ENUM ? kSlotMax : kRowSlotMax);

  // 3. neighbor list of particle i: four serial scans in the reference's order, entries are tile slots
  int C = 0;
  int round = 0, rws = -(1 << 30);  // windowed search: round counter, first slot of the resident window
  const int self = i - tl.a0;
  XY pi = {0.0, 0.0};
  if (live) pi = load_xy(self);
  if (ENUM && !diag::kNoSearch) {
    if (slots_fit && (LDS ? live : true)) {
      const double xi = pi.x, yi = pi.y;
      // One scan: `count` candidates from tile slot `first`, walking by `step`, examined one by one in
      // order with the reference's window conditions: window(xj, xi) says 0 = stop the scan, 1 =
      // outside the window, 2 = inside.
      //   LDS tile: straight from the tile.
      //   Tile too large for LDS (a block in or next to a pile-up; all threads of the block take
      //   part): the workgroup moves a window of CAP slots over the tile, placed on a grid of
      //   half windows around the unfinished thread that is furthest behind, staged with coalesced
      //   loads and kept for as long as somebody has candidates inside (a tile a little over the
      //   LDS budget needs two windows for all four scans).  Inside a window a thread first walks
      //   kSerial of its own candidates; ranges longer than that (a sparse particle walking a whole
      //   pile in the next row, hardly a hit among thousands) are then taken one owner at a time
      //   by the whole wave, 64 candidates per step, hits ranked by lane = scan order.  Every
      //   particle still sees its candidates in the reference's order, so the lists are the same.
      constexpr int kHalf = CAP / 2, kSerial = 32;
      WindowProbe probe;  // (diagnostic builds only: sc_diag.h)
      const double dstop = w.d * (1.0 + 0x1p-20);
      // next free entry of this thread's list (LDS tiles) as a byte offset into `list`; the entries from kMaxNbr on are spare
      // ones that take the writes of a full list, so that an append is a store and an add -- no branch, one clamp per batch
      constexpr unsigned kRow = sizeof(unsigned short);
      char* const lbase = list.base;
      const unsigned lo0 = t * (unsigned)kListStride, lend = kMaxNbr * kRow + lo0;
      unsigned lo = lo0;
      auto scan = [&](bool want, int first, int count, int step, auto window) {
        if constexpr (LDS) {
          // kBatch candidates per iteration: their LDS reads are issued together (one latency per batch
          // instead of one per candidate) and their tests are independent instruction streams.  The loop
          // stops on a conservative test of dx (never before the reference's window ends: a candidate
          // beyond d (1 + 2^-20) in x can neither be in the window nor within d); a hit is decided
          // exactly -- the reference's window expression AND the distance predicate.
          constexpr int kBatch = SC_SCAN_BATCH;
          // The last candidate of a batch may lie one to kBatch - 1 slots past the range: the tile array has that
          // much padding at both ends, what is read there is never a hit (v + k < count).  x is monotone along a
          // scan, so the last candidate of the batch decides the stop.
          bool done = !want || lo == lend;
          for (int v = 0; v < count && !done; v += kBatch) {
            XY q[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) q[k] = txy[first + (v + k) * step];
            // A candidate within the distance is inside the reference's window unless its |dx| is within 2^-20 d of d (then
            // the rounding of x +- d could decide otherwise: World::dsafe, sandcrate_hip.hip) -- the window expression
            // itself (two additions and two compares per candidate in the backward scans) is evaluated only for a batch
            // in which some lane has such a hit: exact ties at distance d, i.e. tests, not fluids.
            unsigned inc[kBatch];
            bool edge = false;
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
              const double dx = q[k].x - xi, dy = q[k].y - yi;
#ifdef SC_ABL_NOVALID
              const bool near = (dx * dx + dy * dy <= w.t_nbr);  // (timing ablation only: reads past the range may hit)
#else
              const bool near = (dx * dx + dy * dy <= w.t_nbr) & (v + k < count);
#endif
              edge |= near & !(fabs(dx) < w.dsafe);
              inc[k] = near ? kRow : 0u;
            }
            if (__ballot(edge)) {
#pragma unroll
              for (int k = 0; k < kBatch; ++k)
                if (window(q[k].x, xi) != 2) inc[k] = 0u;
            }
            const double dxl = q[kBatch - 1].x - xi;
            const bool stop = step > 0 ? dxl > dstop : dxl < -dstop;
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {  // trim (:91-93): a full list writes its spare rows
              *(unsigned short*)(lbase + lo) = (unsigned short)(first + (v + k) * step);
              lo += inc[k];
            }
            lo = min(lo, lend);  // once per batch: a list that fills up inside a batch runs into the kBatch spare rows
            done = stop | (lo == lend);
          }
        } else {
          const int lane = t & 63, wave0 = t & ~63;
          int pos = first, left = want ? count : 0;
          bool fresh = true;  // this thread has not walked its serial stretch of this scan yet
          probe.scan_begins(left);
          probe.clock_other();
          for (;;) {
            // the unfinished position that is furthest behind, block-wide (keys double-buffered by round)
            const int key = wave_min_all(left > 0 ? pos * step : INT_MAX);
            int* wk = wkey + (round & 1) * (kTileW / 64);
            ++round;
            if (lane == 0) wk[t >> 6] = key;
            __syncthreads();
            int k0 = wk[0];
#pragma unroll
            for (int k = 1; k < kTileW / 64; ++k) k0 = min(k0, wk[k]);
            if (k0 == INT_MAX) break;  // uniform: nobody has candidates left in this range
            probe.round();
            const int p0 = k0 * step;
            probe.clock_round();
            if ((unsigned)(p0 - rws) >= (unsigned)CAP) {  // not in the resident window: stage the one around it
              rws = step > 0 ? (p0 / kHalf) * kHalf : max(0, (p0 / kHalf - 1) * kHalf);
              probe.staging();
              constexpr int kPer = (CAP + kTileW - 1) / kTileW;  // (CAP need not be a multiple of the tile width)
              XY r[kPer];
#pragma unroll
              for (int k = 0; k < kPer; ++k) {
                const int slot = rws + t + k * kTileW;
                if (slot < total && t + k * kTileW < CAP) {
                  const int j = tile_index(tl, slot);
                  r[k] = sxy[j];
                }
              }
#pragma unroll
              for (int k = 0; k < kPer; ++k)
                if (rws + t + k * kTileW < total && t + k * kTileW < CAP) txy[t + k * kTileW] = r[k];
              __syncthreads();
            }
            probe.clock_stage();
            const int ws = rws;
            auto inside = [&](int p) { return (unsigned)(p - ws) < (unsigned)CAP; };
            // This thread's next stretch inside the window, at most kSerial candidates, walked like an LDS tile's scan:
            // kWinBatch candidates per iteration, their LDS reads issued together, hits decided exactly, an append is a
            // store and a clamped add (no branch), the batch's last candidate decides a conservative stop.  For that
            // last candidate to be one of the scan's own the stretch is a multiple of kWinBatch unless the range ends
            // in it (up to three candidates before the window's end wait for the wave-wide turn or the next window).
            constexpr int kWinBatch = SC_SCAN_BATCH;
            static_assert(kSerial % kWinBatch == 0, "the serial stretch is a whole number of batches");
            {
              const int avail = step > 0 ? ws + CAP - pos : pos - ws + 1;  // slots of the window from pos on
              const int lim = left <= avail ? left : (avail & ~(kWinBatch - 1));
#if SC_SERIAL_ONCE
              // a thread walks kSerial candidates on its own ONCE per scan, in the first window that holds its position:
              // what is left after that is a long walk (the serial stretch ends most scans), and in the windows that follow
              // its next 32 candidates would cost the wave this loop's eight iterations and spare the wave-wide turn nothing
              // (the heaviest tiles of a pile-up: nine windows, 45 k of 160 k cycles in this loop)
              const bool mine_now = left > 0 && inside(pos);
              const int cnt = mine_now && fresh ? min(lim, kSerial) : 0;
              fresh = fresh && !mine_now;
#else
              const int cnt = left > 0 && inside(pos) ? min(lim, kSerial) : 0;
#endif
              const int base = pos - ws;
              unsigned lw = lo0 + (unsigned)C * kRow;
              bool stopped = false;
              int v = 0;
              for (; v < cnt && !stopped; v += kWinBatch) {
                XY q[kWinBatch];
#pragma unroll
                for (int k = 0; k < kWinBatch; ++k) q[k] = txy[base + (v + k) * step];
                unsigned inc[kWinBatch];
                bool edge = false;  // (as in the LDS scan: the window expression only for a batch with a hit at the window's edge)
#pragma unroll
                for (int k = 0; k < kWinBatch; ++k) {
                  const double dx = q[k].x - xi, dy = q[k].y - yi;
                  const bool near = (dx * dx + dy * dy <= w.t_nbr) & (v + k < cnt);
                  edge |= near & !(fabs(dx) < w.dsafe);
                  inc[k] = near ? kRow : 0u;
                }
                if (__ballot(edge)) {
#pragma unroll
                  for (int k = 0; k < kWinBatch; ++k)
                    if (window(q[k].x, xi) != 2) inc[k] = 0u;
                }
                const double dxl = q[kWinBatch - 1].x - xi;
                const bool stop = step > 0 ? dxl > dstop : dxl < -dstop;
#pragma unroll
                for (int k = 0; k < kWinBatch; ++k) {  // trim (:91-93): a full list writes its spare rows
                  *(unsigned short*)(lbase + lw) = (unsigned short)(pos + (v + k) * step);
                  lw += inc[k];
                }
                lw = min(lw, lend);
                stopped = stop | (lw == lend);
              }
              const int walked = min(v, cnt);
              C = (int)((lw - lo0) / kRow);
              pos += walked * step;
              left = stopped ? 0 : left - walked;
            }
            probe.clock_serial();
            unsigned long long m = __ballot(left > 0 && inside(pos));
            while (m) {
              probe.turn();
              // the owner's state is read with v_readlane (the owner is wave-uniform): no LDS round trips on the way
              const int owner = __builtin_amdgcn_readfirstlane(__ffsll(m) - 1);
              auto of = [&](int v) { return __builtin_amdgcn_readlane(v, owner); };
              int opos = of(pos), oleft = of(left);
              const int oC = of(C);
              const double oxi = __hiloint2double(of(__double2hiint(xi)), of(__double2loint(xi)));
              const double oyi = __hiloint2double(of(__double2hiint(yi)), of(__double2loint(yi)));
              const int avail = step > 0 ? ws + CAP - opos : opos - ws + 1;  // slots of the window from opos on
              int span = min(oleft, avail);
              // A long stretch ahead: 64 probes spread over it.  The candidates before the window (verdict 1: the
              // adjacent-row scans start at the window cell's first particle, in a pile thousands short of x_i - d)
              // are a prefix of the scan, so everything up to the last leading probe that is still outside is skipped.
              if (span > 4 * 64) {
                const int stride = span >> 6;
                const XY q = txy[opos + lane * stride * step - ws];
                const unsigned long long out = __ballot(window(q.x, oxi) == 1);
                const int lead = out == ~0ull ? 64 : __ffsll((long long)~out) - 1;
                if (lead > 0) {
                  const int skip = (lead - 1) * stride + 1;
                  opos += skip * step;
                  oleft -= skip;
                  span -= skip;
                }
              }
              // the rest of the owner's stretch inside this window, kCoop x 64 candidates per step (their LDS reads
              // issued together), hits ranked in scan order
              constexpr int kCoop = SC_COOP;
              const int need = kMaxNbr - oC;
              int taken = 0, done = 0;
              bool stopped = false;
              while (done < span && !stopped) {
                const int nc = min(span - done, 64 * kCoop);
                XY q[kCoop];
#pragma unroll
                for (int c = 0; c < kCoop; ++c) q[c] = txy[c * 64 + lane < nc ? opos + (done + c * 64 + lane) * step - ws : 0];
#pragma unroll
                for (int c = 0; c < kCoop; ++c) {
                  if (!stopped && c * 64 < nc) {  // wave-uniform
                    const bool valid = c * 64 + lane < nc;
                    const double dx = q[c].x - oxi, dy = q[c].y - oyi;
                    const bool near = dx * dx + dy * dy <= w.t_nbr;
#if SC_COOP_FAST
                    // 64 candidates none of which is within the distance or past the window's end (the rule in these
                    // walks: a sparse particle beside a pile of thousands) leave nothing to record: no verdicts, no ranks
                    if (!__ballot(valid && (near || window(q[c].x, oxi) == 0))) continue;
#endif
                    const int verdict = valid ? window(q[c].x, oxi) : 1;
                    const bool hit = verdict == 2 && near;
                    const unsigned long long stopm = __ballot(verdict == 0);
                    const int nlive = stopm ? __ffsll(stopm) - 1 : 64;  // candidates ahead of the stop
                    const unsigned long long hitm = __ballot(hit && lane < nlive);
                    const int rank = taken + __popcll(hitm & ((1ull << lane) - 1ull));
                    if (hit && lane < nlive && rank < need)
                      list(oC + rank, wave0 + owner) = (unsigned short)(opos + (done + c * 64 + lane) * step);
                    taken = min(taken + (int)__popcll(hitm), need);
                    stopped = stopm != 0 || taken == need;
                  }
                }
                done += nc;
              }
              if (lane == owner) {
                C = oC + taken;
                pos = opos + done * step;
                left = stopped ? 0 : oleft - done;
              }
              m = __ballot(left > 0 && inside(pos));
            }
            probe.clock_coop();
          }
          probe.clock_round();
        }
      };
      // same strip, after i: x_j <= x_i + d                                  (:106-109)
      scan(live, self + 1, e0 - (i + 1), 1, [&](double xj, double xq) { return xj > xq + w.d ? 0 : 2; });
      SC_STAMP(0, 3);
      probe.hits_after_first(C);
      // The scans of the adjacent strips start at a CELL's first particle, up to a cell's worth of candidates short of the
      // window (x_i - d lies anywhere in the cell before the particle's): an LDS tile's thread first finds, among the
      // first seven candidates, the ones that lie before the window even by the conservative stop distance -- x is
      // monotone along a scan, three probes -- and starts behind them.  (What is skipped can be neither inside the
      // window nor within the distance: the lists are the same.  A wave's scan runs as long as its slowest lane, and the
      // lanes with the longest lead-in were the slowest.)
      auto lead_in = [&](int first, int count, int step) -> int {
        int s = 0;
        if constexpr (LDS && SC_LEAD_IN) {
#pragma unroll
          for (int h = 4; h >= 1; h >>= 1) {
            const int probe = s + h - 1;
            const double xj = txy[first + min(probe, max(count - 1, 0)) * step].x;
            const bool before = probe < count && (step > 0 ? xj < xi - dstop : xj > xi + dstop);
            s += before ? h : 0;
          }
        }
        return s;
      };
      // next strip: x_i - d <= x_j <= x_i + d                                (:112-119)
      {
        const int first = tl.n0 + (b1 - tl.a1), count = e1 - b1, s = lead_in(first, count, 1);
        scan(live && C < kMaxNbr, first + s, count - s, 1,
             [&](double xj, double xq) { return xj > xq + w.d ? 0 : (xj >= xq - w.d ? 2 : 1); });
      }
      SC_STAMP(0, 4);
      probe.hits_after_second(C);
      // reverse edges (:85-88): i is a forward candidate of j, same strip
      scan(live && C < kMaxNbr, self - 1, i - b0, -1, [&](double xj, double xq) { return !(xq <= xj + w.d) ? 0 : 2; });
      SC_STAMP(0, 5);
      // reverse edges from the previous strip
      {
        const int first = tl.n0 + tl.n1 + (em - 1 - tl.a2), count = em - bm, s = lead_in(first, count, -1);
        scan(live && C < kMaxNbr, first - s, count - s, -1,
             [&](double xj, double xq) { return !(xq <= xj + w.d) ? 0 : (xq >= xj - w.d ? 2 : 1); });
      }
      if constexpr (LDS) C = (int)((lo - lo0) / kRow);
      probe.flush();
    } else if (live) {
      // a tile beyond 65535 particles (a block inside one gigantic bucket) cannot use u16 slots:
      // entries go straight to the table as -(index+1); correctness path only
      const double xi = pi.x, yi = pi.y;
      const double xhi = xi + w.d, xlo = xi - w.d;
      auto push = [&](int j) { nbr[(size_t)C++ * cap + i] = -j - 1; };
      for (int j = i + 1; j < e0 && C < kMaxNbr; ++j) {
        const XY pj = sxy[j];
        const double xj = pj.x;
        if (xj > xhi) break;
        const double dx = xj - xi, dy = pj.y - yi;
        if (dx * dx + dy * dy <= w.t_nbr) push(j);
      }
      for (int j = b1; j < e1 && C < kMaxNbr; ++j) {
        const XY pj = sxy[j];
        const double xj = pj.x;
        if (xj > xhi) break;
        if (xj >= xlo) {
          const double dx = xj - xi, dy = pj.y - yi;
          if (dx * dx + dy * dy <= w.t_nbr) push(j);
        }
      }
      for (int j = i - 1; j >= b0 && C < kMaxNbr; --j) {
        const XY pj = sxy[j];
        const double xj = pj.x;
        if (!(xi <= xj + w.d)) break;
        const double dx = xj - xi, dy = pj.y - yi;
        if (dx * dx + dy * dy <= w.t_nbr) push(j);
      }
      for (int j = em - 1; j >= bm && C < kMaxNbr; --j) {
        const XY pj = sxy[j];
        const double xj = pj.x;
        if (!(xi <= xj + w.d)) break;
        if (xi >= xj - w.d) {
          const double dx = xj - xi, dy = pj.y - yi;
          if (dx * dx + dy * dy <= w.t_nbr) push(j);
        }
      }
    }
  } else if (live) {
    // lists were built by an earlier launch: bring them in
    const NbrRow row = rows[i];
    C = row_count(row);
    if (slots_fit) {
#pragma unroll
      for (int k = 0; k < kMaxNbr / 2; ++k) list.word(k, t) = (unsigned)row_entry(row, 2 * k) | ((unsigned)row_entry(row, 2 * k + 1) << 16);
    }
  }

  if (diag::kNoSearch) C = 0;
  SC_STAMP(0, 6);
  // 3b. The table's slots refer to the ranges published in tileBoundsT -- for the usual tile the candidate ranges.  A
  // tile whose candidate ranges exceed pass B's LDS budget (a block in or beside a pile: thousands of candidates,
  // of which the lists name a few hundred) publishes the ranges its lists actually reach instead -- per range from
  // the lowest to the highest slot named, the block's own particles included -- and renumbers its entries, so that
  // pass B stages these tiles in LDS like any other instead of gathering every neighbor from global memory.  A tile
  // that was searched through the window does the same for its own pair math: the reach (a few hundred entries) is
  // staged in the window's LDS and the entries are renumbered in the lists before the pair loop, which then reads its
  // neighbors like an LDS tile's (gathered from global memory, four per round trip, it took three times as long).
  const int sr1 = tl.n0, sr2 = tl.n0 + tl.n1;  // first slot of the next rows' / previous rows' range
  const bool strim = STAGE && ENUM && slots_fit && total > kTileCapB;  // uniform over the workgroup
  bool staged = false;                       // the reach is in txy and the lists hold renumbered entries
  // The reach, from the waves' extremes in wkey: nb[0..5] = the three index ranges, sub[0..2] = what renumbering takes
  // off an entry of each range; -> entries in all.  Derived where it is needed (here for the staging, below for the
  // table) rather than kept: a dozen uniform values alive across the pair loop spilled scalar registers in every tile.
  auto reach = [&](int* nb, int* sub) -> int {
    int lo0 = INT_MAX, hi0 = -1, lo1 = INT_MAX, hi1 = -1, lo2 = INT_MAX, hi2 = -1;
#pragma unroll
    for (int k = 0; k < kTileW / 64; ++k) {
      const int* wk = wkey + 6 * k;
      lo0 = min(lo0, wk[0]); hi0 = max(hi0, wk[1]);
      lo1 = min(lo1, wk[2]); hi1 = max(hi1, wk[3]);
      lo2 = min(lo2, wk[4]); hi2 = max(hi2, wk[5]);
    }
    const int m0 = hi0 - lo0 + 1;  // never empty: the block's own particles
    const int m1 = hi1 >= lo1 ? hi1 - lo1 + 1 : 0, m2 = hi2 >= lo2 ? hi2 - lo2 + 1 : 0;
    if (m1 == 0) lo1 = sr1;
    if (m2 == 0) lo2 = sr2;
    nb[0] = tl.a0 + lo0;
    nb[1] = nb[0] + m0;
    nb[2] = tl.a1 + (lo1 - sr1);
    nb[3] = nb[2] + m1;
    nb[4] = tl.a2 + (lo2 - sr2);
    nb[5] = nb[4] + m2;
    sub[0] = lo0;
    sub[1] = lo1 - m0;
    sub[2] = lo2 - m0 - m1;
    return m0 + m1 + m2;
  };
  // the waves' extremes of the slots the lists name, per range, into wkey (every thread of the workgroup calls this)
  auto extremes = [&]() {
    int lo0 = live ? self : INT_MAX, hi0 = live ? self : -1, lo1 = INT_MAX, hi1 = -1, lo2 = INT_MAX, hi2 = -1;
    if (live)
      for (int s = 0; s < C; ++s) {
        const int e = list(s, t);
        if (e < sr1) {
          lo0 = min(lo0, e);
          hi0 = max(hi0, e);
        } else if (e < sr2) {
          lo1 = min(lo1, e);
          hi1 = max(hi1, e);
        } else {
          lo2 = min(lo2, e);
          hi2 = max(hi2, e);
        }
      }
    lo0 = wave_min_all(lo0);
    hi0 = wave_max_all(hi0);
    lo1 = wave_min_all(lo1);
    hi1 = wave_max_all(hi1);
    lo2 = wave_min_all(lo2);
    hi2 = wave_max_all(hi2);
    __syncthreads();  // the scans are done with wkey (and with the window)
    if ((t & 63) == 0) {
      int* wk = wkey + 6 * (t >> 6);
      wk[0] = lo0; wk[1] = hi0; wk[2] = lo1; wk[3] = hi1; wk[4] = lo2; wk[5] = hi2;
    }
    __syncthreads();
  };
  if constexpr (STAGE && !LDS) if (strim) {
    extremes();
    if constexpr (!LDS) {
      if (DENS) {
        int nb[6], sub[3];
        const int mt = reach(nb, sub);
        if (mt <= CAP) {  // uniform
          staged = true;
          if (live)
            for (int s = 0; s < C; ++s) {
              const int e = list(s, t);
              list(s, t) = (unsigned short)(e - (e < sr1 ? sub[0] : e < sr2 ? sub[1] : sub[2]));
            }
          const Tile part{nb[0], nb[1] - nb[0], nb[2], nb[3] - nb[2], nb[4], nb[5] - nb[4]};
          for (int slot = t; slot < mt; slot += kTileW) {
            const int j = tile_index(part, slot);
            txy[slot] = sxy[j];
          }
          __syncthreads();
        }
      }
    }
  }
  // 4. pair math of pass A: populate_colliders (crate.py:161-175), pressures (:261-275), normals (:337-342)
  if (DENS && live) {
    // no decision is taken in this block (P, s are float-tolerance outputs): multiply-adds may fuse here although the
    // file is compiled with -ffp-contract=off
#pragma clang fp contract(fast)
    // Per pair (crate.py:167-174, :270, :342), arranged for the fewest float64 instructions:
    //   r = p_i - (p_j + eta),  |r|^2 = s2,  1/|r| = rinv,  c = clip(|r| / d, 0, 1)  (|r| >= 0: only the upper clip acts)
    //   overlap w = 1 - c  ->  sum_w = C - sum c;   (1 - w) w n = c (1 - c) rinv r
    // A zero distance gives rinv = NaN: max(NaN, 0) = 0 makes c = 0, overlap 1 -- what the reference computes from
    // dist = 0 (crate.py:270) -- while g = NaN reaches ax / ay like the reference's 0/0 (crate.py:174).
    double sumc = 0, ax = 0, ay = 0;
    const int off = (NOISE == SC_NOISE_HOST) ? offById[idi] : 0;
    uint64_t z = noise_base(w.noise_key, idi);
    const double ox = pair_origin<NOISE>(w, pi.x), oy = pair_origin<NOISE>(w, pi.y);
    constexpr int kFetch = LDS ? 1 : 4;  // global-memory tiles: four neighbors per round trip
    XY qq[kFetch];
    const int Cloop = diag::pairs_a(C);
    if (STAGE && !LDS && staged) {  // the neighbors are in the staged reach: an LDS tile's loop
      for (int s = 0; s < Cloop; ++s) {
        const XY q = txy[list(s, t)];
        double rx, ry;
        pair_offset<NOISE>(w, z, s, eta, off, ox - q.x, oy - q.y, rx, ry);
        z += kGold;
        const double s2 = rx * rx + ry * ry;
        const double rinv = rsqrt_nr(s2);
        const double c = fmin(fmax(s2 * rinv * w.inv_d, 0.0), 1.0);
        sumc += c;
        const double g = fma(-c, c, c) * rinv;
        ax += g * rx;
        ay += g * ry;
      }
    } else
    for (int s = 0; s < Cloop; ++s) {
      if (s % kFetch == 0) {
#pragma unroll
        for (int k = 0; k < kFetch; ++k) {
          const int sk = s + k;
          if (slots_fit) {
            qq[k] = load_xy(sk < C ? (int)list(sk, t) : self);
          } else {
            const int j = sk < C ? -nbr[(size_t)sk * cap + i] - 1 : i;
            qq[k] = sxy[j];
          }
        }
      }
      XY q = qq[0];
#pragma unroll
      for (int k = 1; k < kFetch; ++k)
        if (s % kFetch == k) q = qq[k];
      double rx, ry;
      pair_offset<NOISE>(w, z, s, eta, off, ox - q.x, oy - q.y, rx, ry);
      z += kGold;
      const double s2 = rx * rx + ry * ry;
      const double rinv = rsqrt_nr(s2);
      const double c = fmin(fmax(s2 * rinv * w.inv_d, 0.0), 1.0);  // crate.py:270: clip(dist / d, 0, 1)
      sumc += c;
      const double g = fma(-c, c, c) * rinv;             // crate.py:342 with n = r / dist (:174)
      ax += g * rx;
      ay += g * ry;
    }
    P[i] = C ? fmax(((double)C - sumc) - w.ignored, 0.0) : 0.0;  // crate.py:265-273
    snn[i] = XY{ax, ay};
  }

  SC_STAMP(0, 7);
  // 5. lists out: the thread's row of the table -- its list as it stands in LDS, twelve bits per entry, and the count -- in
  // two 16-byte stores.
  // The table's slots refer to the ranges published in tileBoundsT -- for the usual tile the candidate ranges, for a tile
  // whose candidate ranges exceed pass B's LDS budget the reach of its lists, the entries renumbered (3b above).
  if (ENUM) {
    int nb[6] = {tl.a0, tl.a0 + tl.n0, tl.a1, tl.a1 + tl.n1, tl.a2, tl.a2 + tl.n2}, sub[3] = {0, 0, 0};
    bool trim = slots_fit && total > kTileCapB;  // uniform over the workgroup
    bool in_rows = slots_fit;                     // the entries go into the rows (else: into the 32-bit table)
    if (trim) {
      if (!(STAGE && !LDS)) extremes();  // (a tile that staged its reach for the pair math has them in wkey already)
      int rnb[6], rsub[3];
      if (reach(rnb, rsub) <= kRowSlotMax) {
#pragma unroll
        for (int k = 0; k < 6; ++k) nb[k] = rnb[k];
        sub[0] = rsub[0]; sub[1] = rsub[1]; sub[2] = rsub[2];
      } else {  // lists that reach across more than a row's 12 bits can number (never a tile that staged its reach: that fits the window)
        trim = false;
        in_rows = false;
      }
    }
    if (t == 0) {
      int* tbT = tileBoundsT + 6 * tile_id;
      tbT[0] = nb[0]; tbT[1] = nb[1]; tbT[2] = nb[2]; tbT[3] = nb[3]; tbT[4] = nb[4]; tbT[5] = nb[5];
    }
    if (live) {
      C = min(C, kMaxNbr);  // (never more by construction; the rows of the table end there)
      unsigned int lw[kMaxNbr / 2];
#pragma unroll
      for (int k = 0; k < kMaxNbr / 2; ++k) lw[k] = 0u;
      if (in_rows) {
        if (trim && !staged) {
          for (int s = 0; s < C; ++s) {
            const int e = list(s, t);
            list(s, t) = (unsigned short)(e - (e < sr1 ? sub[0] : e < sr2 ? sub[1] : sub[2]));
          }
        }
#pragma unroll
        for (int k = 0; k < kMaxNbr / 2; ++k) lw[k] = list.word(k, t);  // (entries from C on: whatever the scans left there)
      } else if (slots_fit) {  // (a tile beyond 16-bit slots wrote the 32-bit table as it searched)
        for (int s = 0; s < C; ++s) nbr[(size_t)s * cap + i] = -tile_index(tl, (int)list(s, t)) - 1;
      }
      rows[i] = row_pack(lw, C);
    }
  }
  SC_STAMP(0, 8);
}

// ------------------------------------------------------------------------------------------
// Pass A.  ENUM: build the neighbor lists in the reference's canonical order
//   [same row, to the right, x ascending] [row+1, x ascending]
//   [same row, to the left, x descending] [row-1, x descending], cut at 20
// (collision_detector.py:9-121: forward candidates :106-119 with the distance filter :75-80, a
// reverse edge j->i exists exactly when i is a forward candidate of j :85-88, trim :91-93) and
// write them slot-major with the counts.  DENS: populate_colliders (crate.py:161-175),
// compute_particle_pressures (:261-275) and pass 1 of apply_tension (:337-342) -> P, sx, sy.
// One launch does both, except in SC_NOISE_HOST mode where the host's noise block can only be
// indexed after all counts are known: then <ENUM only> runs in sc_step_begin and <DENS only> in
// sc_step_finish.
// ------------------------------------------------------------------------------------------
// STAGE: tiles searched through the window stage the reach of their lists for the pair math (pass_a_body, 3b).  Its own
// instantiation, launched while the scans report big buckets: inlined into the one kernel the extra paths cost the
// uniform regime 13 spilled scalar registers and 1 us per tick.
template <int NOISE, bool ENUM, bool DENS, int CAP, bool STAGE = false>
__global__ void __launch_bounds__(kTileW)
    k_pass_a(World w, const int* __restrict__ counters, const XY* __restrict__ sxy,
             const int* __restrict__ id, const int* __restrict__ cell, Buckets bk,
             int* __restrict__ nbr, NbrRow* __restrict__ rows, int cap,
             const double* __restrict__ eta, const int* __restrict__ offById, double* __restrict__ P, XY* __restrict__ snn,
             const int* __restrict__ tileBounds, int* __restrict__ tileBand,
             int* __restrict__ tileBoundsT) {
  constexpr int kPad = SC_SCAN_BATCH - 1;  // a batched scan may read this far past either end of the tile
  __shared__ XY txy_padded[CAP + 2 * kPad];
  XY* const txy = txy_padded + kPad;
  __shared__ unsigned int list_words[kTileW * kListWords];  // tile slots of the neighbors, a row per thread (Lists)
  const Lists list{(char*)list_words};
  __shared__ int wkey[6 * (kTileW / 64)];  // the windowed scans' round keys (2 per wave); the lists' reach per range (6 per wave)

  const int t = threadIdx.x;
  SC_TIMELINE_KERNEL(0);
  const int tile_id = tile_of_block_ends_first(tiles_expected(w));
  const int i0 = tile_id * kTileW;
  const int i = i0 + t;
  SC_STAMP(0, 0);
  // everything that does not depend on the live count is requested before the count is waited for
  const int ic = min(i, cap - 1);
  const int cpacked = cell[ic];
  const int idi = (DENS && NOISE != SC_NOISE_NONE) ? id[ic] : 0;
  // the tile's three ranges: k_reorder published them, so staging need not wait for the bucket lookups below
  const int* tb = tileBounds + 6 * tile_id;
  const int tb0 = tb[0], tb1 = tb[1], tb2 = tb[2], tb3 = tb[3], tb4 = tb[4], tb5 = tb[5];
  const int n = counters[C_NT];
  if (i0 >= n || tick_abandoned(counters)) return;
  const int m = min(kTileW, n - i0);
  const bool live = t < m;

  // slabs: does this block hold a particle that may end the tick in a halo band (its column within the band
  // plus the band margin of a cut)?  Pass B runs those blocks first and lets the halo exchange start while the
  // interior blocks are still computing (sc_set_halo_overlap).
  if (ENUM && w.slab) {
    bool band = false;
    if (live) {
      const int c = cpacked & kCellMask;
      const long long col = w.slab_axis ? (long long)(c / w.ncols) + w.row0 : (long long)(c % w.ncols) + w.col0;
      band = (w.has_left && col < w.own_lo + w.halo + w.band_margin) || (w.has_right && col >= w.own_hi - w.halo - w.band_margin);
    }
    const int any = __syncthreads_or(band);
    if (t == 0) tileBand[tile_id] = any;
  }

  // 1. the particle's own candidate ranges (cell -> six bucket boundaries)
  Tile tl;
  tl.a0 = tb0;
  tl.n0 = tb1 - tb0;
  tl.a1 = tb2;
  tl.n1 = tb3 - tb2;
  tl.a2 = tb4;
  tl.n2 = tb5 - tb4;
  int e0 = 0, b0 = 0, b1 = 0, e1 = 0, bm = 0, em = 0;
  if (live) {
    const int c = cpacked & kCellMask;
    e0 = bk(c + 2);
    b0 = bk(c - 1);
    b1 = bk(c + w.ncols - 1);
    e1 = bk(c + w.ncols + 2);
    bm = bk(c - w.ncols - 1);
    em = bk(c - w.ncols + 2);
  }
  const int total = tl.n0 + tl.n1 + tl.n2;
  // a tile that fits but is much denser than usual (some dense cell plus its sparse surroundings) is better off
  // on the windowed path, which hands long fruitless walks to the whole wave (measured: profiles/README.md)
  const bool in_lds = total <= min(CAP, kDenseTile);
  SC_STAMP_VALUE(0, 10, total);
  SC_STAMP_VALUE(0, 11, tile_id);

  // 2. stage (x, y) of the three ranges; every load of the tile is in flight before the first LDS write
  if (in_lds) {
    constexpr int kPer = (CAP + kTileW - 1) / kTileW;
    XY r[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int s = t + k * kTileW;
      if (s < total) {
        const int j = tile_index(tl, s);
        r[k] = sxy[j];
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int s = t + k * kTileW;
      if (s < total) txy[s] = r[k];
    }
  }
  SC_STAMP(0, 1);
  __syncthreads();
  SC_STAMP(0, 2);

  if (in_lds)
    pass_a_body<NOISE, ENUM, DENS, true, CAP>(w, tl, total, txy, list, wkey, t, i, live, idi, e0, b0, b1, e1, bm, em, sxy, nbr, rows,
                                         cap, eta, offById, P, snn, tile_id, tileBoundsT);
  else {
    // Beyond the LDS budget the threads take the block's particles in stride (thread t: particle (t mod 64) * waves +
    // t / 64).  Consecutive particles share their surroundings, and the expensive ones -- sparse particles beside a
    // pile in the adjacent row, each with a walk of hundreds of candidates for a handful of hits -- come in runs of
    // dozens: in storage order one wave would hold them all and walk them one owner at a time while the other
    // three wait at the round's barrier (measured: that one wave WAS the kernel's duration in the pile-up regime).
    constexpr int kWaves = kTileW / 64;
    const int ip = i0 + (t & 63) * kWaves + (t >> 6);
    const bool livep = ip - i0 < m;
    const int icp = min(ip, cap - 1);
    const int cp = cell[icp];
    const int idp = (DENS && NOISE != SC_NOISE_NONE) ? id[icp] : 0;
    if (livep) {
      const int c = cp & kCellMask;
      e0 = bk(c + 2);
      b0 = bk(c - 1);
      b1 = bk(c + w.ncols - 1);
      e1 = bk(c + w.ncols + 2);
      bm = bk(c - w.ncols - 1);
      em = bk(c - w.ncols + 2);
    }
    pass_a_body<NOISE, ENUM, DENS, false, CAP, STAGE && ENUM && DENS>(w, tl, total, txy, list, wkey, t, ip, livep, idp, e0, b0, b1, e1, bm, em, sxy, nbr,
                                          rows, cap, eta, offById, P, snn, tile_id, tileBoundsT);
  }
}

// ------------------------------------------------------------------------------------------
// Pass B "force + integrate": pass 2 of apply_tension (crate.py:343-353), apply_gravity (:309-310),
// apply_pressure (:295-307), apply_viscosity (:316-323), apply_wall_bounce (:245-259),
// apply_continuous_collision_velocity_fix (:177-200; geometry_utils.py:136-143, :182-222) and
// apply_particles_velocity (:360-361).  Reads the sorted arrays (through the LDS tile), writes the
// storage arrays in the same order: that is the next tick's input.
// ------------------------------------------------------------------------------------------
// The tile of pass B in LDS: (x, y), (sx, sy) and P of the three ranges -- 40 bytes per entry, 37.5 KiB for
// 256 particles, so that sixteen waves share a CU (a 64-byte record with the velocities allowed ten).  The
// neighbors' start-of-tick velocities are only summed (crate.py:175, :319-323): they are staged into the
// (x, y) array once the pair loop is done with it.
// f(slot) for slot = S, S + 1, ... while slot < count, the slots compile-time constants and the tests NESTED: a lane leaves at
// its count and the wave's exec mask only ever narrows (a flat `if (s < count)` per unrolled slot restores it every time:
// four control instructions per slot instead of three, and a wave whose lanes are all done still visits every slot).
template <int S, int N, class F>
__device__ __forceinline__ void nested_slots_flat(const int count, F&& f) {  // (the flat form, for comparison)
  if constexpr (S < N) {
    if (S < count) f(std::integral_constant<int, S>{});
    nested_slots_flat<S + 1, N>(count, f);
  }
}
template <int S, int N, class F>
__device__ __forceinline__ void nested_slots(const int count, F&& f) {
  if constexpr (S < N) {
    if (S < count) {
      f(std::integral_constant<int, S>{});
      nested_slots<S + 1, N>(count, f);
    }
  }
}

struct PairSums {
  double tx, ty;    // the velocity change of apply_tension + the particle part of apply_pressure, dt included
  double mtx, mty;  // force monitor only: the share of apply_tension in it
};

// Phase 3a of pass B for one particle: the pair loop.  LDS: where the tile is (compile time, see the
// header).  js[] are neighbor-table entries (tile slots, or -(index+1)); `self` is the particle's own slot.
//   tension  (crate.py:347-353): v += dt sum_j [ss ((s_i - s_j) . n_ij) + (P_i + P_j - 2 tp)] n_ij
//   pressure (crate.py:301-306): v += dt pamp sum_j (P_i + P_j) n_ij
// Neither reads a velocity, and gravity in between is a constant, so the two sums are taken together:
//   dv = sum_j w_j n_ij,   w_j = (dt ss) (ds . n_ij) + (P_i + P_j) dt (1 + pamp) - 2 tp dt,   n_ij = r rinv
// (one multiply-add chain per pair instead of two accumulations; rounding differs at 1e-16).
// `entry(s)`: the table entry of slot s (s is a compile-time constant at every call: the loops are unrolled) -- from the
// row's packed words, or from the 32-bit table of a gigantic tile.
template <int NOISE, bool LDS, bool MON, class Entry>
__device__ __forceinline__ PairSums pass_b_pairs(const World& w, const Tile& tl, const XY* txy, const XY* tss,
                                                 const double* tP, const int self, const int Cn, const int idi,
                                                 const Entry entry, const XY* __restrict__ sxy,
                                                 const double* __restrict__ eta,
                                                 const int* __restrict__ offById, const double* __restrict__ P,
                                                 const XY* __restrict__ snn,
                                                 double& xi, double& yi, double& Pi) {
  // No decision is taken in this block, so multiply-adds may fuse (the file is compiled with -ffp-contract=off).  The
  // velocities it produces do feed decisions later -- the sign test of the wall bounce and the orientation tests of the
  // crossing check -- which therefore see inputs that may differ from NumPy's by an ulp: inside the 1e-5 contract,
  // not bit-faithful (the reference's own sums over neighbors are order-dependent to the same degree).
#pragma clang fp contract(fast)
  auto load = [&](int e, XY& pos, XY& nrm, double& pr) {
    if constexpr (LDS) {
      pos = txy[e];
      nrm = tss[e];
      pr = tP[e];
    } else {
      const int j = entry_index(tl, e);
      pos = sxy[j];
      nrm = snn[j];
      pr = P[j];
    }
  };
  XY mp, ms;
  load(self, mp, ms, Pi);
  xi = mp.x;
  yi = mp.y;
  const double sxi = ms.x, syi = ms.y;
  const double ox = pair_origin<NOISE>(w, xi), oy = pair_origin<NOISE>(w, yi);
  const int off = (NOISE == SC_NOISE_HOST) ? offById[idi] : 0;
  const uint64_t zbase = noise_base(w.noise_key, idi);
  const double k_ss = w.k_ss, k_pp = w.k_pp;
  double k_0 = w.k_0;
#if SC_B_LEAN_LOOP
  asm volatile("" : "+v"(k_0));  // (held in a vector register: as the third operand of an fma next to k_pp it was moved there once per pair)
#endif
  double tx = 0, ty = 0, mtx = 0, mty = 0;
  // one pair: slot s (a compile-time constant) with the hash key z
  auto pair = [&](auto slot, const uint64_t z, const uint64_t mix) {
    constexpr int s = decltype(slot)::value;
    XY op, os;
    double oP;
    load(entry(s), op, os, oP);
    double rx, ry;
    pair_offset<NOISE>(w, z, s, eta, off, ox - op.x, oy - op.y, rx, ry, mix);
    const double rinv = rsqrt_nr(rx * rx + ry * ry);
    const double dot = ((sxi - os.x) * rx + (syi - os.y) * ry) * rinv;  // (s_i - s_j) . n_ij
    const double wr = fma(dot, k_ss, fma(Pi + oP, k_pp, k_0)) * rinv;
    tx += wr * rx;
    ty += wr * ry;
    if constexpr (MON) {  // the tension share on its own, next to -- not instead of -- the sums above
      const double wt = fma(dot, k_ss, fma(Pi + oP, w.dt, k_0)) * rinv;
      mtx += wt * rx;
      mty += wt * ry;
    }
  };
#if SC_B_LEAN_LOOP
  // nested slot tests (nested_slots), the hash's running key and its constants in vector registers (noise_regs)
  const NoiseRegs nr = noise_regs();
  uint64_t z = zbase;
  nested_slots<0, kMaxNbr>(Cn, [&](auto slot) {
    pair(slot, z, nr.mix);
    z += nr.gold;
  });
#else
  nested_slots_flat<0, kMaxNbr>(Cn, [&](auto slot) { pair(slot, zbase + (uint64_t) decltype(slot)::value * kGold, kMix); });
#endif
  return PairSums{tx, ty, mtx, mty};
}

// Phases 3b-4 of pass B for one particle: the sum of the neighbors' start-of-tick velocities (from `tv`,
// the (vx, vy) of the tile, or from global memory) and the per-particle epilogue.
// MON: the force monitor of the reference's HUD (force_monitor.py:13-37 wraps the six force phases of
// crate.py:110-123 and averages |dv| of each) -- `mon` receives this particle's |dv| per phase, in the order
// tension, gravity, pressure, viscosity, wall_bounce, continuous_collision.
constexpr int kMonPhases = 6;
constexpr double kNearSteps = 8.0;  // cells a particle may move per tick and still be served by its block's near-segment masks
template <bool LDS, bool MON, class Entry>
__device__ __forceinline__ void pass_b_finish(const World& w, const Tile& tl, const XY* tv, const int C, const int Cn,
                                              const int ws, const Entry entry, const XY* __restrict__ svv,
                                              const double* __restrict__ wrec,
                                              const PairSums ps, const double xi, const double yi, const double Pi,
                                              double vxi, double vyi, double& xn, double& yn, double& vxn, double& vyn,
                                              double (&mon)[kMonPhases], const unsigned near_now = ~0u) {
  auto norm2 = [](double a, double b) { return sqrt(a * a + b * b); };
  double ux = 0, uy = 0;
  auto add_velocity = [&](auto slot) {
    constexpr int s = decltype(slot)::value;
    XY ov;
    if constexpr (LDS) {
      ov = tv[entry(s)];
    } else {
      const int j = entry_index(tl, entry(s));
      ov = svv[j];
    }
    ux += ov.x;  // crate.py:175: the neighbors' start-of-tick velocities
    uy += ov.y;
  };
#if SC_B_LEAN_LOOP
  nested_slots<0, kMaxNbr>(Cn, add_velocity);
#else
  nested_slots_flat<0, kMaxNbr>(Cn, add_velocity);
#endif

  // 4. per-particle epilogue
  double Ux = 0, Uy = 0, Cx = 0, Cy = 0, V = 0;
  {
#pragma clang fp contract(fast)
  vxi += ps.tx;  // crate.py:352 and the particle part of :306
  vyi += ps.ty;
  vxi += w.dt_gx;  // crate.py:310
  vyi += w.dt_gy;
  double wallx = 0, wally = 0;
  if (ws >= 0) {
    const double* rec = wrec + 5 * (size_t)ws;
    Ux = rec[0]; Uy = rec[1]; Cx = rec[2]; Cy = rec[3]; V = rec[4];
    const double dpa = w.dt_pamp * Pi;  // wall colliders carry pressure 0 and are not normalised (crate.py:286-306)
    wallx = dpa * Ux;
    wally = dpa * Uy;
    vxi += wallx;
    vyi += wally;
  }
  const double dv = w.dt_visc;  // crate.py:319-323: sum_j (v0_j - v_i), v_i the current velocity
  const double visx = dv * (ux - C * vxi), visy = dv * (uy - C * vyi);
  vxi += visx;
  vyi += visy;
  if constexpr (MON) {
    mon[0] = norm2(ps.mtx, ps.mty);
    mon[1] = norm2(w.dt_gx, w.dt_gy);
    mon[2] = norm2(ps.tx - ps.mtx + wallx, ps.ty - ps.mty + wally);
    mon[3] = norm2(visx, visy);
    mon[4] = mon[5] = 0.0;
  }
  }
  if (ws >= 0) {  // crate.py:245-259
    // the mean over the V contacts: V is 1 or 2 nearly always, and x / 1, x / 2, x / 4 are x * 1, x * 0.5, x * 0.25 bit for
    // bit -- four float64 divisions less for a wave whose wall particles all have such a V (every wave of a pile on a wall)
    const double invV = V == 1.0 ? 1.0 : (V == 2.0 ? 0.5 : (V == 4.0 ? 0.25 : 0.0));
    double nx, ny, cvx, cvy;
    if (__ballot(invV == 0.0) == 0) {
      nx = Ux * invV; ny = Uy * invV; cvx = Cx * invV; cvy = Cy * invV;
    } else {
      nx = Ux / V; ny = Uy / V; cvx = Cx / V; cvy = Cy / V;
    }
    const double nn = sqrt(nx * nx + ny * ny);
    nx /= nn;
    ny /= nn;
    const double qq = (vxi - cvx) * nx + (vyi - cvy) * ny;
    if (qq < 0) {
      const double cx = -1 * qq * nx, cy = -1 * qq * ny;
      const double bx0 = vxi, by0 = vyi;
      vxi += cx;
      vyi += cy;
      vxi += cx * w.decay;
      vyi += cy * w.decay;
      if constexpr (MON) mon[4] = norm2(vxi - bx0, vyi - by0);
    }
  }
  // continuous collision: movement p -> p + v*dt against the 2S padded segments
  const double mx = vxi * w.dt, my = vyi * w.dt;
  if (!(ws == -1 && mx * mx + my * my < w.ccd_skip2)) {
    const double bx = xi + mx, by = yi + my;    // crate.py:183-184
    const double abx = bx - xi, aby = by - yi;  // geometry_utils.py:205 uses (b - a)
    double fac = 1.0;
    // (the lanes that are here move less than kNearSteps d -- then only the segments near the block can be crossed --
    // or the wave looks at every segment)
    const double reach = kNearSteps * w.d * (1 - 1e-6);
    const unsigned segs = __ballot(!(mx * mx + my * my < reach * reach)) ? ~0u : near_now;
    for (int mm = 0; mm < 2 * w.nseg; ++mm) {
      if (!(segs >> (mm < w.nseg ? mm : mm - w.nseg) & 1u)) continue;  // the padded twins of segment k: k and nseg + k
      const Seg s = w.pad[mm];
      const double dcx = s.bx - s.ax, dcy = s.by - s.ay;
      if (!(dcy * abx + (-dcx) * aby < 0)) continue;  // opposite_direction_map (:205)
      // orientation(p,q,r) = sign((q.y-p.y)*(r.x-q.x) - (q.x-p.x)*(r.y-q.y))  (:212-222)
      const double o1 = (by - yi) * (s.ax - bx) - (bx - xi) * (s.ay - by);          // (a,b,c)
      const double o2 = (by - yi) * (s.bx - bx) - (bx - xi) * (s.by - by);          // (a,b,d)
      const double o3 = (s.by - s.ay) * (xi - s.bx) - (s.bx - s.ax) * (yi - s.by);  // (c,d,a)
      const double o4 = (s.by - s.ay) * (bx - s.bx) - (s.bx - s.ax) * (by - s.by);  // (c,d,b)
      const int g1 = (o1 > 0) - (o1 < 0), g2 = (o2 > 0) - (o2 < 0), g3 = (o3 > 0) - (o3 < 0), g4 = (o4 > 0) - (o4 < 0);
      const bool n1 = o1 != o1, n2 = o2 != o2, n3 = o3 != o3, n4 = o4 != o4;  // np.sign(nan) = nan, nan != x
      if ((g1 != g2 || n1 || n2) && (g3 != g4 || n3 || n4)) {
        // calc_collision_point(a, ab = v*dt, c, cd): cross(a-c, cd) / cross(cd, ab)  (:141-143)
        const double acx = xi - s.ax, acy = yi - s.ay;
        const double f = (acx * dcy - acy * dcx) / (dcx * my - dcy * mx);
        if (f < fac) fac = f;  // crate.py:198-199
      }
    }
    if constexpr (MON) mon[5] = norm2(vxi * fac - vxi, vyi * fac - vyi);
    vxi *= fac;  // crate.py:200
    vyi *= fac;
  }
  xn = xi + w.dt * vxi;  // crate.py:361
  yn = yi + w.dt * vyi;
  vxn = vxi;
  vyn = vyi;
}

// FUSED: the epilogue also runs K1 of the NEXT tick (removal, wall contacts, hard wall fix, cell
// index, bucket count -- sc_kernels.h: wall_and_cell) on the freshly integrated position, with the
// next tick's walls `wn` (sc_set_next_inputs).  That tick then starts at the bucket scan: one launch
// and one read+write of the positions less per tick.
// GROUP: the fused cell count groups scrambled waves by cell (sc_kernels.h: count_cells); launched while big buckets exist
// BANDED: the instantiation the halo overlap launches with slabs of rows (a window of band blocks, parts 1 and 3 below);
// kept out of the default kernel, where its branches cost 0.8 us per tick in scalar registers
// (diagnostic build: the stamps of even and odd ticks go to different buffers, so that the last tick of a run -- which has
// no look-ahead -- does not overwrite the tick before it)
#define SC_STAMP_B(slot) SC_STAMP((w.tick & 1) ? 3 : 1, slot)
#define SC_STAMP_VALUE_B(slot, value) SC_STAMP_VALUE((w.tick & 1) ? 3 : 1, slot, value)
template <int NOISE, bool FUSED, bool MON = false, bool GROUP = false, bool BANDED = false>
__global__ void __launch_bounds__(kTileW)
    k_pass_b(World w, int* __restrict__ counters, const XY* __restrict__ sxy, const XY* __restrict__ svv,
             const int* __restrict__ id,
             const int* __restrict__ wslot, const int* __restrict__ cell, const int* __restrict__ nbr,
             const NbrRow* __restrict__ rows, int cap, const double* __restrict__ eta,
             const int* __restrict__ offById, const double* __restrict__ P, const XY* __restrict__ snn,
             const double* __restrict__ wrec, double* __restrict__ xo,
             double* __restrict__ yo, double* __restrict__ vxo, double* __restrict__ vyo, int* __restrict__ ido,
             const int* __restrict__ tileBounds, volatile int* __restrict__ progress,
             WallInputs wn, int* __restrict__ cellS, int* __restrict__ wslotS, int* __restrict__ cellCount,
             double* __restrict__ wrec_next, double* __restrict__ haloL,
             double* __restrict__ haloR, int haloCap, double* __restrict__ monitor, const int* __restrict__ tileBand,
             int part, int bandw, int epoch) {
  static_assert(!(MON && FUSED), "the force monitor runs with the plain force kernel");
  __shared__ XY txy[kTileCapB];   // (x, y) of the tile; (vx, vy) once the pair loop is done
  __shared__ XY tss[kTileCapB];   // (sx, sy)
  __shared__ double tP[kTileCapB];

  const int t = threadIdx.x;
  SC_TIMELINE_KERNEL((w.tick & 1) ? 6 : 1);
  // part 1 / 2: the blocks with / without band particles only (halo overlap: two launches, the exchange starts
  // between them); 0: all blocks.  bandw > 0 (slabs of rows: the band blocks are the first and last of the sorted
  // order): part 1 is a launch of 2 bandw workgroups over the first and the last bandw blocks -- a small kernel
  // instead of a second pass over the whole grid --, part 2 leaves exactly those band blocks out.  (Running the two
  // side by side on two streams was measured and dropped: each waits ~8 us for the other stream's event.)
  // part 3 (slabs of rows again): ONE launch.  Its first 2 bandw workgroups take the window blocks, the others the
  // blocks in between (placed by XCD as usual); every window block counts itself done, and the one that completes the
  // count publishes the launch's epoch -- k_wait_band, a one-thread kernel on the side stream, polls for it and lets
  // the exchange go while the blocks in between are still computing.  No second launch, no event between kernels.
  int tile_id;
  bool window_block = false, between = false;
  if (BANDED && (part == 1 || part == 3) && bandw > 0) {
    const int b = blockIdx.x;
    if (b < bandw) {  // the low window
      tile_id = b;
      window_block = part == 3;
    } else if (b < 2 * bandw) {  // the high window: only these blocks have to wait for the live count before anything else
      const int nt = (counters[C_NT] + kTileW - 1) / kTileW;
      tile_id = nt - 1 - (b - bandw);
      if (tile_id < bandw) return;  // fewer than 2 bandw blocks: the low window has them
      window_block = part == 3;
    } else {  // part 3: the blocks between the windows, bandw .. nt - bandw - 1 (checked once the count is here)
      const int nb = min((int)gridDim.x - 2 * bandw, max(tiles_expected(w) - 2 * bandw, 0)), ib = b - 2 * bandw;
      int k = ib;
      if (ib < nb) {
        const int q = nb >> 3, r = nb & 7, xcd = ib & 7;
        k = xcd * q + min(xcd, r) + (ib >> 3);
      }
      tile_id = bandw + k;
      between = true;
    }
  } else {
    tile_id = tile_of_block(tiles_expected(w));
  }
  const int i0 = tile_id * kTileW;
  const int i = i0 + t;
  SC_STAMP_B(0);
  // 1. one round trip: the three ranges (published by pass A for this very block), the particle's
  // scalars and its row of the table (three 16-byte loads) -- none of these loads waits for another
  const int ic = min(i, cap - 1);
  // (a block between the windows may map beyond the last block of the arrays before the live count says so)
  const int* tb = tileBounds + 6 * (BANDED ? min(tile_id, (cap - 1) / kTileW) : tile_id);
  const int tb0 = tb[0], tb1 = tb[1], tb2 = tb[2], tb3 = tb[3], tb4 = tb[4], tb5 = tb[5];
  const int cpacked = cell[ic];
  const NbrRow row = rows[ic];
  const int Craw = row_count(row);
  const int ws_raw = wslot[ic];
  const int idi = id[ic];
  const int n = counters[C_NT];
  if (tick_abandoned(counters)) return;  // (before anything is written: the storage arrays keep the state the tick started from)
  if (tile_id == 0 && t == 0 && part != 2) {
    counters[C_NS] = n;    // the storage arrays now hold the n live particles
    counters[C_SUMC] = 0;  // per-tick counters start the next tick at zero (sc_step_stats reads them
                           // between sc_step_begin and sc_step_finish, i.e. before this kernel)
    counters[C_SUMC_HI] = 0;
    counters[C_MAXC] = 0;
    counters[C_NBIG] = 0;
    counters[C_NTASKS] = 0;
    SC_TIMELINE_EPOCH(w.tick + 1);
    // host-mapped: the host keeps at most a few ticks of launches queued (progress[1]), sizes heuristics by a recent live
    // count ([2]) and keeps its bound of the ids handed out near the device's count ([3]: sc_emit_particles).  The tick
    // number goes LAST, behind a release: a reader that sees the same tick before and after reading [2] and [3] has
    // that tick's values
    progress[2] = n;
    progress[3] = counters[C_NEXT_ID];
    __hip_atomic_store(const_cast<int*>(&progress[1]), w.tick + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (BANDED && part == 3 && n == 0 && blockIdx.x == 0 && t == 0)  // nothing at all: nobody else would publish the epoch
    __hip_atomic_store(&counters[C_BAND_FLAG], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (i0 >= n) return;
  if (BANDED && between && tile_id >= (n + kTileW - 1) / kTileW - bandw) return;  // that block belongs to the high window
  bool late_block = part == 2;  // this block's halo records (if any) come too late for the message
  if (part == 1 || part == 2) {
    const int nt = (n + kTileW - 1) / kTileW;
    const bool first_launch = tileBand[tile_id] != 0 && (bandw <= 0 || tile_id < bandw || tile_id >= nt - bandw);
    if (first_launch != (part == 1)) return;
  } else if (BANDED && part == 3) {
    late_block = !window_block;
  }
  SC_STAMP_B(1);
  const int m = min(kTileW, n - i0);
  const bool live = t < m;
  // Which wall segments can a particle of this block have to do with?  The block is a run of the (row, x) order: its
  // start-of-tick positions lie in the box spanned by its first and last particle (one strip: between their x; more:
  // the full width; in y within a cell height of their y), and lane k holds segment k's distance to that box -- one pass
  // for all segments instead of a loop over them per particle.  Two masks, a bit per segment, the same in every lane:
  //   near_now   this tick's segments within r + kNearSteps d of the box: the only ones a particle that moves less than
  //              kNearSteps d can cross (pass_b_finish);
  //   near_next  the next tick's segments within far_box + kNearSteps d of the box: the only ones the look-ahead wall pass
  //              has to look at for a particle that moved less than that (19 of 20 blocks have none: the loop over the
  //              segments was most of the 2.4 us that epilogue added to a wave's 8.4).
  // kNearSteps = 8 cells: the contract workload heats up until its fastest particles cross several cells per tick; a
  // wave with one such particle used to fall back to every segment.
  // The loads are requested here; the masks are formed behind the tile's loads, before the barrier.
  const int ilast = i0 + m - 1;
  const int cell_first = cell[i0], cell_last = cell[ilast];
  const XY p_first = sxy[i0], p_last = sxy[ilast];
  const double x_first = p_first.x, x_last = p_last.x, y_first = p_first.y, y_last = p_last.y;
  const int seg_k = min(t & 63, kMaxSeg - 1);
  const Seg seg_now = w.seg[seg_k];
  Seg seg_next{0, 0, 0, 0};
  if (FUSED) seg_next = wn.seg[seg_k];

  Tile tl;
  tl.a0 = tb0;
  tl.n0 = tb1 - tb0;
  tl.a1 = tb2;
  tl.n1 = tb3 - tb2;
  tl.a2 = tb4;
  tl.n2 = tb5 - tb4;
  const int total = tl.n0 + tl.n1 + tl.n2;
  const bool in_lds = total <= kTileCapB;
  // the table entry of slot s: from the row's packed words (the pair loops keep the ten words, not twenty entries, in
  // registers); a block inside one gigantic bucket reads indices from the 32-bit table instead
  const auto entry16 = [&](int s) -> int { return row_entry(row, s); };
  const bool wide = total > kRowSlotMax;
  const auto entry_any = [&](int s) -> int { return wide ? nbr[(size_t)s * cap + ic] : row_entry(row, s); };
  SC_STAMP_VALUE_B(10, total);
  SC_STAMP_VALUE_B(11, tile_id);
  const bool ghost = w.slab && (cpacked & kGhostBit);
  const int C = live ? diag::pairs_b(Craw) : 0;
  const int Cn = ghost ? 0 : C;  // ghosts serve as neighbors only
  const int ws = live ? ws_raw : -1;

  // 2. stage the three ranges; every load of the tile is in flight before the first LDS write.  The
  // velocities wait in registers for their turn in the (x, y) array.
  constexpr int kPer = (kTileCapB + kTileW - 1) / kTileW;
  XY rv[kPer];
  const XY v0 = svv[ic];
  const double vx0 = v0.x, vy0 = v0.y;
  if (in_lds) {
    XY rp[kPer], rs[kPer];
    double rP[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int s = t + k * kTileW;
      if (s < total) {
        const int j = tile_index(tl, s);
        rp[k] = sxy[j];
        rs[k] = snn[j];
        rP[k] = P[j];
        rv[k] = svv[j];
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int s = t + k * kTileW;
      if (s < total) {
        txy[s] = rp[k];
        tss[s] = rs[k];
        tP[s] = rP[k];
      }
    }
  }
  unsigned near_now, near_next = ~0u;
  const double kMove = kNearSteps * w.d;
  {
    // one strip: the last particle's cell is less than a row of cells after the first one's and its x is not smaller
    // (in a later strip either the cell is a row further or the column -- hence x -- is smaller)
    const bool one_strip = (cell_last & kCellMask) - (cell_first & kCellMask) < w.ncols && x_last >= x_first;
    const double bx0 = one_strip ? x_first : w.lo, bx1 = one_strip ? x_last : w.hi;
    const double by0 = y_first - w.d, by1 = y_last + w.d;
    auto box_gap = [&](const Seg& sg, double& ox, double& oy) {
      ox = fmax(fmax(fmin(sg.ax, sg.bx) - bx1, bx0 - fmax(sg.ax, sg.bx)), 0.0);
      oy = fmax(fmax(fmin(sg.ay, sg.by) - by1, by0 - fmax(sg.ay, sg.by)), 0.0);
    };
    double ox, oy;
    box_gap(seg_now, ox, oy);
    // (far_box = r + 2 d, what a step of 2 d can reach: kNearSteps - 2 more cells for the steps the crossing test allows)
    const double reach_now = w.far_box + (kNearSteps - 2) * w.d;
    near_now = (unsigned)__ballot((t & 63) < w.nseg && ox <= reach_now && oy <= reach_now);
    if (FUSED) {
      box_gap(seg_next, ox, oy);
      near_next = (unsigned)__ballot((t & 63) < wn.nseg && ox <= wn.far_box + kMove && oy <= wn.far_box + kMove);
    }
  }
  __syncthreads();

  SC_STAMP_B(2);
  // 3-4. pair math and epilogue; ghosts and lanes without a particle skip it
  double mon[kMonPhases] = {0, 0, 0, 0, 0, 0};
  double xn = __builtin_huge_val(), yn = 0.0, vxn = 0.0, vyn = 0.0;  // a ghost's copy: +inf makes the next
  int idn = -1;                                                        // removal test (crate.py:152) drop it
  const bool active = live && !ghost;
  const int self = i - tl.a0;
  double xi = 0, yi = 0, Pi = 0;  // the particle's start-of-tick position and its pressure
  if (in_lds) {
    PairSums ps{0, 0, 0, 0};
    if (active) ps = pass_b_pairs<NOISE, true, MON>(w, tl, txy, tss, tP, self, Cn, idi, entry16, sxy, eta, offById, P, snn, xi, yi, Pi);
    SC_STAMP_B(3);
    __syncthreads();  // everybody is done with (x, y): the array now takes the velocities
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int s = t + k * kTileW;
      if (s < total) txy[s] = rv[k];
    }
    if constexpr (GROUP && FUSED) {  // the pressures' array is free from here on: it takes the workgroup's cell table (below)
      int* tkey = reinterpret_cast<int*>(tP);
      cell_tab_clear(tkey, tkey + kCellTabSlots);
      if (t == 0) tkey[2 * kCellTabSlots] = 0;  // waves that have added their particles
    }
    __syncthreads();
    SC_STAMP_B(4);
    if (active) {
      idn = idi;
      pass_b_finish<true, MON>(w, tl, txy, C, Cn, ws, entry16, svv, wrec, ps, xi, yi, Pi, vx0, vy0, xn, yn, vxn, vyn, mon, near_now);
    }
  } else if (active) {
    idn = idi;
    const PairSums ps = pass_b_pairs<NOISE, false, MON>(w, tl, txy, tss, tP, self, Cn, idi, entry_any, sxy, eta, offById, P, snn, xi, yi, Pi);
    pass_b_finish<false, MON>(w, tl, txy, C, Cn, ws, entry_any, svv, wrec, ps, xi, yi, Pi, vx0, vy0, xn, yn, vxn, vyn, mon, near_now);
  }
  SC_STAMP_B(5);
  if constexpr (MON) {  // sums over the wave, one atomic per wave and phase; [kMonPhases] counts the particles
    double cnt = active ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k <= kMonPhases; ++k) {
      double v = k < kMonPhases ? (active ? mon[k] : 0.0) : cnt;
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if ((t & 63) == 0 && v != 0.0) atomicAdd(&monitor[k], v);
    }
  }
  bool packed = false;  // this lane wrote a halo record
  if (FUSED) {
    int cnext = -1, wsn = -1;
    const double xp = xn, yp = yn;  // as integrated: what a halo message carries (the receiver runs its own K1)
    // the next tick's segments near this block (near_next), or all of them when some particle moved further than that
    // mask allows for
    const bool strayed = active && !(fabs(xn - xi) <= kMove && fabs(yn - yi) <= kMove);
    const unsigned segs = __ballot(strayed) ? ~0u : near_next;
    SC_STAMP_B(16);
    if (active) cnext = wall_and_cell(wn, xn, yn, wsn, counters, i, wrec_next, segs);
    SC_STAMP_B(17);
    if (live) {
      cellS[i] = cnext;
      if (cnext >= 0) wslotS[i] = wsn;
    }
    SC_STAMP_B(18);
    if (GROUP && in_lds) {
      // pile-up regime: the workgroup's particles are grouped by cell in LDS (cell_tab_*: sc_kernels.h), and the wave that
      // adds its particles last sends one atomic per cell for all four -- no barrier: a wave's LDS operations execute in
      // order, so whoever draws the last arrival number finds every other wave's entries in the table
      static_assert(sizeof(double) * kTileCapB >= sizeof(int) * (2 * kCellTabSlots + 1), "the table lives in the pressures' array");
      int* tkey = reinterpret_cast<int*>(tP);
      int* tcnt = tkey + kCellTabSlots;
      if (cnext >= 0) atomicAdd(&tcnt[cell_tab_insert(tkey, cnext & kCellMask)], 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (and the compiler keeps the arrival behind the entries)
      int last = 0;
      if ((t & 63) == 0) last = atomicAdd(&tkey[2 * kCellTabSlots], 1) == kTileW / 64 - 1;
      asm volatile("" ::: "memory");
      if (__builtin_amdgcn_readfirstlane(last)) {
        for (int s = t & 63; s < kCellTabSlots; s += 64) {
          const int k = tkey[s];
          if (k >= 0) atomicAdd(&cellCount[k], tcnt[s]);
        }
      }
    } else {
      count_cells<GROUP>(cnext, cellCount);  // every lane of the wave takes part
    }
    SC_STAMP_B(19);
    // slabs: the coming tick's halo message is packed here too (same rule and same pre-wall-fix position as
    // k_halo_pack); a workgroup-uniform branch, every lane of the wave takes part
    if (wn.slab && haloL)
      packed = halo_pack_one(active, xp, yp, vxn, vyn, idn, wn.d, wn.slab_axis, w.own_lo, w.own_hi, w.halo, w.has_left,
                             w.has_right, haloL, haloR, haloCap, counters, late_block);
  }
  SC_STAMP_B(6);
  if (live) {
    xo[i] = xn;
    yo[i] = yn;
    vxo[i] = vxn;
    vyo[i] = vyn;
    ido[i] = idn;
  }
  SC_STAMP_B(7);
  if (BANDED && window_block) {  // part 3: this window block is done; the halo records it wrote become visible device-wide
    const int wrote = __syncthreads_or(packed);
    if (t == 0) {
      // ONE release per block that wrote records, and only a release: a __threadfence() by every thread writes back
      // AND invalidates the XCD's L2 three thousand times under the blocks that are still computing (measured: the
      // kernel took twice as long)
      if (wrote) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const int nt = (n + kTileW - 1) / kTileW;
      if (atomicAdd(&counters[C_BAND_DONE], 1) + 1 == min(2 * bandw, nt)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // behind the other blocks' releases (one invalidate per launch)
        counters[C_BAND_DONE] = 0;
        __hip_atomic_store(&counters[C_BAND_FLAG], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// The side stream's wait for the window blocks of a part-3 force kernel (hipStreamWaitValue32 is not usable on this
// stack: scripts/wait_value_probe.hip).  One thread; gives up after ~50 ms and says so.
__global__ void k_wait_band(int* __restrict__ counters, int epoch) {
  const long long t0 = wall_clock64();
  // (relaxed polls: an acquire per poll invalidates this XCD's caches under the force kernel that is running)
  while (__hip_atomic_load(&counters[C_BAND_FLAG], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch < 0) {
    if (wall_clock64() - t0 > 5000000LL) {  // 100 MHz
      atomicOr(&counters[C_FLAGS], F_BAND_TIMEOUT);
      break;
    }
    __builtin_amdgcn_s_sleep(64);
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

}  // namespace sc
