// Device-side types and helpers shared by the kernels of libsandcrate_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <climits>

#include "sandcrate_hip.h"

namespace sc {

#ifndef SC_TILE_W
#define SC_TILE_W 256
#endif
constexpr int kBlock = 256;          // 4 wave64 per workgroup
constexpr int kMaxNbr = SC_MAX_NEIGHBORS;
constexpr int kMaxSeg = SC_MAX_SEGMENTS;
constexpr int kMaxBody = SC_MAX_BODIES;

// Two float64 as one 16-byte vector: a tile read is a single ds_read_b128, and the sorted arrays -- (x, y), (vx, vy) and
// the surface normals (sx, sy) -- are stored as such pairs: one 16-byte request where two 8-byte ones would go (the
// fabric serves 8-byte accesses at 0.54-0.70 of the 16-byte rate).
typedef double XY __attribute__((ext_vector_type(2)));

struct Seg {
  double ax, ay, bx, by;
};

struct BodyK {
  double px, py, vx, vy, omega;
  int nseg;
  int pad_;
};

// Everything a tick's kernels need besides the particle arrays, passed BY VALUE as a kernel
// argument (about 2.3 KB of the 4 KB kernarg segment): uniform data is then read with scalar
// loads, nothing has to be staged through a device buffer, and a captured launch keeps its own
// copy.  Coefficients are live-editable in the reference (playback.py:221-226), so they are
// never compiled in.
struct World {
  // coefficients (crate.py:42-57)
  double dt, r, d, decay, pamp, ignored, level, visc, ss, tp, gx, gy;
  double inv_d;      // 1/d, for the float-tolerance math only (never for a decision)
  double eta_scale;  // d * collider_noise_level / 2^32, for the counter noise
  double eta_half;   // 2^31 * eta_scale = d * collider_noise_level / 2
  // products of the coefficients that pass B would otherwise form per workgroup (the same IEEE products, taken on the host)
  double k_ss, k_pp, k_0;              // dt ss, dt (1 + pamp), -2 tp dt
  double dt_gx, dt_gy, dt_visc, dt_pamp;
  // decision thresholds derived on the host, see sc_host.cpp: make_world()
  double t_nbr;      // largest s with sqrt(s) <= d          (collision_detector.py:78-79)
  double dsafe;      // |dx| below this and within the distance: inside the reference's x-window whatever the rounding of
                     // x +- d (d (1 - 2^-20) when every |x| / d is below 2^30, else 0: the window expression always decides)
  double t_wall;     // largest s with sqrt(s) <= r * 1.2    (crate.py:229)
  double lo, hi;     // -r, 1 + r                             (crate.py:152)
  double touch_box;  // bounding-box reject radius for wall contact
  double far_box;    // bounding-box radius beyond which no padded segment can be crossed
  double ccd_skip2;  // squared step length below which `far` particles skip the crossing test
  // cell grid: row = floor(y/d) - row0, col = floor(x/d) - col0; ring of empty cells around it
  long long row0, col0;
  double row0d, col0d;  // the same as float64 (exact: far below 2^53), for the cell index of K1
  int nrows, ncols;
  int nseg, nbody;
  int noise_mode;
  int tick;
  unsigned long long noise_key;
  // slab decomposition (single GPU: slab = 0 and everything is owned).  A particle's column / row is
  // floor(x / d) / floor(y / d) of the position it has when the tick starts.
  long long own_lo, own_hi;  // owned columns (slab_axis 0) or rows (1): [own_lo, own_hi)
  int slab, halo;            // slab mode on/off; ghost band width in columns / rows
  int slab_axis;             // 0: slabs are ranges of columns floor(x / d); 1: of rows floor(y / d)
  int band_margin;           // halo overlap: how far from the halo band a block still counts as a band block
  int live_hint;             // particles expected to be live (a recent tick's count plus slack; the launch bound when
                             // unknown): only the XCD-aware placement of blocks uses it, never a result
  int has_left, has_right;
  Seg seg[kMaxSeg];
  Seg pad[2 * kMaxSeg];
  BodyK body[kMaxBody];
};

// What K1 (removal, wall contacts, wall fix, cell index) reads of a tick's inputs: the subset of
// World that the fused epilogue of pass B needs for the NEXT tick (sc_set_next_inputs).  Field names
// match World so that one template serves both.
struct WallInputs {
  double r, d, inv_d, lo, hi, t_wall, touch_box, far_box, row0d, col0d;
  long long row0, col0, own_lo, own_hi;
  int nrows, ncols, nseg, nbody, slab, slab_axis;
  Seg seg[kMaxSeg];
  BodyK body[kMaxBody];
};

// indices into the small device-side counter block
enum Counter {
  C_NS = 0,     // particles in the storage arrays at the start of the tick
  C_NT = 1,     // live particles after removal (= entries of the sorted arrays)
  C_FLAGS = 2,  // error bits
  C_WREC = 3,   // particles with a wall record, counted on demand (sc_step_stats)
  C_SUMC = 4,   // sum of neighbor counts (low 32 bits)
  C_MAXC = 5,   // max neighbor count
  C_SUMC_HI = 6,
  C_NEXT_ID = 7,  // id the next emitted particle gets (k_rng_emit counts on the device; uploads set it)
  C_SPARE8 = 8,
  C_TICKET = 9,  // workgroups that have finished the current halo kernel (last one publishes / bumps)
  C_NBIG = 10,   // buckets above kSortThreshold listed this tick
  C_NTASKS = 11, // ... and k_sort_big's tasks for them (one 64-bit atomic with C_NBIG: keep the two adjacent, C_NBIG even)
  C_COUNT = 12,
  // on cache lines of their own, away from the counters every workgroup reads (k_wait_band polls the flag):
  C_BAND_DONE = 32,  // halo overlap in one launch: the window blocks of the force kernel that have finished
  C_BAND_FLAG = 64,  // ... and the epoch of the launch whose window blocks are all done
  C_ALLOC = 96
};

enum Flag { F_OUT_OF_GRID = 1, F_NAN = 2, F_HALO_OVERFLOW = 4, F_CAPACITY = 8, F_HALO_LATE = 16, F_BAND_TIMEOUT = 32, F_SCAN_TIMEOUT = 64 };

// columns / rows a particle may move in one tick and still be packed in time (halo overlap).  With slabs of rows the
// band blocks are few whatever the margin; with columns every row has them, and each column of margin adds as many.
constexpr int kBandMarginColumns = 2, kBandMarginRows = 8;

constexpr int kGhostBit = 1 << 30;  // set in a particle's packed cell index when it is a ghost
constexpr int kCellMask = kGhostBit - 1;
constexpr int kHaloFields = 5;      // x, y, vx, vy, id per halo record

// Counter-based collider noise: two uniforms with 32 bits each from one 64-bit hash of
// (tick key, particle id, slot).  oracle/tick.py:counter_noise_u01 is the same function.
//   z = (id * 32 + slot) * GOLD + key;  z ^= z >> 32;  z *= MIX;  z ^= z >> 32;  u_x = hi32 / 2^32, u_y = lo32 / 2^32
// The per-particle part of z is hoisted out of the pair loops (noise_base) and the slot part is a
// running add of GOLD, so one pair costs one 64-bit multiply.  The offset (u - 0.5) * d * level is
// formed as (hi32 - 2^31) * (d * level / 2^32): one signed convert and one multiply per component.
constexpr uint64_t kGold = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kMix = 0xD6E8FEB86659FD93ull;

__device__ __host__ __forceinline__ uint64_t mix64(uint64_t z) {  // used for the per-tick key only
  z ^= z >> 33;
  z *= 0xFF51AFD7ED558CCDull;
  z ^= z >> 33;
  z *= 0xC4CEB9FE1A85EC53ull;
  z ^= z >> 33;
  return z;
}

__device__ __forceinline__ uint64_t noise_base(uint64_t key, int id) {
  return ((uint64_t)(uint32_t)id * 32ull) * kGold + key;
}

__device__ __forceinline__ void noise_eta(uint64_t z, double eta_scale, double& ex, double& ey) {
  z ^= z >> 32;
  z *= kMix;
  z ^= z >> 32;
  const int hi = (int)((uint32_t)(z >> 32) ^ 0x80000000u);  // hi32 - 2^31 as a signed integer
  const int lo = (int)((uint32_t)z ^ 0x80000000u);
  ex = (double)hi * eta_scale;
  ey = (double)lo * eta_scale;
}

// Runs of equal keys among the lanes of a wave.  Storage order is the previous tick's sorted order,
// so consecutive lanes mostly fall into the same cell: one atomic per run instead of one per
// particle takes the same-address serialisation out of the cell counters (a cell that collects
// thousands of particles is hit by n/64 atomics instead of n).
struct LaneRun {
  int head;  // lane index of the first lane of this lane's run
  int len;   // number of lanes in the run
  bool is_head;
};

__device__ __forceinline__ LaneRun lane_run(int key) {
  const int lane = threadIdx.x & 63;
  const int prev = __builtin_amdgcn_update_dpp(key, key, 0x138, 0xf, 0xf, false);  // wave_shr:1 (lane 0 keeps its own)
  bool is_head = lane == 0 || key != prev;
  unsigned long long heads = __ballot(is_head);  // every lane of the wave must call this
  unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
  int head = 63 - __clzll(below);
  unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
  int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
  // lanes past the end of the data are given distinct negative keys by the callers, so `next`
  // never merges them into a real run
  LaneRun r;
  r.head = head;
  r.len = next - head;
  r.is_head = is_head;
  return r;
}

// Wave-wide minimum / maximum, the same value in every lane: four row shifts and two row broadcasts on the DPP path
// (12 vector instructions and a v_readlane) instead of six rounds through the LDS crossbar (ds_bpermute: ~100 clocks
// each for a wave on its own).  Every lane of the wave must call these.
__device__ __forceinline__ int wave_min_all(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x111, 0xf, 0xf, false));  // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x112, 0xf, 0xf, false));  // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x114, 0xf, 0xf, false));  // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x118, 0xf, 0xf, false));  // row_shr:8: lane 15 of a row holds the row
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_all(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(v, 63);
}

// Inclusive prefix sum over the lanes of a wave (lane i: v_0 + ... + v_i), on the DPP path like the reductions above.
__device__ __forceinline__ int wave_scan_add(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8: prefix sums inside every row of 16
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15: rows 1, 3 += the row before
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31: rows 2, 3 += rows 0 + 1
  return v;
}

// Lane i's copy of lane (i ^ J)'s value: on the DPP path for J below 16 (quad permutes, row shifts under bank masks,
// a row rotate), through the LDS crossbar otherwise.  Every lane of the wave must call this.
template <int J>
__device__ __forceinline__ int xor_lane(int v) {
  if constexpr (J == 1) {
    return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);  // quad_perm:[1,0,3,2]
  } else if constexpr (J == 2) {
    return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);  // quad_perm:[2,3,0,1]
  } else if constexpr (J == 4) {
    const int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);  // row_shl:4 into lanes 0-3, 8-11 of a row
    return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false);        // row_shr:4 into lanes 4-7, 12-15
  } else if constexpr (J == 8) {
    return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);  // row_ror:8
  } else {
    return __shfl_xor(v, J, 64);
  }
}
template <int J>
__device__ __forceinline__ double xor_lane(double v) {
  return __hiloint2double(xor_lane<J>(__double2hiint(v)), xor_lane<J>(__double2loint(v)));
}

__device__ __forceinline__ int wave_sum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
  return v;
}

}  // namespace sc

#include "sc_diag.h"
