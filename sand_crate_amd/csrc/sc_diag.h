// Everything of libsandcrate_hip.so that exists for measuring it, in one place; the product build compiles none of it
// (every hook below is an empty inline function or an empty macro there) and the kernels carry one-line hooks only.
//   -DSC_STAMPS      per-wave clock stamps at phase boundaries (scripts/stamp_phases*.py, pile_stamps.py)
//   -DSC_TIMELINE    every wave's start, end and hardware slot (scripts/timeline.py)
//   -DSC_ABL_*       ablations for frozen-state timing (scripts/frozen_time.py): they change the physics
//       SC_ABL_A_NOENUM   pass A without the neighbor search (empty lists)
//       SC_ABL_A_NOPAIRS  pass A without its pair math       SC_ABL_B_NOPAIRS  pass B without its pair loop
//       SC_ABL_CAPC=k     both pair loops cut at k neighbors
#pragma once
#include "sc_device.h"

namespace sc {
namespace diag {
#ifdef SC_ABL_A_NOENUM
constexpr bool kNoSearch = true;
#else
constexpr bool kNoSearch = false;
#endif
#ifdef SC_ABL_CAPC
constexpr int kCapC = SC_ABL_CAPC;
#else
constexpr int kCapC = kMaxNbr;
#endif
#ifdef SC_ABL_A_NOPAIRS
constexpr int kCapA = 0;
#else
constexpr int kCapA = kCapC;
#endif
#ifdef SC_ABL_B_NOPAIRS
constexpr int kCapB = 0;
#else
constexpr int kCapB = kCapC;
#endif
// the number of pairs the pair loops of pass A / pass B take of a list of C entries (C itself in the product build)
__device__ __forceinline__ int pairs_a(int C) { return kCapA >= kMaxNbr ? C : min(C, kCapA); }
__device__ __forceinline__ int pairs_b(int C) { return kCapB >= kMaxNbr ? C : min(C, kCapB); }
}  // namespace diag

// -DSC_STAMPS: per-wave clock stamps at phase boundaries, drained (s_waitcnt 0) so that a stamp means "everything
// before is done".  The stamps go to a buffer no kernel reads (sc_debug_stamps reads it
// on the host); the product build compiles none of this.
#ifdef SC_STAMPS
constexpr int kStampSlots = 24, kStampWaves = 1 << 16;
constexpr int kStampKernels = 6;  // 0: pass A, 1: pass B (even ticks), 2: k_sort_big, 3: pass B (odd ticks), 4: scatter, 5: reorder
__device__ long long g_stamps[kStampKernels][kStampWaves][kStampSlots];
#define SC_STAMP(kernel, slot)                                                                          \
  do {                                                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
    const long long now_ = __builtin_amdgcn_s_memtime();                                                 \
    const int wv_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                                 \
    if ((threadIdx.x & 63) == 0 && wv_ < kStampWaves) g_stamps[kernel][wv_][slot] = now_;                \
  } while (0)
#define SC_STAMP_VALUE(kernel, slot, value)                                                             \
  do {                                                                                                   \
    const int wv_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                                 \
    if ((threadIdx.x & 63) == 0 && wv_ < kStampWaves) g_stamps[kernel][wv_][slot] = (value);             \
  } while (0)
#else
#define SC_STAMP(kernel, slot) do { } while (0)
#define SC_STAMP_VALUE(kernel, slot, value) do { } while (0)
#endif

// Diagnostic build only (-DSC_TIMELINE): when and where every wave of pass A / pass B ran -- start and end on the 100 MHz
// constant clock (the same on every XCD) and the hardware slot (HW_ID: wave, SIMD, CU, SE; XCC_ID) -- so that the host can
// draw the occupancy of every CU over the kernel (scripts/timeline.py).  Two clock reads and one 32-byte store per wave.
#ifdef SC_TIMELINE
// 0: pass A, 1: pass B, 2: scan, 3: scatter, 4: reorder, 5: sort_big; 6, 7: pass B and scan of odd ticks (so that the gap from one
// tick's pass B to the next tick's scan can be read off: g_tl_epoch = the tick pass B last started, published by its tile 0)
constexpr int kTlWaves = 1 << 16, kTlKernels = 8;
__device__ int g_tl_epoch;
__device__ long long g_timeline[kTlKernels][kTlWaves][4];
struct TimelineGuard {  // records at every exit of the kernel (the destructor runs on each return path)
  int kernel;
  long long t0;
  __device__ __forceinline__ explicit TimelineGuard(int k) : kernel(k), t0(__builtin_amdgcn_s_memrealtime()) {}
  __device__ __forceinline__ ~TimelineGuard() {
    const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && wv < kTlWaves) {
      g_timeline[kernel][wv][0] = t0;
      g_timeline[kernel][wv][1] = __builtin_amdgcn_s_memrealtime();
      g_timeline[kernel][wv][2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
      g_timeline[kernel][wv][3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    }
  }
};
#define SC_TIMELINE_KERNEL(kernel) sc::TimelineGuard timeline_guard_(kernel)
#define SC_TIMELINE_SCAN() SC_TIMELINE_KERNEL((sc::g_tl_epoch & 1) ? 7 : 2)
#define SC_TIMELINE_EPOCH(tick) sc::g_tl_epoch = (tick)
#else
#define SC_TIMELINE_KERNEL(kernel) do { } while (0)
#define SC_TIMELINE_SCAN() do { } while (0)
#define SC_TIMELINE_EPOCH(tick) do { } while (0)
#endif


// k_sort_big: the largest bin a wave of a sorting task saw (its counting loop runs as long as that bin is)
#ifdef SC_STAMPS
#define SC_STAMP_LARGEST_BIN()                                                                          \
  do {                                                                                                   \
    int big_ = 0;                                                                                        \
    for (int u_ = 0; u_ < kPerT; ++u_)                                                                   \
      if (tid + u_ * kSortBlock < len) big_ = max(big_, hist[bin[u_] + 1] - hist[bin[u_]]);              \
    for (int o_ = 32; o_ > 0; o_ >>= 1) big_ = max(big_, __shfl_xor(big_, o_, 64));                      \
    SC_STAMP_VALUE(2, 9, big_);                                                                          \
  } while (0)
#else
#define SC_STAMP_LARGEST_BIN() do { } while (0)
#endif

// The windowed scans' bookkeeping (how many rounds, stagings and wave-wide turns a wave took, and where its clock went),
// written with the stamps; an empty object in the product build.
struct WindowProbe {
#ifdef SC_STAMPS
  long long rounds = 0, stagings = 0, wants = 0, turns = 0, t_round = 0, t_stage = 0, t_serial = 0, t_coop = 0, t_other = 0;
  long long last = __builtin_amdgcn_s_memtime();
  int scans = 0, hits_first = 0, hits_second = 0;
  __device__ __forceinline__ void scan_begins(int left) { wants |= (long long)__popcll(__ballot(left > 0)) << (8 * scans++); }
  __device__ __forceinline__ void round() { ++rounds; }
  __device__ __forceinline__ void staging() { ++stagings; }
  __device__ __forceinline__ void turn() { ++turns; }
  __device__ __forceinline__ void clock(long long& acc) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long now = __builtin_amdgcn_s_memtime();
    acc += now - last;
    last = now;
  }
  __device__ __forceinline__ void clock_round() { clock(t_round); }
  __device__ __forceinline__ void clock_stage() { clock(t_stage); }
  __device__ __forceinline__ void clock_serial() { clock(t_serial); }
  __device__ __forceinline__ void clock_coop() { clock(t_coop); }
  __device__ __forceinline__ void clock_other() { clock(t_other); }
  __device__ __forceinline__ static int wave_total(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  }
  __device__ __forceinline__ void hits_after_first(int C) { hits_first = wave_total(C); }
  __device__ __forceinline__ void hits_after_second(int C) {
    hits_second = wave_total(C);
    SC_STAMP_VALUE(0, 9, (long long)hits_first | ((long long)hits_second << 32));
  }
  __device__ __forceinline__ void flush() {
    SC_STAMP_VALUE(0, 12, rounds);
    SC_STAMP_VALUE(0, 13, stagings);
    SC_STAMP_VALUE(0, 14, wants);
    SC_STAMP_VALUE(0, 15, turns);
    SC_STAMP_VALUE(0, 16, t_round);
    SC_STAMP_VALUE(0, 17, t_stage);
    SC_STAMP_VALUE(0, 18, t_serial);
    SC_STAMP_VALUE(0, 19, t_coop);
    SC_STAMP_VALUE(0, 20, t_other);
  }
#else
  __device__ __forceinline__ void scan_begins(int) {}
  __device__ __forceinline__ void round() {}
  __device__ __forceinline__ void staging() {}
  __device__ __forceinline__ void turn() {}
  __device__ __forceinline__ void clock_round() {}
  __device__ __forceinline__ void clock_stage() {}
  __device__ __forceinline__ void clock_serial() {}
  __device__ __forceinline__ void clock_coop() {}
  __device__ __forceinline__ void clock_other() {}
  __device__ __forceinline__ void hits_after_first(int) {}
  __device__ __forceinline__ void hits_after_second(int) {}
  __device__ __forceinline__ void flush() {}
#endif
};

}  // namespace sc
