// libsandcrate_hip.so -- host side of the C ABI declared in include/sandcrate_hip.h.
// Owns the device memory (float64 SoA particle arrays, cell buckets, neighbor table), builds the
// per-tick kernel argument block and enqueues the kernels of sc_kernels.h on one HIP stream.
// gfx950 (MI355X) only; there is no CPU path in this library.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "sandcrate_hip.h"
#include "sc_kernels.h"
#include "sc_rccl.h"
#include "sc_rng.h"
#include "sc_tiled.h"

using namespace sc;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(SC_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
  } while (0)

enum KernelId { K_APPEND = 0, K_WALL_BIN, K_SCAN, K_SCATTER, K_REORDER, K_NEIGHBORS, K_NOISE_OFFSETS, K_DENSITY, K_FORCE, K_HALO_PACK, K_HALO_UNPACK, K_PASS_A };
const char* kKernelNames[SC_NUM_KERNELS] = {"append",    "wall_bin",      "cell_scan", "scatter", "reorder",
                                            "neighbors", "noise_offsets", "density",   "force_integrate",
                                            "halo_pack", "halo_unpack", "neighbors_density"};

// Largest s with sqrt(s) <= R.  sqrt is correctly rounded and monotone, so for s >= 0
// (sqrt(s) <= R) == (s <= threshold): the kernels compare squared distances and skip the sqrt
// while taking exactly the reference's decision (collision_detector.py:78-79, crate.py:229).
double sq_threshold(double R) {
  if (!(R >= 0)) return -1.0;
  if (std::isinf(R)) return R;
  double t = R * R;
  const double inf = std::numeric_limits<double>::infinity();
  while (std::sqrt(t) > R) t = std::nextafter(t, -inf);
  while (std::sqrt(std::nextafter(t, inf)) <= R) t = std::nextafter(t, inf);
  return t;
}

template <class T>
hipError_t dalloc(T** p, size_t n) {
  return hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
}

}  // namespace

struct sc_ctx {
  int device = 0;
  int num_cus = 256;
  int tile_choice = 0;  // 0 = by grid size, 1 = always the narrow pass A tile, 2 = always the wide one (SANDCRATE_TILE, for tests)
  hipStream_t own_stream = nullptr, stream = nullptr;
  int64_t cap = 0;
  // particle sets: [0] storage order (input of a tick, output of pass B), [1] cell-sorted
  double *x[2] = {}, *y[2] = {}, *vx[2] = {}, *vy[2] = {};
  int* id[2] = {};
  int *cellS = nullptr, *wslotS = nullptr, *cellT = nullptr, *wslotT = nullptr;
  SortKey* keys = nullptr;  // a bucket slot's (x, id, storage index): k_scatter writes, k_sort_big sorts, k_reorder ranks
  int* keyCell = nullptr;  // the packed cell of the particle in a bucket slot (k_scatter writes it next to the key)
  int* tileBounds = nullptr;   // per block of kTileW sorted particles: its three candidate ranges (k_reorder)
  int* tileBoundsT = nullptr;  // ... the three ranges its neighbor-table slots refer to (the search; sc_tiled.h)
  int* tileBand = nullptr;  // per block of pass A / B: holds a particle that may be packed into a halo message
  // halo overlap (sc_set_halo_overlap): the exchange runs on the side stream between the two launches of pass B
  bool overlap = false, band_pending = false;
  bool band_by_flag = false;  // slabs of rows: the split force kernel is ONE launch + a polling kernel on the side stream (sc_set_band_flag)
  bool band_flagged = false;  // the pending band is announced by the flag (k_wait_band), not by ev_band
  int band_epoch = 0;
  hipEvent_t ev_band = nullptr, ev_xchg = nullptr;
  int *cellCount = nullptr, *cellStart = nullptr, *sortedStamp = nullptr;
  unsigned long long* scanDesc = nullptr;  // the bucket scan's look-back descriptors, one per 2048 cells (k_scan_cells)
  unsigned scanStamp = 0;                  // ... and the stamp of its last launch
  int scan_max_polls = kScanMaxPolls;      // ... and how often a workgroup asks for a predecessor's total before it gives up (sc_set_scan_patience)
  int2* sortTasks = nullptr;  // k_sort_big's task list (cell, chunk | length): the scan writes it
  bool piles_now = false;     // the hint "big buckets exist", latched once per tick (sc_step_begin)
  RcclComm comm = nullptr;  // RCCL communicator of the slab chain (sc_comm_init), or null
  int comm_rank = -1, comm_world = 0;
  double *haloL = nullptr, *haloR = nullptr;  // send buffers of the last sc_halo_pack (caller-owned device memory)
  int haloCap = 0;
  int64_t halo_ring_from = 0;  // first tick whose halo counts in the progress block belong to the current state
  int64_t live_hint_from = 0;  // the live count the device publishes is usable once a tick >= this one has finished
  RngState* rng = nullptr;     // NumPy's MT19937 stream on the device (sc_rng_set_state), or null
  double* monitor = nullptr;   // force monitor: sum of |dv| per phase and the particle count (sc_enable_force_monitor)
  bool monitor_on = false;
  // checkpoint (sc_checkpoint_begin / _finish): device-side snapshot, pinned host copy, side stream
  double* snap_d[4] = {};
  int* snap_id_d = nullptr;
  RngState* snap_rng_d = nullptr;
  double* snap_h[4] = {};
  int* snap_id_h = nullptr;
  int* snap_counters_h = nullptr;  // C_COUNT counters + [C_COUNT] = RngState follows in snap_rng_h
  RngState* snap_rng_h = nullptr;
  int64_t snapAlloc = 0, snap_n_bound = 0, snap_tick = -1;
  bool snap_has_rng = false, snap_pending = false;
  hipStream_t side_stream = nullptr;
  hipEvent_t snap_ready = nullptr, snap_done = nullptr;
  int* colHist = nullptr;      // sc_column_histogram
  int64_t colHistAlloc = 0;
  // host-mapped progress block written by the GPU, read by the host without synchronisation:
  // [0] big buckets seen by the last finished scan, [1] ticks finished, [2] live particles of that tick,
  // [4 + 4 (tick % kHaloRing) ..]: halo record counts of that tick (sent left / right, received left / right)
  int64_t emit_most = 0;  // the largest per-call bound of emitted particles so far (sc_emit_particles)
  int* bigHintHost = nullptr;
  int* bigHintDev = nullptr;
  bool force_rank_big = false;
  int64_t cellAlloc = 0;
  double* wrec[2] = {nullptr, nullptr};  // wall records of even / odd ticks
  int* nbr = nullptr;              // neighbor table of tiles beyond 65535 entries: -(sorted index + 1), 32 bit
  NbrRow* rows = nullptr;  // neighbor table: a 32-byte row per sorted particle (twenty 12-bit tile slots and the count)
  double* P = nullptr;
  XY *sxy = nullptr, *svv = nullptr, *snn = nullptr;  // the sorted positions and velocities, the surface normals: 16-byte pairs
  int* counters = nullptr;
  // SC_NOISE_HOST
  int *cntById = nullptr, *offById = nullptr, *idBlockSums = nullptr;
  int64_t idAlloc = 0;
  double* eta = nullptr;
  int64_t etaAlloc = 0, etaPairs = 0;
  bool offsets_pending = false;  // the offsets of this tick are left to the launch that draws the noise (k_rng_noise_small)
  // staging for uploads
  double *stage_xy = nullptr, *stage_vxy = nullptr;
  int64_t stageAlloc = 0;

  sc_params params{};
  bool have_params = false;
  int nseg = 0, nbody = 0;
  Seg seg[kMaxSeg]{};
  Seg pad[2 * kMaxSeg]{};
  BodyK body[kMaxBody]{};
  int noise_mode = SC_NOISE_NONE;
  uint64_t seed = 0;
  int64_t tick = 0;
  int64_t upper = 0;    // host-side upper bound of the stored particle count
  int64_t next_id = 0;
  bool in_step = false;
  int64_t normals_valid = 0;
  bool custom_grid = false;  // sc_neighbor_search: grid from the data, no walls, no removal
  long long grid_row0 = 0, grid_col0 = 0;
  int grid_nrows = 0, grid_ncols = 0;
  double custom_d = 0;
  bool slab = false;
  long long own_lo = 0, own_hi = 0;
  int slab_axis = 0;  // 0: slabs of columns (x), 1: of rows (y)
  int halo = 0, has_left = 0, has_right = 0;
  int* stage_ids = nullptr;
  std::vector<int> ids_host;
  int64_t stats_live = -1;  // live count read by sc_step_stats inside the current tick, or -1
  int* owned_out = nullptr;
  World w{};
  // sc_set_next_inputs: the promised inputs of the tick after the current one
  bool have_next = false;
  sc_params next_params{};
  int next_nseg = 0, next_nbody = 0;
  Seg next_seg[kMaxSeg]{};
  BodyK next_body[kMaxBody]{};
  bool prebinned = false;     // the last sc_step_finish already ran K1 of the coming tick ...
  WallInputs promised{};      // ... with these inputs

  bool timing = false;
  struct Ev {
    hipEvent_t a, b;
    int k;
  };
  std::vector<Ev> ev_used, ev_free;
  double ms[SC_NUM_KERNELS] = {};
  int64_t launches[SC_NUM_KERNELS] = {};
};

namespace {

struct Bracket {  // two HIP events around a launch when timing is on
  sc_ctx* c;
  sc_ctx::Ev ev{};
  bool on;
  Bracket(sc_ctx* ctx, int k) : c(ctx), on(ctx->timing) {
    if (!on) return;
    if (!c->ev_free.empty()) {
      ev = c->ev_free.back();
      c->ev_free.pop_back();
    } else if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) {
      on = false;
      return;
    }
    ev.k = k;
    (void)hipEventRecord(ev.a, c->stream);
  }
  ~Bracket() {
    if (!on) return;
    (void)hipEventRecord(ev.b, c->stream);
    c->ev_used.push_back(ev);
  }
};

int grid_for(int64_t n) { return (int)std::max<int64_t>(1, (n + kBlock - 1) / kBlock); }

// In slab mode the stored count changes on the device every tick (halo records arrive without the
// host knowing how many), so launches cover the capacity; surplus workgroups exit on their first load.
int64_t launch_bound(const sc_ctx* c);

int ensure_cells(sc_ctx* c, int64_t ncells) {
  if (ncells + 1 <= c->cellAlloc) return SC_OK;
  if (ncells > (int64_t)1 << 28) return fail(SC_ERR_CAPACITY, "cell grid of %lld cells is too large", (long long)ncells);
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->cellCount) (void)hipFree(c->cellCount);
  if (c->cellStart) (void)hipFree(c->cellStart);
  if (c->scanDesc) (void)hipFree(c->scanDesc);
  if (c->sortedStamp) (void)hipFree(c->sortedStamp);
  int64_t n = ncells + 1 + ncells / 4;
  HIPCHK(dalloc(&c->cellCount, n));
  HIPCHK(dalloc(&c->cellStart, n + 1));
  HIPCHK(dalloc(&c->scanDesc, n / kScanPerBlock + 4));
  HIPCHK(hipMemsetAsync(c->scanDesc, 0, (n / kScanPerBlock + 4) * sizeof(unsigned long long), c->stream));  // stamp 0: never launched
  HIPCHK(dalloc(&c->sortedStamp, n));
  HIPCHK(hipMemsetAsync(c->sortedStamp, 0, n * sizeof(int), c->stream));
  HIPCHK(hipMemsetAsync(c->cellCount, 0, n * sizeof(int), c->stream));
  c->cellAlloc = n;
  return SC_OK;
}

int ensure_ids(sc_ctx* c, int64_t n) {
  if (n <= c->idAlloc) return SC_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->cntById) (void)hipFree(c->cntById);
  if (c->offById) (void)hipFree(c->offById);
  if (c->idBlockSums) (void)hipFree(c->idBlockSums);
  int64_t m = n + n / 2 + 1024;
  HIPCHK(dalloc(&c->cntById, m));
  HIPCHK(dalloc(&c->offById, m + 1));
  HIPCHK(dalloc(&c->idBlockSums, m / kScanPerBlock + 2));
  c->idAlloc = m;
  return SC_OK;
}

int ensure_stage(sc_ctx* c, int64_t n) {
  if (n <= c->stageAlloc) return SC_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->stage_xy) (void)hipFree(c->stage_xy);
  if (c->stage_vxy) (void)hipFree(c->stage_vxy);
  if (c->stage_ids) (void)hipFree(c->stage_ids);
  int64_t m = n + n / 2 + 256;
  HIPCHK(dalloc(&c->stage_xy, 2 * m));
  HIPCHK(dalloc(&c->stage_vxy, 2 * m));
  HIPCHK(dalloc(&c->stage_ids, m));
  c->stageAlloc = m;
  return SC_OK;
}

// exclusive scan of in[0..n) into out[0..n], out[n] = total (also to *total_out if given)
int launch_scan(sc_ctx* c, const int* in, int* out, int64_t n, int* blockSums, int* total_out) {
  int nb = (int)((n + kScanPerBlock - 1) / kScanPerBlock);
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(kBlock), 0, c->stream, in, out, (int)n, blockSums);
  hipLaunchKernelGGL(k_scan_fix, dim3(nb), dim3(kBlock), 0, c->stream, out, (int)n, blockSums, nb, total_out);
  HIPCHK(hipGetLastError());
  return SC_OK;
}

// Kernel-argument block of this tick.  The cell grid covers [-r, 1+r]^2 -- where
// remove_particles (crate.py:152) leaves particles -- plus three cells of margin for the hard wall
// fix, plus a ring of always-empty cells so that c-1 / c+1 / c+-ncols never leave the arrays.
int build_world(sc_ctx* c, World& w, const sc_params& p, int nseg, const Seg* seg, const Seg* pad, int nbody,
                const BodyK* body, int64_t tick) {
  std::memset(&w, 0, sizeof w);
  const double inf = std::numeric_limits<double>::infinity();
  if (c->custom_grid) {
    w.d = c->custom_d;
    w.r = w.d / 2;
    w.lo = -inf;
    w.hi = inf;
    w.row0 = c->grid_row0;
    w.col0 = c->grid_col0;
    w.nrows = c->grid_nrows;
    w.ncols = c->grid_ncols;
    w.t_nbr = sq_threshold(w.d);
    w.t_wall = -1.0;
    w.far_box = w.touch_box = -1.0;
    w.ccd_skip2 = inf;
  } else {
    if (!(p.particle_radius > 0) || !std::isfinite(p.particle_radius))
      return fail(SC_ERR_ARG, "particle_radius must be positive and finite");
    w.dt = p.dt;
    w.r = p.particle_radius;
    w.d = p.particle_radius * 2;  // crate.py:65-67
    w.decay = p.wall_collision_decay;
    w.pamp = p.pressure_amplifier;
    w.ignored = p.ignored_pressure;
    w.level = p.collider_noise_level;
    w.visc = p.viscosity;
    w.ss = p.surface_smoothing;
    w.tp = p.target_pressure;
    w.gx = p.gravity_x;
    w.gy = p.gravity_y;
    w.lo = -w.r;    // crate.py:152
    w.hi = 1 + w.r;
    w.t_nbr = sq_threshold(w.d);
    double r12 = w.r * 1.2;  // crate.py:229
    w.t_wall = sq_threshold(r12);
    w.touch_box = r12 * (1 + 1e-6) + 1e-12;
    w.far_box = (w.r + 2 * w.d) * (1 + 1e-6) + 1e-12;
    w.ccd_skip2 = (2 * w.d) * (2 * w.d) * (1 - 1e-6);
    long long cmin = (long long)std::floor(w.lo / w.d) - 3;
    long long cmax = (long long)std::floor(w.hi / w.d) + 3;
    // slabs keep a local grid: the slab, its ghost band, one column / row of slack for the wall fix
    long long ccmin = cmin, ccmax = cmax, rrmin = cmin, rrmax = cmax;
    if (c->slab) {
      long long& lo = c->slab_axis ? rrmin : ccmin;
      long long& hi = c->slab_axis ? rrmax : ccmax;
      lo = std::max(cmin, c->own_lo - c->halo - 1);
      hi = std::min(cmax, c->own_hi + c->halo);
      if (hi < lo) hi = lo;
    }
    w.row0 = rrmin - 1;
    w.nrows = (int)(rrmax - rrmin + 1) + 2;
    w.col0 = ccmin - 1;
    w.ncols = (int)(ccmax - ccmin + 1) + 2;
  }
  w.inv_d = 1.0 / w.d;
  // With dx = fl(x_j - x_i), |dx| < d (1 - 2^-20) puts the true difference below d (1 - 2^-21); fl(x +- d) is off by at most
  // |x +- d| 2^-53 <= d 2^-21 as long as |x| / d < 2^32: then x_j is inside [fl(x_i - d), fl(x_i + d)] and x_i inside
  // [fl(x_j - d), fl(x_j + d)] -- both forms of the reference's window (collision_detector.py:106-119, :85-88) hold.
  {
    const long long far = std::max(std::llabs(w.col0), std::llabs(w.col0 + w.ncols)) + 2;
    const long long far_r = std::max(std::llabs(w.row0), std::llabs(w.row0 + w.nrows)) + 2;
    w.dsafe = std::max(far, far_r) < (1LL << 30) ? w.d * (1.0 - 0x1p-20) : 0.0;
  }
  w.row0d = (double)w.row0;
  w.col0d = (double)w.col0;
  w.eta_scale = (w.d * w.level) * (1.0 / 4294967296.0);
  w.eta_half = (w.d * w.level) * 0.5;
  w.k_ss = w.dt * w.ss;
  w.k_pp = w.dt * (1 + w.pamp);
  w.k_0 = -2 * w.tp * w.dt;
  w.dt_gx = w.dt * w.gx;
  w.dt_gy = w.dt * w.gy;
  w.dt_visc = w.dt * w.visc;
  w.dt_pamp = w.dt * w.pamp;
  w.nseg = nseg;
  w.nbody = nbody;
  std::memcpy(w.seg, seg, sizeof w.seg);
  if (pad) std::memcpy(w.pad, pad, sizeof w.pad);
  std::memcpy(w.body, body, sizeof w.body);
  w.noise_mode = c->noise_mode;
  w.tick = (int)tick;
  w.noise_key = mix64(c->seed + (uint64_t)(tick + 1) * kGold);
  w.slab = c->slab ? 1 : 0;
  w.slab_axis = c->slab ? c->slab_axis : 0;
  w.band_margin = w.slab_axis ? kBandMarginRows : kBandMarginColumns;
  w.own_lo = c->slab ? c->own_lo : std::numeric_limits<long long>::min();
  w.own_hi = c->slab ? c->own_hi : std::numeric_limits<long long>::max();
  w.halo = c->halo;
  {  // where the particles are expected to end: for slabs a recent tick's live count (blocks beyond it are placed one by one)
    const int64_t done = *(volatile int*)(c->bigHintHost + 1), published = *(volatile int*)(c->bigHintHost + 2);
    const int64_t bound = launch_bound(c);
    w.live_hint = (int)(c->slab && published > 0 && done > c->live_hint_from
                            ? std::min<int64_t>(bound, (int64_t)published + 2048)
                            : bound);
  }
  w.has_left = c->has_left;
  w.has_right = c->has_right;
  return SC_OK;
}

int make_world(sc_ctx* c) {
  if (!c->custom_grid && !c->have_params) return fail(SC_ERR_STATE, "sc_set_params has not been called");
  int rc = build_world(c, c->w, c->params, c->nseg, c->seg, c->pad, c->nbody, c->body, c->tick);
  if (rc) return rc;
  return ensure_cells(c, (int64_t)c->w.nrows * c->w.ncols);
}

// the part of a tick's inputs that K1 reads (see WallInputs)
WallInputs wall_inputs_of(const World& w) {
  WallInputs k;
  std::memset(&k, 0, sizeof k);
  k.r = w.r; k.d = w.d; k.inv_d = w.inv_d; k.lo = w.lo; k.hi = w.hi; k.t_wall = w.t_wall; k.touch_box = w.touch_box; k.far_box = w.far_box;
  k.row0 = w.row0; k.col0 = w.col0; k.row0d = w.row0d; k.col0d = w.col0d; k.own_lo = w.own_lo; k.own_hi = w.own_hi;
  k.nrows = w.nrows; k.ncols = w.ncols; k.nseg = w.nseg; k.nbody = w.nbody; k.slab = w.slab; k.slab_axis = w.slab_axis;
  std::memcpy(k.seg, w.seg, sizeof k.seg);
  std::memcpy(k.body, w.body, sizeof k.body);
  return k;
}


int read_counters(sc_ctx* c, int* out) {
  HIPCHK(hipMemcpyAsync(out, c->counters, C_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return SC_OK;
}

int check_flags(int flags) {
  if (flags & F_NAN)
    return fail(SC_ERR_DOMAIN, "a particle position became NaN (zero distance to a wall, crate.py:206); it was dropped");
  if (flags & F_OUT_OF_GRID) return fail(SC_ERR_DOMAIN, "a particle left the cell grid; it was dropped");
  if (flags & F_HALO_OVERFLOW) return fail(SC_ERR_CAPACITY, "a halo buffer was too small; ghost particles were lost");
  if (flags & F_CAPACITY) return fail(SC_ERR_CAPACITY, "received halo particles exceed the context capacity");
  if (flags & F_BAND_TIMEOUT)
    return fail(SC_ERR_HIP, "the halo exchange waited 50 ms for the band blocks of the force kernel and gave up");
  if (flags & F_SCAN_TIMEOUT)
    return fail(SC_ERR_HIP, "the bucket scan waited for a workgroup that never published its total and gave up; the tick was "
                "skipped (the particles are as the tick found them)");
  if (flags & F_HALO_LATE)
    return fail(SC_ERR_DOMAIN, "a particle moved more than the band margin (%d columns / %d rows) in one tick and missed the "
                "overlapped halo message: run without halo overlap", kBandMarginColumns, kBandMarginRows);
  return SC_OK;
}

int put_particles(sc_ctx* c, const double* xy, const double* vxy, int64_t n, bool reset, const int64_t* ids = nullptr) {
  if (n < 0 || (n > 0 && (!xy || !vxy))) return fail(SC_ERR_ARG, "bad particle arrays");
  if (c->in_step) return fail(SC_ERR_STATE, "particles cannot change between sc_step_begin and sc_step_finish");
  if (c->prebinned && !reset)
    return fail(SC_ERR_STATE, "particles cannot be appended after sc_set_next_inputs promised the next tick");
  if (c->prebinned && reset) {  // the promised tick is abandoned: forget its bucket counts
    HIPCHK(hipMemsetAsync(c->cellCount, 0, c->cellAlloc * sizeof(int), c->stream));
    c->prebinned = false;
  }
  int64_t base = reset ? 0 : c->upper;
  if (base + n > c->cap)
    return fail(SC_ERR_CAPACITY, "%lld particles exceed the context capacity %lld", (long long)(base + n), (long long)c->cap);
  if (reset) {
    c->next_id = 0;
    c->normals_valid = 0;
    c->halo_ring_from = c->tick;  // counts published before this belong to another state
  }
  if (c->next_id + n > std::numeric_limits<int>::max()) return fail(SC_ERR_CAPACITY, "particle ids exhausted");
  if (n > 0) {
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->stage_xy, xy, 2 * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->stage_vxy, vxy, 2 * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int* dev_ids = nullptr;
    int64_t max_id = -1;
    std::vector<int>& ids32 = c->ids_host;  // outlives the asynchronous copy below
    if (ids) {
      HIPCHK(hipStreamSynchronize(c->stream));  // an earlier copy out of ids_host has finished
      ids32.resize(n);
      for (int64_t k = 0; k < n; ++k) {
        if (ids[k] < 0 || ids[k] > std::numeric_limits<int>::max() - 1) return fail(SC_ERR_ARG, "particle id out of range");
        ids32[k] = (int)ids[k];
        max_id = std::max<int64_t>(max_id, ids[k]);
      }
      dev_ids = c->stage_ids;  // its own allocation of stageAlloc entries (ensure_stage)
      HIPCHK(hipMemcpyAsync(dev_ids, ids32.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
      c->next_id = std::max<int64_t>(c->next_id, max_id + 1 - n);
    }
    Bracket br(c, K_APPEND);
    hipLaunchKernelGGL(k_append, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, c->stage_xy, c->stage_vxy, (int)n,
                       (int)c->next_id, dev_ids, c->counters, c->x[0], c->y[0], c->vx[0], c->vy[0], c->id[0], reset ? 1 : 0, (int)c->cap);
  }
  c->upper = base + n;
  c->next_id += n;
  c->live_hint_from = c->tick;  // counts published by earlier ticks do not include these particles
  hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, c->stream, c->counters, (int)n, reset ? 1 : 0, (int)c->next_id, (int)c->cap);
  HIPCHK(hipGetLastError());
  return SC_OK;
}

int64_t launch_bound(const sc_ctx* c) { return c->slab ? c->cap : c->upper; }

int tile_grid(const sc_ctx* c) { return (int)std::max<int64_t>(1, (launch_bound(c) + kTileW - 1) / kTileW); }

// neighbor search (+ pass A unless the host's noise block has to be indexed first)
bool piles_expected(const sc_ctx* c);

template <int NOISE, bool ENUM, bool DENS, int CAP>
void launch_pass_a_cap(sc_ctx* c) {
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(tile_grid(c)), dim3(kTileW), 0, c->stream, c->w, c->counters,
                       c->sxy, c->id[1], c->cellT, Buckets{c->cellStart}, c->nbr, c->rows, (int)c->cap, c->eta, c->offById,
                       c->P, c->snn, ENUM ? c->tileBounds : c->tileBoundsT, c->tileBand, c->tileBoundsT);
  };
  if (ENUM && DENS && piles_expected(c))  // dense tiles ahead: the instantiation that stages their lists' reach
    launch(k_pass_a<NOISE, ENUM, DENS, CAP, ENUM && DENS>);
  else
    launch(k_pass_a<NOISE, ENUM, DENS, CAP, false>);
}

template <int NOISE, bool ENUM, bool DENS>
void launch_pass_a(sc_ctx* c, int kernel_id) {
  Bracket br(c, kernel_id);
  // up to 16 waves per CU: all resident with the wide tile too; beyond that the narrow tile's higher
  // occupancy wins (measured with 128-wide tiles: 262,144 particles 35.9 -> 32.2 us wide; 1,048,576: 78 us
  // narrow, 82 us wide)
  // (slabs size their grids by capacity; the live count a recent tick published is the better estimate of the work)
  const int published = *(volatile int*)(c->bigHintHost + 2);
  const int tiles = c->slab && published > 0 ? (published + kTileW - 1) / kTileW + 64 : tile_grid(c);
  if (c->tile_choice ? c->tile_choice == 2 : tiles <= (8 * 128 / kTileW) * c->num_cus)
    launch_pass_a_cap<NOISE, ENUM, DENS, kTileCapAWide>(c);
  else
    launch_pass_a_cap<NOISE, ENUM, DENS, kTileCapA>(c);
}

// Big buckets were seen by the last scan the host knows about (an unsynchronised, possibly stale hint in host-mapped
// memory): the tick sorts its big buckets before ranking them and its cell counts group scrambled waves by cell.
// Both only cost time when they are wrong; results do not depend on the choice.
// (latched by sc_step_begin: every launch of a tick sees the same answer -- the sort of the big buckets and the grouping
// variants of scatter, search and force kernel go together)
bool piles_expected(const sc_ctx* c) { return c->piles_now; }

template <int NOISE, bool FUSED, bool MON = false>
void launch_pass_b(sc_ctx* c, const WallInputs& wn, int part = 0) {
  const int cur = (int)(c->tick & 1), nxt = cur ^ 1;
  // slabs of rows: the band blocks lie within (ghost rows + halo + margin) rows of either end of the sorted order; the
  // window takes twice the blocks those rows hold on average (a band block outside it is handled by part 2 and, should
  // it have anything to pack, reported like a particle that was too fast)
  int bandw = 0;
  if (part && c->slab_axis == 1) {
    const int64_t rows = std::max<int64_t>(1, std::min<int64_t>(c->own_hi, c->w.row0 + c->w.nrows) - std::max<int64_t>(c->own_lo, c->w.row0));
    const int64_t band_rows = 2 * c->halo + kBandMarginRows + 2;
    bandw = (int)std::min<int64_t>(tile_grid(c), 2 * band_rows * (c->w.live_hint / rows + 1) / kTileW + 16);
  }
  const int grid = part == 1 && bandw ? 2 * bandw : part == 3 ? tile_grid(c) + 2 * bandw : tile_grid(c);
  hipStream_t stream = c->stream;
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kTileW), 0, stream, c->w, c->counters, c->sxy, c->svv,
                       c->id[1], c->wslotT, c->cellT, c->nbr, c->rows, (int)c->cap, c->eta, c->offById, c->P,
                       c->snn, c->wrec[cur], c->x[0], c->y[0], c->vx[0], c->vy[0], c->id[0], c->tileBoundsT,
                       c->bigHintDev, wn, c->cellS, c->wslotS, c->cellCount, c->wrec[nxt], c->haloL, c->haloR, c->haloCap,
                       c->monitor, c->tileBand, part, bandw, c->band_epoch);
  };
  auto pick = [&]() {
    const bool group = FUSED && piles_expected(c);
    if (FUSED && bandw > 0) {  // the instantiation with the band window (slabs of rows, halo overlap)
      if (group)
        launch(k_pass_b<NOISE, FUSED, MON, FUSED, FUSED>);
      else
        launch(k_pass_b<NOISE, FUSED, MON, false, FUSED>);
    } else if (group) {
      launch(k_pass_b<NOISE, FUSED, MON, FUSED>);
    } else {
      launch(k_pass_b<NOISE, FUSED, MON, false>);
    }
  };
  Bracket br(c, K_FORCE);
  pick();
}

template <int NOISE>
void launch_pass_b_any(sc_ctx* c, bool fused, const WallInputs& wn) {
  if (c->monitor_on) {
    launch_pass_b<NOISE, false, true>(c, wn);
  } else if (fused && c->slab && c->overlap && c->haloL && (c->has_left || c->has_right)) {
    // halo overlap: the blocks that may pack halo records first; once they are done (ev_band) the exchange of the
    // coming tick may start on the side stream while the interior blocks run
    if (c->slab_axis == 1 && c->band_by_flag) {
      // slabs of rows: one launch, the window blocks first; the side stream polls for their completion (k_wait_band)
      c->band_epoch += 1;
      launch_pass_b<NOISE, true>(c, wn, 3);
      c->band_flagged = true;
    } else {
      launch_pass_b<NOISE, true>(c, wn, 1);
      (void)hipEventRecord(c->ev_band, c->stream);
      launch_pass_b<NOISE, true>(c, wn, 2);
      c->band_flagged = false;
    }
    c->band_pending = true;
  } else if (fused)
    launch_pass_b<NOISE, true>(c, wn);
  else
    launch_pass_b<NOISE, false>(c, wn);
}

}  // namespace

extern "C" {

const char* sc_last_error(void) { return g_err.c_str(); }
int sc_abi_version(void) { return SC_ABI_VERSION; }
const char* sc_kernel_name(int i) { return (i >= 0 && i < SC_NUM_KERNELS) ? kKernelNames[i] : ""; }

int sc_create(int device, int64_t capacity, sc_ctx** out) {
  if (!out || capacity < 1 || capacity > (int64_t)100000000) return fail(SC_ERR_ARG, "bad capacity");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(SC_ERR_ARG, "device %d of %d", device, ndev);
  HIPCHK(hipSetDevice(device));
  sc_ctx* c = new sc_ctx();
  c->device = device;
  c->cap = capacity;
  if (hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->num_cus < 1)
    c->num_cus = 256;
  if (const char* tile = std::getenv("SANDCRATE_TILE"))
    c->tile_choice = !std::strcmp(tile, "narrow") ? 1 : !std::strcmp(tile, "wide") ? 2 : 0;
  size_t n = (size_t)capacity;
  hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  c->stream = c->own_stream;
  // [0]: the storage set (a tick's input and output); [1]: the sorted set, of which only the ids are an array of their
  // own -- positions and velocities are the pairs sxy / svv
  if (e == hipSuccess) e = dalloc(&c->x[0], n);
  if (e == hipSuccess) e = dalloc(&c->y[0], n);
  if (e == hipSuccess) e = dalloc(&c->vx[0], n);
  if (e == hipSuccess) e = dalloc(&c->vy[0], n);
  for (int s = 0; s < 2 && e == hipSuccess; ++s) e = dalloc(&c->id[s], n);
  if (e == hipSuccess) e = dalloc(&c->cellS, n);
  if (e == hipSuccess) e = dalloc(&c->wslotS, n);
  if (e == hipSuccess) e = dalloc(&c->cellT, n);
  if (e == hipSuccess) e = dalloc(&c->wslotT, n);
  if (e == hipSuccess) e = dalloc(&c->keys, n);
  if (e == hipSuccess) e = dalloc(&c->keyCell, n);
  if (e == hipSuccess) e = dalloc(&c->tileBounds, 6 * (n / kTileW + 2));
  if (e == hipSuccess) e = dalloc(&c->tileBoundsT, 6 * (n / kTileW + 2));
  if (e == hipSuccess) e = dalloc(&c->tileBand, n / kTileW + 2);
  if (e == hipSuccess) e = hipMemsetAsync(c->tileBand, 0, (n / kTileW + 2) * sizeof(int), c->stream);
  if (e == hipSuccess) e = dalloc(&c->sortTasks, (size_t)kMaxSortTasks);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->bigHintHost, kProgressInts * sizeof(int), hipHostMallocMapped);
  if (e == hipSuccess) {
    for (int k = 0; k < kProgressInts; ++k) c->bigHintHost[k] = 0;
    e = hipHostGetDevicePointer((void**)&c->bigHintDev, c->bigHintHost, 0);
  }
  if (e == hipSuccess) e = dalloc(&c->wrec[0], 5 * n);
  if (e == hipSuccess) e = dalloc(&c->wrec[1], 5 * n);
  if (e == hipSuccess) e = dalloc(&c->nbr, (size_t)kMaxNbr * n);
  if (e == hipSuccess) e = dalloc(&c->rows, n);
  if (e == hipSuccess) e = dalloc(&c->P, n);
  if (e == hipSuccess) e = dalloc(&c->snn, n);
  if (e == hipSuccess) e = dalloc(&c->sxy, n);
  if (e == hipSuccess) e = dalloc(&c->svv, n);
  if (e == hipSuccess) e = dalloc(&c->counters, (size_t)C_ALLOC);
  if (e == hipSuccess) e = hipMemsetAsync(c->counters, 0, C_ALLOC * sizeof(int), c->stream);
  if (e != hipSuccess) {
    int rc = fail(SC_ERR_HIP, "sc_create: %s", hipGetErrorString(e));
    sc_destroy(c);
    return rc;
  }
  *out = c;
  return SC_OK;
}

int sc_destroy(sc_ctx* c) {
  if (!c) return SC_OK;
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  if (c->comm) (void)sc_comm_destroy(c);
  for (int s = 0; s < 2; ++s) {
    (void)hipFree(c->x[s]);
    (void)hipFree(c->y[s]);
    (void)hipFree(c->vx[s]);
    (void)hipFree(c->vy[s]);
    (void)hipFree(c->id[s]);
  }
  if (c->ev_band) (void)hipEventDestroy(c->ev_band);
  if (c->ev_xchg) (void)hipEventDestroy(c->ev_xchg);
  void* ptrs[] = {c->cellS, c->wslotS, c->cellT, c->wslotT, c->keys, c->keyCell, c->tileBounds, c->tileBoundsT, c->tileBand, c->cellCount, c->cellStart, c->scanDesc, c->sortedStamp, c->sortTasks, c->wrec[0], c->wrec[1],
                  c->nbr, c->rows, c->P, c->snn, c->sxy, c->svv, c->counters, c->cntById, c->offById, c->idBlockSums, c->eta,
                  c->stage_xy, c->stage_vxy, c->stage_ids, c->owned_out, c->colHist, c->rng, c->monitor,
                  c->snap_d[0], c->snap_d[1], c->snap_d[2], c->snap_d[3], c->snap_id_d, c->snap_rng_d};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& v : {c->ev_used, c->ev_free})
    for (auto& e : v) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
  for (void* p : {(void*)c->snap_h[0], (void*)c->snap_h[1], (void*)c->snap_h[2], (void*)c->snap_h[3], (void*)c->snap_id_h,
                  (void*)c->snap_counters_h, (void*)c->snap_rng_h})
    if (p) (void)hipHostFree(p);
  if (c->snap_ready) (void)hipEventDestroy(c->snap_ready);
  if (c->snap_done) (void)hipEventDestroy(c->snap_done);
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  if (c->bigHintHost) (void)hipHostFree(c->bigHintHost);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return SC_OK;
}

int sc_set_stream(sc_ctx* c, void* s) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  HIPCHK(hipStreamSynchronize(c->stream));
  c->stream = (hipStream_t)s;
  return SC_OK;
}

int sc_use_own_stream(sc_ctx* c) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  HIPCHK(hipStreamSynchronize(c->stream));
  c->stream = c->own_stream;
  return SC_OK;
}

int sc_upload_state(sc_ctx* c, const double* xy, const double* vxy, int64_t n) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  HIPCHK(hipSetDevice(c->device));
  return put_particles(c, xy, vxy, n, true);
}

int sc_append_particles(sc_ctx* c, const double* xy, const double* vxy, int64_t n) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  HIPCHK(hipSetDevice(c->device));
  return put_particles(c, xy, vxy, n, false);
}

int sc_synchronize(sc_ctx* c) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  int h[C_COUNT];
  int rc = read_counters(c, h);
  if (rc) return rc;
  if (!c->in_step) c->upper = h[C_NS];
  if (h[C_FLAGS]) {
    HIPCHK(hipMemsetAsync(c->counters + C_FLAGS, 0, sizeof(int), c->stream));
    if (h[C_FLAGS] & F_SCAN_TIMEOUT) {
      // the tick was abandoned behind its scan: its bucket counts were never consumed, and whatever a look-ahead
      // promised for the tick after it was never computed -- the next tick starts from the storage arrays
      HIPCHK(hipMemsetAsync(c->cellCount, 0, c->cellAlloc * sizeof(int), c->stream));
      c->prebinned = false;
    }
    return check_flags(h[C_FLAGS]);
  }
  return SC_OK;
}

int sc_set_scan_patience(sc_ctx* c, int64_t polls) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  c->scan_max_polls = polls < 0 ? -1 : (int)std::min<int64_t>(polls, kScanMaxPolls);
  return SC_OK;
}

int sc_count(sc_ctx* c, int64_t* n) {
  if (!c || !n) return fail(SC_ERR_ARG, "null argument");
  int h[C_COUNT];
  int rc = read_counters(c, h);
  if (rc) return rc;
  *n = c->in_step ? h[C_NT] : h[C_NS];
  if (!c->in_step) c->upper = h[C_NS];
  return SC_OK;
}

int sc_set_params(sc_ctx* c, const sc_params* p) {
  if (!c || !p) return fail(SC_ERR_ARG, "null argument");
  if (c->in_step) return fail(SC_ERR_STATE, "coefficients cannot change inside a tick");
  c->params = *p;
  c->have_params = true;
  return SC_OK;
}

int sc_set_segments(sc_ctx* c, const double* segments, const double* padded, int32_t ns, const sc_body* bodies,
                    int32_t nb) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "segments cannot change inside a tick");
  if (ns < 0 || ns > kMaxSeg) return fail(SC_ERR_CAPACITY, "%d segments, at most %d", ns, kMaxSeg);
  if (nb < 0 || nb > kMaxBody) return fail(SC_ERR_CAPACITY, "%d bodies, at most %d", nb, kMaxBody);
  if (ns > 0 && (!segments || !padded)) return fail(SC_ERR_ARG, "null segment arrays");
  int total = 0;
  for (int b = 0; b < nb; ++b) total += bodies[b].n_segments;
  if (nb > 0 && total != ns) return fail(SC_ERR_ARG, "bodies own %d segments, %d given", total, ns);
  c->nseg = ns;
  c->nbody = nb;
  std::memset(c->seg, 0, sizeof c->seg);
  std::memset(c->pad, 0, sizeof c->pad);
  std::memset(c->body, 0, sizeof c->body);
  for (int k = 0; k < ns; ++k) c->seg[k] = Seg{segments[4 * k], segments[4 * k + 1], segments[4 * k + 2], segments[4 * k + 3]};
  for (int k = 0; k < 2 * ns; ++k) c->pad[k] = Seg{padded[4 * k], padded[4 * k + 1], padded[4 * k + 2], padded[4 * k + 3]};
  for (int b = 0; b < nb; ++b)
    c->body[b] = BodyK{bodies[b].position_x,        bodies[b].position_y, bodies[b].center_velocity_x,
                       bodies[b].center_velocity_y, bodies[b].angular_clockwise_velocity, bodies[b].n_segments, 0};
  return SC_OK;
}

int sc_set_noise_mode(sc_ctx* c, int mode, uint64_t seed) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (mode < SC_NOISE_NONE || mode > SC_NOISE_COUNTER) return fail(SC_ERR_ARG, "noise mode %d", mode);
  if (c->in_step) return fail(SC_ERR_STATE, "noise mode cannot change inside a tick");
  c->noise_mode = mode;
  c->seed = seed;
  return SC_OK;
}

// host-noise mode: every particle's offset into the tick's rand(sum C_i, 2) block (crate.py:165-170 draws in id order)
static int launch_noise_offsets(sc_ctx* c) {
  Bracket br(c, K_NOISE_OFFSETS);
  HIPCHK(hipMemsetAsync(c->cntById, 0, c->next_id * sizeof(int), c->stream));
  hipLaunchKernelGGL(k_count_by_id, dim3(grid_for(launch_bound(c))), dim3(kBlock), 0, c->stream, c->counters, c->id[1],
                     (const unsigned int*)c->rows, c->cntById);
  return launch_scan(c, c->cntById, c->offById, c->next_id, c->idBlockSums, nullptr);
}

int sc_step_begin(sc_ctx* c) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "sc_step_begin called twice");
  if (c->slab && c->noise_mode == SC_NOISE_HOST) return fail(SC_ERR_STATE, "SC_NOISE_HOST is not available in slab mode");
  HIPCHK(hipSetDevice(c->device));
  // Keep at most kMaxTicksQueued ticks of launches in flight.  The GPU publishes the number of
  // finished ticks in host-mapped memory (pass B); nothing else is needed to know how far ahead the
  // host is, and a bounded queue keeps per-tick hints (big buckets) at most that many ticks stale.
  constexpr int64_t kMaxTicksQueued = 4;
  if (!c->custom_grid) {
    int spins = 0;
    while (c->tick - (int64_t) * (volatile int*)(c->bigHintHost + 1) > kMaxTicksQueued) {
      if (++spins > 64) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;  // nothing queued: the counter is simply behind (re-upload)
        if (q != hipErrorNotReady) return fail(SC_ERR_HIP, "stream error while waiting for queued ticks: %s", hipGetErrorString(q));
        spins = 0;
      }
      sched_yield();
    }
  }
  int rc = make_world(c);
  if (rc) return rc;
  const World& w = c->w;
  int grid = grid_for(launch_bound(c));
  int cap = (int)c->cap;
  // big buckets were seen by the last scan the host knows about (an unsynchronised, possibly stale hint in host-mapped
  // memory), read ONCE per tick
  c->piles_now = c->force_rank_big || *(volatile int*)c->bigHintHost > 0;
  if (c->prebinned) {
    // the previous sc_step_finish ran K1 of this tick with the promised inputs: they must be the inputs
    const WallInputs now = wall_inputs_of(w);
    if (std::memcmp(&now, &c->promised, sizeof now) != 0)
      return fail(SC_ERR_STATE, "this tick's coefficients / segments differ from what sc_set_next_inputs promised");
    c->prebinned = false;
  } else {
    Bracket br(c, K_WALL_BIN);
    hipLaunchKernelGGL(k_wall_bin, dim3(grid), dim3(kBlock), 0, c->stream, w, c->counters, c->x[0], c->y[0], c->cellS,
                       c->wslotS, c->cellCount, c->wrec[c->tick & 1], cap);
  }
  {
    Bracket br(c, K_SCAN);
    const int64_t ncells = (int64_t)w.nrows * w.ncols;
    const int nb = (int)((ncells + 1 + kScanPerBlock - 1) / kScanPerBlock);  // covers the one-past-the-end entry
    c->scanStamp = c->scanStamp % 0x3FFFFFFFu + 1;  // 1 .. 2^30 - 1: never the cleared descriptors' 0
    hipLaunchKernelGGL(k_scan_cells, dim3(nb), dim3(kBlock), 0, c->stream, c->cellCount, c->cellStart, (int)ncells,
                       c->scanDesc, c->scanStamp, c->counters, c->sortTasks, c->scan_max_polls);
  }
  {
    Bracket br(c, K_SCATTER);
    if (piles_expected(c))
      hipLaunchKernelGGL(k_scatter<true>, dim3(grid), dim3(kBlock), 0, c->stream, c->counters, c->cellS, c->x[0],
                         c->id[0], Buckets{c->cellStart}, c->cellCount, c->keys, c->keyCell, cap, w.live_hint);
    else
      hipLaunchKernelGGL(k_scatter<false>, dim3(grid), dim3(kBlock), 0, c->stream, c->counters, c->cellS, c->x[0],
                         c->id[0], Buckets{c->cellStart}, c->cellCount, c->keys, c->keyCell, cap, w.live_hint);
  }
  const int stamp = (int)((c->tick + 1) & 0x3FFFFFFF);
  // big buckets were seen by the last scan the host knows about (an unsynchronised, possibly stale
  // hint in host-mapped memory): rank this tick's big buckets over the whole GPU first
  if (piles_expected(c)) {
    Bracket br(c, K_SCAN);
    hipLaunchKernelGGL(k_sort_big, dim3(kSortGridPerCu * c->num_cus), dim3(kSortBlock), 0, c->stream, c->counters, c->sortTasks,
                       Buckets{c->cellStart}, c->keys, c->sortedStamp, stamp);
  }
  {
    Bracket br(c, K_REORDER);
    hipLaunchKernelGGL(k_reorder, dim3((int)std::max<int64_t>(1, (launch_bound(c) + kReorderBlock - 1) / kReorderBlock)),
                       dim3(kReorderBlock), 0, c->stream, c->counters, c->keys, c->keyCell,
                       c->cellS, Buckets{c->cellStart}, c->wslotS, c->y[0], c->vx[0], c->vy[0], c->sxy, c->svv,
                       c->id[1], c->cellT, c->wslotT, c->sortedStamp, stamp, w.ncols, c->tileBounds, w.live_hint, c->bigHintDev);
  }
  // SC_NOISE_HOST (and the stand-alone search) stop after the lists: the host's rand block can only be
  // indexed once every count is known.  Otherwise the search and pass A are one launch.
  if (c->noise_mode == SC_NOISE_HOST || c->custom_grid)
    launch_pass_a<SC_NOISE_NONE, true, false>(c, K_NEIGHBORS);
  else if (c->noise_mode == SC_NOISE_COUNTER)
    launch_pass_a<SC_NOISE_COUNTER, true, true>(c, K_PASS_A);
  else
    launch_pass_a<SC_NOISE_NONE, true, true>(c, K_PASS_A);
  c->offsets_pending = false;
  if (c->noise_mode == SC_NOISE_HOST && c->next_id > 0) {
    rc = ensure_ids(c, c->next_id);
    if (rc) return rc;
    // a small world whose stream the device holds: the offsets are taken by the same launch that draws the noise
    // (sc_step_finish: k_rng_noise_small) -- unless the host brings its own block after all (sc_set_noise_host)
    if (c->rng && c->next_id <= kSmallIds)
      c->offsets_pending = true;
    else if ((rc = launch_noise_offsets(c)))
      return rc;
  }
  HIPCHK(hipGetLastError());
  c->in_step = true;
  c->etaPairs = -1;
  c->stats_live = -1;
  return SC_OK;
}

int sc_step_stats(sc_ctx* c, sc_stats* out) {
  if (!c || !out) return fail(SC_ERR_ARG, "null argument");
  if (!c->in_step) return fail(SC_ERR_STATE, "sc_step_stats needs sc_step_begin first");
  hipLaunchKernelGGL(k_count_stats, dim3(1), dim3(kBlock), 0, c->stream, c->counters, (const unsigned int*)c->rows, c->wslotT);
  int h[C_COUNT];
  int rc = read_counters(c, h);
  if (rc) return rc;
  c->stats_live = h[C_NT];
  out->particles = h[C_NT];
  out->neighbor_slots = (int64_t)(uint32_t)h[C_SUMC] + ((int64_t)h[C_SUMC_HI] << 32);
  out->max_neighbors = h[C_MAXC];
  out->wall_particles = h[C_WREC];
  out->flags = h[C_FLAGS];
  out->reserved = 0;
  return SC_OK;
}

int sc_set_noise_host(sc_ctx* c, const double* u01, int64_t n_pairs) {
  if (!c || n_pairs < 0 || (n_pairs > 0 && !u01)) return fail(SC_ERR_ARG, "bad noise array");
  if (!c->in_step) return fail(SC_ERR_STATE, "sc_set_noise_host needs sc_step_begin first");
  if (c->offsets_pending) {  // (the host draws this tick's block itself: the offsets into it are needed after all)
    c->offsets_pending = false;
    int rc = launch_noise_offsets(c);
    if (rc) return rc;
  }
  if (n_pairs > c->etaAlloc) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->eta) (void)hipFree(c->eta);
    c->eta = nullptr;
    int64_t m = n_pairs + n_pairs / 2 + 1024;
    HIPCHK(dalloc(&c->eta, 2 * m));
    c->etaAlloc = m;
  }
  if (n_pairs > 0)
    HIPCHK(hipMemcpyAsync(c->eta, u01, 2 * n_pairs * sizeof(double), hipMemcpyHostToDevice, c->stream));
  c->etaPairs = n_pairs;
  return SC_OK;
}

int sc_step_finish(sc_ctx* c) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (!c->in_step) return fail(SC_ERR_STATE, "sc_step_finish needs sc_step_begin first");
  if (c->noise_mode == SC_NOISE_HOST && c->etaPairs < 0) {
    if (!c->rng) return fail(SC_ERR_STATE, "SC_NOISE_HOST: sc_set_noise_host must be called every tick (or sc_rng_set_state once)");
    // the device holds the stream: draw the tick's rand(sum C_i, 2) there; sum C_i is the last entry of the
    // offsets sc_step_begin scanned, so the host never learns it
    const int64_t room = (int64_t)kMaxNbr * c->cap;
    if (room > c->etaAlloc) {
      HIPCHK(hipStreamSynchronize(c->stream));
      if (c->eta) (void)hipFree(c->eta);
      c->eta = nullptr;
      HIPCHK(dalloc(&c->eta, 2 * (size_t)room));
      c->etaAlloc = room;
    }
    if (c->next_id > 0) {
      Bracket br(c, K_NOISE_OFFSETS);
      if (c->offsets_pending)
        hipLaunchKernelGGL(k_rng_noise_small, dim3(1), dim3(kSmallBlock), 0, c->stream, c->rng, c->id[1], (const unsigned int*)c->rows,
                           (int)c->next_id, c->cntById, c->offById, c->eta, (long long)c->etaAlloc, c->counters);
      else
        hipLaunchKernelGGL(k_rng_noise, dim3(1), dim3(kRngBlock), 0, c->stream, c->rng, c->offById + c->next_id, c->eta,
                           (long long)c->etaAlloc, c->counters);
      c->offsets_pending = false;
    }
    c->etaPairs = 0;
  }
  // look-ahead: run K1 of the next tick in pass B's epilogue.  Slabs: pass B also packs the next halo
  // message into the buffers of the last sc_halo_pack, and sc_halo_unpack does K1 for what it appends.
  WallInputs wn;
  std::memset(&wn, 0, sizeof wn);
  const bool slab_ready = !c->slab || c->haloL || !(c->has_left || c->has_right);
  const bool fused = c->have_next && slab_ready && !c->custom_grid && !c->monitor_on;  // the monitor runs with the plain kernel
  if (fused) {
    World next;
    int rc = build_world(c, next, c->next_params, c->next_nseg, c->next_seg, nullptr, c->next_nbody, c->next_body,
                         c->tick + 1);
    if (rc) return rc;
    if (next.nrows != c->w.nrows || next.ncols != c->w.ncols) {
      rc = ensure_cells(c, (int64_t)next.nrows * next.ncols);  // the radius changed: the grid may have grown
      if (rc) return rc;
    }
    wn = wall_inputs_of(next);
  }
  c->have_next = false;
  switch (c->noise_mode) {
    case SC_NOISE_HOST:
      launch_pass_a<SC_NOISE_HOST, false, true>(c, K_DENSITY);
      launch_pass_b_any<SC_NOISE_HOST>(c, fused, wn);
      break;
    case SC_NOISE_COUNTER: launch_pass_b_any<SC_NOISE_COUNTER>(c, fused, wn); break;
    default:
      if (c->custom_grid) launch_pass_a<SC_NOISE_NONE, false, true>(c, K_DENSITY);
      launch_pass_b_any<SC_NOISE_NONE>(c, fused, wn);
      break;
  }
  if (fused) {
    c->prebinned = true;
    c->promised = wn;
  }
  HIPCHK(hipGetLastError());
  c->in_step = false;
  c->tick += 1;
  c->normals_valid = 1;
  // pass B stores exactly the live particles of this tick: a count the host has read inside the tick
  // (sc_step_stats) brings the host-side bound back down, so that a long run of emit / remove / emit
  // without downloads does not accumulate `upper` as everything ever emitted
  if (c->stats_live >= 0 && !c->slab) c->upper = c->stats_live;
  c->stats_live = -1;
  return SC_OK;
}

int sc_step(sc_ctx* c, int32_t n_ticks) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->noise_mode == SC_NOISE_HOST) return fail(SC_ERR_STATE, "sc_step is not available in SC_NOISE_HOST mode");
  if (c->slab && n_ticks > 1) return fail(SC_ERR_STATE, "slab mode: one tick per halo exchange");
  for (int t = 0; t < n_ticks; ++t) {
    int rc = sc_step_begin(c);
    if (rc) return rc;
    if (t + 1 < n_ticks) {  // the next tick of this call has the same inputs: promise them
      c->next_params = c->params;
      c->next_nseg = c->nseg;
      c->next_nbody = c->nbody;
      std::memcpy(c->next_seg, c->seg, sizeof c->next_seg);
      std::memcpy(c->next_body, c->body, sizeof c->next_body);
      c->have_next = true;
    }
    rc = sc_step_finish(c);
    if (rc) return rc;
  }
  return SC_OK;
}

int sc_tick(sc_ctx* c, const sc_tick_inputs* now, const sc_tick_inputs* next) {
  if (!c || !now) return fail(SC_ERR_ARG, "null argument");
  // (SC_NOISE_HOST needs the host's noise block between the two halves of a tick -- unless the device holds the stream)
  if (c->noise_mode == SC_NOISE_HOST && !c->rng)
    return fail(SC_ERR_STATE, "sc_tick is not available in SC_NOISE_HOST mode unless the device holds the stream (sc_rng_set_state)");
  int rc = sc_set_params(c, &now->params);
  if (rc) return rc;
  rc = sc_set_segments(c, now->segments, now->padded, now->n_segments, now->bodies, now->n_bodies);
  if (rc) return rc;
  rc = sc_step_begin(c);
  if (rc) return rc;
  if (next) {
    rc = sc_set_next_inputs(c, &next->params, next->segments, next->n_segments, next->bodies, next->n_bodies);
    if (rc) {
      c->have_next = false;
      (void)sc_step_finish(c);  // leave the context between ticks; the error of the promise is what is reported
      return rc;
    }
  }
  return sc_step_finish(c);
}

int sc_set_next_inputs(sc_ctx* c, const sc_params* p, const double* segments, int32_t ns, const sc_body* bodies,
                       int32_t nb) {
  if (!c || !p) return fail(SC_ERR_ARG, "null argument");
  if (!c->in_step) return fail(SC_ERR_STATE, "sc_set_next_inputs belongs between sc_step_begin and sc_step_finish");
  if (ns < 0 || ns > kMaxSeg) return fail(SC_ERR_CAPACITY, "%d segments, at most %d", ns, kMaxSeg);
  if (nb < 0 || nb > kMaxBody) return fail(SC_ERR_CAPACITY, "%d bodies, at most %d", nb, kMaxBody);
  if (ns > 0 && !segments) return fail(SC_ERR_ARG, "null segment array");
  int total = 0;
  for (int b = 0; b < nb; ++b) total += bodies[b].n_segments;
  if (nb > 0 && total != ns) return fail(SC_ERR_ARG, "bodies own %d segments, %d given", total, ns);
  c->next_params = *p;
  c->next_nseg = ns;
  c->next_nbody = nb;
  std::memset(c->next_seg, 0, sizeof c->next_seg);
  std::memset(c->next_body, 0, sizeof c->next_body);
  for (int k = 0; k < ns; ++k)
    c->next_seg[k] = Seg{segments[4 * k], segments[4 * k + 1], segments[4 * k + 2], segments[4 * k + 3]};
  for (int b = 0; b < nb; ++b)
    c->next_body[b] = BodyK{bodies[b].position_x,        bodies[b].position_y, bodies[b].center_velocity_x,
                            bodies[b].center_velocity_y, bodies[b].angular_clockwise_velocity, bodies[b].n_segments, 0};
  c->have_next = true;
  return SC_OK;
}

// ---- downloads -----------------------------------------------------------------------------

static int fetch(sc_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return SC_OK;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return SC_OK;
}

int sc_download_state(sc_ctx* c, double* xy, double* vxy, double* pressure, int64_t* ids, int64_t room, int64_t* n_out) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "sc_download_state inside a tick");
  int h[C_COUNT];
  int rc = read_counters(c, h);
  if (rc) return rc;
  int64_t n = h[C_NS];
  c->upper = n;
  if (n_out) *n_out = n;
  if (n > room) return fail(SC_ERR_CAPACITY, "host arrays hold %lld, %lld particles live", (long long)room, (long long)n);
  std::vector<double> hx(n), hy(n), hvx(n), hvy(n), hp(n, 0.0);
  std::vector<int> hid(n);
  size_t b = n * sizeof(double);
  if ((rc = fetch(c, hx.data(), c->x[0], b)) || (rc = fetch(c, hy.data(), c->y[0], b)) ||
      (rc = fetch(c, hvx.data(), c->vx[0], b)) || (rc = fetch(c, hvy.data(), c->vy[0], b)) ||
      (rc = fetch(c, hid.data(), c->id[0], n * sizeof(int))))
    return rc;
  // pressure is valid for the particles of the last finished tick, which are exactly the stored
  // ones unless particles were uploaded/appended since
  int64_t np = c->normals_valid ? std::min<int64_t>(n, h[C_NT]) : 0;
  if (pressure && np > 0 && (rc = fetch(c, hp.data(), c->P, np * sizeof(double)))) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<int> order;
  order.reserve(n);
  for (int64_t k = 0; k < n; ++k)
    if (std::isfinite(hx[k])) order.push_back((int)k);  // slab mode leaves dead ghost copies (x = +inf) behind
  std::sort(order.begin(), order.end(), [&](int a, int b2) { return hid[a] < hid[b2]; });
  n = (int64_t)order.size();
  if (n_out) *n_out = n;
  for (int64_t k = 0; k < n; ++k) {
    int s = order[k];
    if (xy) {
      xy[2 * k] = hx[s];
      xy[2 * k + 1] = hy[s];
    }
    if (vxy) {
      vxy[2 * k] = hvx[s];
      vxy[2 * k + 1] = hvy[s];
    }
    if (pressure) pressure[k] = s < np ? hp[s] : 0.0;
    if (ids) ids[k] = hid[s];
  }
  if (h[C_FLAGS]) {
    HIPCHK(hipMemsetAsync(c->counters + C_FLAGS, 0, sizeof(int), c->stream));
    return check_flags(h[C_FLAGS]);
  }
  return SC_OK;
}

int sc_download_sort(sc_ctx* c, int64_t* y_floored, int64_t* ids, int64_t room, int64_t* n_out) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (!c->in_step) return fail(SC_ERR_STATE, "sc_download_sort is valid between sc_step_begin and sc_step_finish");
  int h[C_COUNT];
  int rc = read_counters(c, h);
  if (rc) return rc;
  int64_t n = h[C_NT];
  if (n_out) *n_out = n;
  if (n > room) return fail(SC_ERR_CAPACITY, "host arrays too small");
  std::vector<int> cell(n), id(n);
  if ((rc = fetch(c, cell.data(), c->cellT, n * sizeof(int))) || (rc = fetch(c, id.data(), c->id[1], n * sizeof(int))))
    return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int64_t k = 0; k < n; ++k) {
    if (y_floored) y_floored[k] = (int64_t)(cell[k] / c->w.ncols) + c->w.row0;
    if (ids) ids[k] = id[k];
  }
  return SC_OK;
}

int sc_download_neighbors(sc_ctx* c, int64_t* ids, int32_t* counts, int64_t* neighbors, double* fixed_xy, int64_t room,
                          int64_t* n_out) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (!c->in_step) return fail(SC_ERR_STATE, "sc_download_neighbors is valid between sc_step_begin and sc_step_finish");
  int h[C_COUNT];
  int rc = read_counters(c, h);
  if (rc) return rc;
  int64_t n = h[C_NT];
  if (n_out) *n_out = n;
  if (n > room) return fail(SC_ERR_CAPACITY, "host arrays too small");
  std::vector<int> id(n);
  std::vector<NbrRow> rows(n);
  std::vector<double> hxy(2 * n);
  std::vector<int> slot(n);
  const int64_t nblocks = (n + kTileW - 1) / kTileW;
  std::vector<int> tb(6 * std::max<int64_t>(nblocks, 1));
  if ((rc = fetch(c, tb.data(), c->tileBoundsT, 6 * nblocks * sizeof(int)))) return rc;
  if ((rc = fetch(c, id.data(), c->id[1], n * sizeof(int))) || (rc = fetch(c, rows.data(), c->rows, n * sizeof(NbrRow))) ||
      (rc = fetch(c, hxy.data(), c->sxy, 2 * n * sizeof(double))))
    return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (neighbors)
    for (int64_t k = 0; k < n * kMaxNbr; ++k) neighbors[k] = -1;
  auto tile_of = [&](int64_t k) {  // the table holds tile slots of the particle's block
    const int* b = tb.data() + 6 * (k / kTileW);
    return Tile{b[0], b[1] - b[0], b[2], b[3] - b[2], b[4], b[5] - b[4]};
  };
  bool any_big = false;  // a block whose tile exceeds 16-bit slots: the 32-bit table holds -(index + 1)
  for (int64_t b = 0; b < nblocks; ++b) {
    const Tile tl = tile_of(b * kTileW);
    any_big |= tl.n0 + tl.n1 + tl.n2 > kRowSlotMax;
  }
  for (int s = 0; s < kMaxNbr && neighbors; ++s) {
    if (any_big) {
      if ((rc = fetch(c, slot.data(), c->nbr + (size_t)s * c->cap, n * sizeof(int)))) return rc;
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    for (int64_t k = 0; k < n; ++k) {
      if (s >= row_count(rows[k])) continue;
      const Tile tl = tile_of(k);
      const bool big = tl.n0 + tl.n1 + tl.n2 > kRowSlotMax;
      neighbors[k * kMaxNbr + s] = id[entry_index(tl, big ? slot[k] : row_entry(rows[k], s))];
    }
  }
  for (int64_t k = 0; k < n; ++k) {
    if (ids) ids[k] = id[k];
    if (counts) counts[k] = (int32_t)row_count(rows[k]);
    if (fixed_xy) {
      fixed_xy[2 * k] = hxy[2 * k];
      fixed_xy[2 * k + 1] = hxy[2 * k + 1];
    }
  }
  return SC_OK;
}

int sc_download_normals(sc_ctx* c, double* sxy, int64_t room, int64_t* n_out) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step || !c->normals_valid) return fail(SC_ERR_STATE, "sc_download_normals needs a finished tick");
  int h[C_COUNT];
  int rc = read_counters(c, h);
  if (rc) return rc;
  int64_t n = h[C_NT];
  if (n_out) *n_out = n;
  if (n > room) return fail(SC_ERR_CAPACITY, "host arrays too small");
  std::vector<double> ab(2 * n);
  std::vector<int> id(n);
  if ((rc = fetch(c, ab.data(), c->snn, 2 * n * sizeof(double))) || (rc = fetch(c, id.data(), c->id[1], n * sizeof(int))))
    return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int p, int q) { return id[p] < id[q]; });
  for (int64_t k = 0; k < n && sxy; ++k) {
    sxy[2 * k] = ab[2 * order[k]];
    sxy[2 * k + 1] = ab[2 * order[k] + 1];
  }
  return SC_OK;
}

// ---- stand-alone reference functions ----------------------------------------------------------

int sc_neighbor_search(int device, const double* xy, int64_t n, double diameter, int64_t* y_floored,
                       int64_t* sorted_indices, int32_t* counts, int64_t* table) {
  if (n < 0 || (n > 0 && !xy) || !(diameter > 0)) return fail(SC_ERR_ARG, "bad arguments");
  if (n == 0) return SC_OK;
  // the grid comes from the data's bounding box; the search itself runs on the device
  double xmin = xy[0], xmax = xy[0], ymin = xy[1], ymax = xy[1];
  for (int64_t i = 0; i < n; ++i) {
    double px = xy[2 * i], py = xy[2 * i + 1];
    if (!std::isfinite(px) || !std::isfinite(py)) return fail(SC_ERR_DOMAIN, "non-finite coordinate at %lld", (long long)i);
    xmin = std::min(xmin, px);
    xmax = std::max(xmax, px);
    ymin = std::min(ymin, py);
    ymax = std::max(ymax, py);
  }
  double c0 = std::floor(xmin / diameter), c1 = std::floor(xmax / diameter);
  double r0 = std::floor(ymin / diameter), r1 = std::floor(ymax / diameter);
  if (!(std::fabs(c0) < 4e15 && std::fabs(c1) < 4e15 && std::fabs(r0) < 4e15 && std::fabs(r1) < 4e15))
    return fail(SC_ERR_DOMAIN, "coordinates too large for the diameter");
  double ncols = c1 - c0 + 3, nrows = r1 - r0 + 3;
  if (ncols * nrows > (double)((int64_t)1 << 27))
    return fail(SC_ERR_CAPACITY, "bounding box of %.0f x %.0f cells is too sparse for a uniform grid", nrows, ncols);
  sc_ctx* c = nullptr;
  int rc = sc_create(device, n, &c);
  if (rc) return rc;
  c->custom_grid = true;
  c->force_rank_big = true;  // no previous tick to take the hint from
  c->custom_d = diameter;
  c->grid_row0 = (long long)r0 - 1;
  c->grid_col0 = (long long)c0 - 1;
  c->grid_nrows = (int)nrows;
  c->grid_ncols = (int)ncols;
  std::vector<double> zero(2 * n, 0.0);
  std::vector<int64_t> ids(n), nb((size_t)n * kMaxNbr);
  std::vector<int32_t> cn(n);
  int64_t got = 0;
  if ((rc = sc_upload_state(c, xy, zero.data(), n)) == SC_OK && (rc = sc_step_begin(c)) == SC_OK &&
      (rc = sc_download_sort(c, y_floored, sorted_indices, n, &got)) == SC_OK &&
      (rc = sc_download_neighbors(c, ids.data(), cn.data(), nb.data(), nullptr, n, &got)) == SC_OK) {
    if (got != n) {
      rc = fail(SC_ERR_DOMAIN, "%lld of %lld particles were binned", (long long)got, (long long)n);
    } else {
      for (int64_t k = 0; k < n; ++k) {
        int64_t i = ids[k];
        if (counts) counts[i] = cn[k];
        if (table) std::memcpy(table + i * kMaxNbr, nb.data() + k * kMaxNbr, kMaxNbr * sizeof(int64_t));
      }
    }
  }
  std::string keep = g_err;
  sc_destroy(c);
  g_err = keep;
  return rc;
}

// geometry_utils.py:146-172 (pad_segments) on the host, operation for operation: o = cw90(b - a) * pad / |b - a| with the
// norm as np.linalg.norm takes it for two components (sqrt of the sum of the squares, separately rounded -- this file is
// compiled with -ffp-contract=off); first every (a + o, b + o), then every (b - o, a - o).  In the library because the
// padded twins of a moving wall are needed every tick and the NumPy form of these thirty operations costs the host 15-25 us.
int sc_pad_segments(const double* segments, int32_t ns, double pad_distance, double* padded) {
  if (ns < 0 || (ns > 0 && (!segments || !padded))) return fail(SC_ERR_ARG, "bad arguments");
  for (int k = 0; k < ns; ++k) {
    const double ax = segments[4 * k], ay = segments[4 * k + 1], bx = segments[4 * k + 2], by = segments[4 * k + 3];
    const double alx = bx - ax, aly = by - ay;
    const double nx = aly, ny = -alx;  // clockwise quarter turn of (end - start)
    const double norm = std::sqrt(nx * nx + ny * ny);
    const double ox = nx * pad_distance / norm, oy = ny * pad_distance / norm;
    double* plus = padded + 4 * k;
    double* minus = padded + 4 * (ns + k);
    plus[0] = ax + ox; plus[1] = ay + oy; plus[2] = bx + ox; plus[3] = by + oy;
    minus[0] = bx - ox; minus[1] = by - oy; minus[2] = ax - ox; minus[3] = ay - oy;
  }
  return SC_OK;
}

int sc_points_to_segments(int device, const double* xy, int64_t n, const double* segments, int32_t ns, double* nearest,
                          double* distances) {
  if (n < 0 || ns < 0 || (n > 0 && !xy) || (ns > 0 && !segments)) return fail(SC_ERR_ARG, "bad arguments");
  if (n == 0 || ns == 0) return SC_OK;
  HIPCHK(hipSetDevice(device));
  double *dxy = nullptr, *dseg = nullptr, *dnear = nullptr, *ddist = nullptr;
  size_t t = (size_t)n * ns;
  hipError_t e = dalloc(&dxy, 2 * (size_t)n);
  if (e == hipSuccess) e = dalloc(&dseg, 4 * (size_t)ns);
  if (e == hipSuccess) e = dalloc(&dnear, 2 * t);
  if (e == hipSuccess) e = dalloc(&ddist, t);
  if (e == hipSuccess) e = hipMemcpy(dxy, xy, 2 * n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dseg, segments, 4 * ns * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_points_to_segments, dim3((unsigned)((t + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, dxy, (int)n,
                       dseg, (int)ns, dnear, ddist);
    e = hipGetLastError();
  }
  if (e == hipSuccess && nearest) e = hipMemcpy(nearest, dnear, 2 * t * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && distances) e = hipMemcpy(distances, ddist, t * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dxy);
  (void)hipFree(dseg);
  (void)hipFree(dnear);
  (void)hipFree(ddist);
  if (e != hipSuccess) return fail(SC_ERR_HIP, "sc_points_to_segments: %s", hipGetErrorString(e));
  return SC_OK;
}


// ---- multi-GPU slabs ---------------------------------------------------------------------------

int sc_set_slab_axis(sc_ctx* c, int32_t axis) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "slab cannot change inside a tick");
  if (axis != 0 && axis != 1) return fail(SC_ERR_ARG, "slab axis: 0 (columns of x) or 1 (rows of y)");
  c->slab_axis = axis;
  c->halo_ring_from = c->tick;
  return SC_OK;
}

int sc_set_slab(sc_ctx* c, int64_t col_lo, int64_t col_hi, int32_t halo, int32_t has_left, int32_t has_right) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "slab cannot change inside a tick");
  if (col_hi <= col_lo || halo < 3) return fail(SC_ERR_ARG, "slab needs col_lo < col_hi and a halo of at least 3 columns");
  c->slab = true;
  c->own_lo = col_lo;
  c->own_hi = col_hi;
  c->halo = halo;
  c->has_left = has_left ? 1 : 0;
  c->has_right = has_right ? 1 : 0;
  c->halo_ring_from = c->tick;  // new cuts: the halo counts of earlier ticks say nothing about the coming ones
  if (!c->owned_out) HIPCHK(dalloc(&c->owned_out, 1));
  return SC_OK;
}

// Records a halo message of tick `tick` carries, from the count the same direction had `kHaloLag` ticks earlier
// (+50 % and 1024 records of headroom, in steps of 256).  Sender and receiver evaluate this on the same number:
// the sender published what it packed, the receiver what the header it received said.
static int64_t halo_message_records(int64_t count, int64_t cap) {
  const int64_t want = count + count / 2 + 1024;
  return std::min<int64_t>(cap, (want + 255) / 256 * 256);
}

int sc_halo_sizes(sc_ctx* c, int64_t cap_records, int64_t* send_left, int64_t* recv_left, int64_t* send_right,
                  int64_t* recv_right) {
  if (!c || !send_left || !recv_left || !send_right || !recv_right || cap_records < 1) return fail(SC_ERR_ARG, "bad arguments");
  if (!c->slab) return fail(SC_ERR_STATE, "sc_set_slab first");
  constexpr int64_t kHaloLag = 6;  // more than the ticks the host may run ahead of the device (sc_step_begin)
  static_assert(kHaloLag < kHaloRing, "the ring must still hold the tick the sizes come from");
  *send_left = *recv_left = *send_right = *recv_right = cap_records;
  const int64_t src = c->tick - kHaloLag;
  if (src < c->halo_ring_from) return SC_OK;  // no history yet: whole buffers
  // tick `src` has finished on the device (at most a few ticks are ever queued), so its counts are published
  int spins = 0;
  while ((int64_t) * (volatile int*)(c->bigHintHost + 1) <= src) {
    if (++spins > 64) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) return fail(SC_ERR_HIP, "stream error while waiting for halo counts: %s", hipGetErrorString(q));
      spins = 0;
    }
    sched_yield();
  }
  if ((int64_t) * (volatile int*)(c->bigHintHost + 1) <= src) return SC_OK;  // counter behind (fresh upload): whole buffers
  const volatile int* ring = c->bigHintHost + 4 + 4 * (src % kHaloRing);
  *send_left = halo_message_records(ring[0], cap_records);
  *send_right = halo_message_records(ring[1], cap_records);
  *recv_left = halo_message_records(ring[2], cap_records);
  *recv_right = halo_message_records(ring[3], cap_records);
  return SC_OK;
}

int sc_column_histogram(sc_ctx* c, int64_t col0, int32_t ncols, int64_t* hist) {
  if (!c || !hist || ncols < 1) return fail(SC_ERR_ARG, "bad arguments");
  if (c->in_step) return fail(SC_ERR_STATE, "sc_column_histogram inside a tick");
  if (!c->have_params && !c->custom_grid) return fail(SC_ERR_STATE, "sc_set_params has not been called");
  HIPCHK(hipSetDevice(c->device));
  if (ncols > c->colHistAlloc) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->colHist) (void)hipFree(c->colHist);
    c->colHist = nullptr;
    HIPCHK(dalloc(&c->colHist, (size_t)ncols + 256));
    c->colHistAlloc = ncols + 256;
  }
  HIPCHK(hipMemsetAsync(c->colHist, 0, ncols * sizeof(int), c->stream));
  const double d = c->custom_grid ? c->custom_d : c->params.particle_radius * 2;
  hipLaunchKernelGGL(k_column_histogram, dim3(grid_for(launch_bound(c))), dim3(kBlock), 0, c->stream, c->counters, c->x[0],
                     c->slab_axis ? c->y[0] : c->x[0], d, (long long)col0, (int)ncols, c->colHist);
  std::vector<int> h(ncols);
  HIPCHK(hipMemcpyAsync(h.data(), c->colHist, ncols * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < ncols; ++k) hist[k] = h[k];
  return SC_OK;
}

int sc_upload_state_ids(sc_ctx* c, const double* xy, const double* vxy, const int64_t* ids, int64_t n) {
  if (!c || (n > 0 && !ids)) return fail(SC_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  return put_particles(c, xy, vxy, n, true, ids);
}

int sc_append_particles_ids(sc_ctx* c, const double* xy, const double* vxy, const int64_t* ids, int64_t n) {
  if (!c || (n > 0 && !ids)) return fail(SC_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  return put_particles(c, xy, vxy, n, false, ids);
}

int sc_halo_pack(sc_ctx* c, double* dev_left, double* dev_right, int64_t cap_records) {
  if (!c || !dev_left || !dev_right || cap_records < 1) return fail(SC_ERR_ARG, "bad halo buffers");
  if (!c->slab) return fail(SC_ERR_STATE, "sc_set_slab first");
  if (c->in_step) return fail(SC_ERR_STATE, "halo exchange happens between ticks");
  int rc = make_world(c);
  if (rc) return rc;
  if (c->prebinned) return fail(SC_ERR_STATE, "the halo message of the promised tick was packed by sc_step_finish");
  c->band_pending = false;  // this message depends on the kernel below, not on a split force kernel
  c->haloL = dev_left;  // stay bound: with sc_set_next_inputs, sc_step_finish packs the next message itself
  c->haloR = dev_right;
  c->haloCap = (int)cap_records;
  Bracket br(c, K_HALO_PACK);
  hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(launch_bound(c))), dim3(kBlock), 0, c->stream, c->w, c->counters, c->x[0],
                     c->y[0], c->vx[0], c->vy[0], c->id[0], dev_left, dev_right, (int)cap_records, (int)c->cap);
  HIPCHK(hipGetLastError());
  return SC_OK;
}

int sc_halo_unpack(sc_ctx* c, const double* from_left, int64_t left_records, const double* from_right,
                   int64_t right_records) {
  if (!c || (!from_left && !from_right) || (from_left && left_records < 1) || (from_right && right_records < 1))
    return fail(SC_ERR_ARG, "bad halo buffers");
  if (!c->slab) return fail(SC_ERR_STATE, "sc_set_slab first");
  if (c->in_step) return fail(SC_ERR_STATE, "halo exchange happens between ticks");
  Bracket br(c, K_HALO_UNPACK);
  const int capL = from_left ? (int)left_records : 0, capR = from_right ? (int)right_records : 0;
  const dim3 grid(grid_for(capL + capR)), block(kBlock);
  int* ring = c->bigHintDev + 4 + 4 * (int)(c->tick % kHaloRing);
  if (c->prebinned) {  // the stored particles went through K1 of the coming tick in pass B: same for the arrivals
    hipLaunchKernelGGL(k_halo_unpack<true>, grid, block, 0, c->stream, from_left, from_right, capL, capR,
                       c->counters, c->x[0], c->y[0], c->vx[0], c->vy[0], c->id[0], (int)c->cap, c->haloL, c->haloR,
                       c->promised, c->cellS, c->wslotS, c->cellCount, c->wrec[c->tick & 1], ring);
  } else {
    WallInputs none;
    std::memset(&none, 0, sizeof none);
    hipLaunchKernelGGL(k_halo_unpack<false>, grid, block, 0, c->stream, from_left, from_right, capL, capR,
                       c->counters, c->x[0], c->y[0], c->vx[0], c->vy[0], c->id[0], (int)c->cap, c->haloL, c->haloR, none,
                       c->cellS, c->wslotS, c->cellCount, c->wrec[c->tick & 1], ring);
  }
  HIPCHK(hipGetLastError());
  return SC_OK;
}

#define RCCLCHK(expr)                                                                       \
  do {                                                                                      \
    int rc_ = (expr);                                                                       \
    if (rc_ != 0) return fail(SC_ERR_HIP, "RCCL: %s failed: %s", #expr, rccl_error(rc_)); \
  } while (0)

static int ensure_side_stream(sc_ctx* c) {
  if (!c->side_stream) HIPCHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
  // (device-side ordering only: without the system-scope fence an event between two kernels costs ~1 us instead of ~10)
  if (!c->ev_band) HIPCHK(hipEventCreateWithFlags(&c->ev_band, hipEventDisableTiming | hipEventDisableSystemFence));
  // (ev_xchg orders halo buffers that a peer GPU wrote: it keeps the system-scope fence)
  if (!c->ev_xchg) HIPCHK(hipEventCreateWithFlags(&c->ev_xchg, hipEventDisableTiming));
  return SC_OK;
}

int sc_set_halo_overlap(sc_ctx* c, int on) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "halo overlap cannot change inside a tick");
  if (on && !c->slab) return fail(SC_ERR_STATE, "sc_set_slab first");
  HIPCHK(hipSetDevice(c->device));
  if (on) {
    int rc = ensure_side_stream(c);
    if (rc) return rc;
  }
  c->overlap = on != 0;
  return SC_OK;
}

int sc_set_band_flag(sc_ctx* c, int on) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "the band mode cannot change inside a tick");
  c->band_by_flag = on != 0;  // (a band that is pending keeps the announcement it was launched with: band_flagged)
  return SC_OK;
}

int sc_side_stream(sc_ctx* c, void** stream) {
  if (!c || !stream) return fail(SC_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  int rc = ensure_side_stream(c);
  if (rc) return rc;
  *stream = (void*)c->side_stream;
  return SC_OK;
}

// side stream <- everything the halo message of the coming tick depends on (the band blocks of pass B when the last
// tick packed it, else all work queued so far); `peer`: also what that context's message depends on
int sc_halo_overlap_begin(sc_ctx* c, sc_ctx* peer) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  HIPCHK(hipSetDevice(c->device));
  int rc = ensure_side_stream(c);
  if (rc) return rc;
  for (sc_ctx* q : {c, peer}) {
    if (!q) continue;
    if (q != c && (rc = ensure_side_stream(q))) return rc;
    if (q->band_pending && q->band_flagged) {  // the window blocks of q's one-launch force kernel
      hipLaunchKernelGGL(k_wait_band, dim3(1), dim3(1), 0, c->side_stream, q->counters, q->band_epoch);
      continue;
    }
    if (!q->band_pending) HIPCHK(hipEventRecord(q->ev_band, q->stream));  // no split pass B before: wait for all of it
    HIPCHK(hipStreamWaitEvent(c->side_stream, q->ev_band, 0));
  }
  return SC_OK;
}

// context's stream <- what was enqueued on the side stream since sc_halo_overlap_begin (the received buffers)
int sc_halo_overlap_end(sc_ctx* c) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (!c->side_stream || !c->ev_xchg) return fail(SC_ERR_STATE, "sc_halo_overlap_begin first");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventRecord(c->ev_xchg, c->side_stream));
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_xchg, 0));
  c->band_pending = false;
  return SC_OK;
}

int sc_comm_available(const char* rccl_path) {
  if (rccl_load(rccl_path)) return fail(SC_ERR_HIP, "%s", rccl_api().error.c_str());
  return SC_OK;
}

int sc_comm_unique_id(const char* rccl_path, void* id) {
  if (!id) return fail(SC_ERR_ARG, "null argument");
  if (rccl_load(rccl_path)) return fail(SC_ERR_HIP, "%s", rccl_api().error.c_str());
  RcclUniqueId u;
  RCCLCHK(rccl_api().GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return SC_OK;
}

int sc_comm_init(sc_ctx* c, const char* rccl_path, const void* id, int32_t rank, int32_t world) {
  if (!c || !id) return fail(SC_ERR_ARG, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(SC_ERR_ARG, "rank %d of %d", rank, world);
  if (c->comm) return fail(SC_ERR_STATE, "sc_comm_init called twice");
  if (rccl_load(rccl_path)) return fail(SC_ERR_HIP, "%s", rccl_api().error.c_str());
  HIPCHK(hipSetDevice(c->device));
  RcclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  RCCLCHK(rccl_api().CommInitRank(&c->comm, world, u, rank));
  c->comm_rank = rank;
  c->comm_world = world;
  return SC_OK;
}

int sc_comm_destroy(sc_ctx* c) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (!c->comm) return SC_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  RcclComm comm = c->comm;
  c->comm = nullptr;
  RCCLCHK(rccl_api().CommDestroy(comm));
  return SC_OK;
}

int sc_halo_exchange(sc_ctx* c, const double* send_left, int64_t send_left_records, double* recv_left,
                     int64_t recv_left_records, int32_t left_rank, const double* send_right, int64_t send_right_records,
                     double* recv_right, int64_t recv_right_records, int32_t right_rank) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (!c->comm) return fail(SC_ERR_STATE, "sc_comm_init first");
  if (c->in_step) return fail(SC_ERR_STATE, "halo exchange happens between ticks");
  if ((left_rank >= 0 && (!send_left || !recv_left || left_rank >= c->comm_world || send_left_records < 1 || recv_left_records < 1)) ||
      (right_rank >= 0 && (!send_right || !recv_right || right_rank >= c->comm_world || send_right_records < 1 || recv_right_records < 1)))
    return fail(SC_ERR_ARG, "neighbor ranks %d / %d need their buffers and record counts and must be below %d", left_rank,
                right_rank, c->comm_world);
  auto doubles = [](int64_t records) { return (size_t)(records + 1) * kHaloFields; };  // + the header record
  const RcclApi& r = rccl_api();
  HIPCHK(hipSetDevice(c->device));
  hipStream_t xs = c->stream;
  if (c->overlap) {  // on the side stream, next to the interior blocks of the last pass B
    int rc0 = sc_halo_overlap_begin(c, nullptr);
    if (rc0) return rc0;
    xs = c->side_stream;
  }
  RCCLCHK(r.GroupStart());
  int rc = 0;
  // posting order is the same on every rank (left pair, then right pair): rank k's right pair meets rank k+1's left pair
  if (left_rank >= 0) {
    if (!rc) rc = r.Send(send_left, doubles(send_left_records), kRcclDouble, left_rank, c->comm, xs);
    if (!rc) rc = r.Recv(recv_left, doubles(recv_left_records), kRcclDouble, left_rank, c->comm, xs);
  }
  if (right_rank >= 0) {
    if (!rc) rc = r.Send(send_right, doubles(send_right_records), kRcclDouble, right_rank, c->comm, xs);
    if (!rc) rc = r.Recv(recv_right, doubles(recv_right_records), kRcclDouble, right_rank, c->comm, xs);
  }
  const int rc_end = r.GroupEnd();
  if (rc) return fail(SC_ERR_HIP, "RCCL: send/recv failed: %s", rccl_error(rc));
  if (rc_end) return fail(SC_ERR_HIP, "RCCL: ncclGroupEnd failed: %s", rccl_error(rc_end));
  if (c->overlap) return sc_halo_overlap_end(c);
  return SC_OK;
}

int sc_owned_count(sc_ctx* c, int64_t* n) {
  if (!c || !n) return fail(SC_ERR_ARG, "null argument");
  if (c->in_step) return fail(SC_ERR_STATE, "sc_owned_count inside a tick");
  int rc = c->slab ? make_world(c) : SC_OK;
  if (rc) return rc;
  if (!c->owned_out) HIPCHK(dalloc(&c->owned_out, 1));
  HIPCHK(hipMemsetAsync(c->owned_out, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_owned_count, dim3(grid_for(launch_bound(c))), dim3(kBlock), 0, c->stream, c->counters, c->x[0],
                     c->owned_out);
  int h = 0;
  HIPCHK(hipMemcpyAsync(&h, c->owned_out, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *n = h;
  return SC_OK;
}

// ---- force monitor ------------------------------------------------------------------------------

int sc_enable_force_monitor(sc_ctx* c, int on) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "the force monitor cannot change inside a tick");
  if (c->prebinned) return fail(SC_ERR_STATE, "the force monitor cannot change after sc_set_next_inputs promised the next tick");
  HIPCHK(hipSetDevice(c->device));
  if (on && !c->monitor) HIPCHK(dalloc(&c->monitor, (size_t)kMonPhases + 1));
  if (on) HIPCHK(hipMemsetAsync(c->monitor, 0, (kMonPhases + 1) * sizeof(double), c->stream));
  c->monitor_on = on != 0;
  return SC_OK;
}

int sc_get_force_monitor(sc_ctx* c, double* sums, int64_t* particles) {
  if (!c || !sums || !particles) return fail(SC_ERR_ARG, "null argument");
  if (!c->monitor_on) return fail(SC_ERR_STATE, "sc_enable_force_monitor first");
  double h[kMonPhases + 1];
  HIPCHK(hipMemcpyAsync(h, c->monitor, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemsetAsync(c->monitor, 0, sizeof h, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < kMonPhases; ++k) sums[k] = h[k];
  *particles = (int64_t)h[kMonPhases];
  return SC_OK;
}

// ---- checkpoint ---------------------------------------------------------------------------------
// The stored state is copied device-to-device on the context's stream (a few microseconds), the copy travels to
// pinned host memory on a side stream, and the ticks that follow run meanwhile; sc_checkpoint_finish waits for the
// side stream only.

int sc_checkpoint_begin(sc_ctx* c) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (c->in_step) return fail(SC_ERR_STATE, "sc_checkpoint_begin inside a tick");
  if (c->snap_pending) return fail(SC_ERR_STATE, "a checkpoint is already under way: sc_checkpoint_finish first");
  HIPCHK(hipSetDevice(c->device));
  // After a promised tick the storage arrays already hold the coming tick's removal and wall fix while its cell
  // indices and bucket counts live in buffers a snapshot does not take: a restore would run that wall pass a second
  // time, on fixed positions.  (Crate.run / physics_tick never leave a promise pending between calls.)
  if (c->prebinned)
    return fail(SC_ERR_STATE, "sc_checkpoint_begin after sc_set_next_inputs promised the next tick: run that tick first");
  // the side stream may exist already (halo overlap creates it): every snapshot resource is created on its own
  {
    const int rc = ensure_side_stream(c);
    if (rc) return rc;
  }
  if (!c->snap_ready) HIPCHK(hipEventCreateWithFlags(&c->snap_ready, hipEventDisableTiming));
  if (!c->snap_done) HIPCHK(hipEventCreateWithFlags(&c->snap_done, hipEventDisableTiming));
  if (!c->snap_counters_h) HIPCHK(hipHostMalloc((void**)&c->snap_counters_h, C_COUNT * sizeof(int), hipHostMallocDefault));
  if (!c->snap_rng_h) HIPCHK(hipHostMalloc((void**)&c->snap_rng_h, sizeof(RngState), hipHostMallocDefault));
  if (!c->snap_rng_d) HIPCHK(dalloc(&c->snap_rng_d, 1));
  const int64_t n = launch_bound(c);  // a host-side bound of the stored count; the exact count travels with the copy
  if (n > c->snapAlloc) {
    HIPCHK(hipStreamSynchronize(c->side_stream));
    for (int k = 0; k < 4; ++k) {
      if (c->snap_d[k]) (void)hipFree(c->snap_d[k]);
      if (c->snap_h[k]) (void)hipHostFree(c->snap_h[k]);
      c->snap_d[k] = nullptr;
      c->snap_h[k] = nullptr;
    }
    if (c->snap_id_d) (void)hipFree(c->snap_id_d);
    if (c->snap_id_h) (void)hipHostFree(c->snap_id_h);
    c->snap_id_d = nullptr;
    c->snap_id_h = nullptr;
    const int64_t m = std::min<int64_t>(c->cap, n + n / 2 + 1024);
    for (int k = 0; k < 4; ++k) {
      HIPCHK(dalloc(&c->snap_d[k], (size_t)m));
      HIPCHK(hipHostMalloc((void**)&c->snap_h[k], std::max<size_t>(m, 1) * sizeof(double), hipHostMallocDefault));
    }
    HIPCHK(dalloc(&c->snap_id_d, (size_t)m));
    HIPCHK(hipHostMalloc((void**)&c->snap_id_h, std::max<size_t>(m, 1) * sizeof(int), hipHostMallocDefault));
    c->snapAlloc = m;
  }
  const double* src[4] = {c->x[0], c->y[0], c->vx[0], c->vy[0]};
  // on the context's stream: after the last tick, before the next one changes the storage arrays
  for (int k = 0; k < 4 && n > 0; ++k)
    HIPCHK(hipMemcpyAsync(c->snap_d[k], src[k], n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  if (n > 0) HIPCHK(hipMemcpyAsync(c->snap_id_d, c->id[0], n * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
  c->snap_has_rng = c->rng != nullptr;
  if (c->rng) HIPCHK(hipMemcpyAsync(c->snap_rng_d, c->rng, sizeof(RngState), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->snap_counters_h, c->counters, C_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipEventRecord(c->snap_ready, c->stream));
  // on the side stream: the snapshot goes to pinned host memory while the context's stream runs on
  HIPCHK(hipStreamWaitEvent(c->side_stream, c->snap_ready, 0));
  for (int k = 0; k < 4 && n > 0; ++k)
    HIPCHK(hipMemcpyAsync(c->snap_h[k], c->snap_d[k], n * sizeof(double), hipMemcpyDeviceToHost, c->side_stream));
  if (n > 0) HIPCHK(hipMemcpyAsync(c->snap_id_h, c->snap_id_d, n * sizeof(int), hipMemcpyDeviceToHost, c->side_stream));
  if (c->rng) HIPCHK(hipMemcpyAsync(c->snap_rng_h, c->snap_rng_d, sizeof(RngState), hipMemcpyDeviceToHost, c->side_stream));
  HIPCHK(hipEventRecord(c->snap_done, c->side_stream));
  c->snap_n_bound = n;
  c->snap_tick = c->tick;
  c->snap_pending = true;
  return SC_OK;
}

int sc_checkpoint_finish(sc_ctx* c, double* xy, double* vxy, int64_t* ids, int64_t room, int64_t* n_out, int64_t* tick,
                         int64_t* next_id, uint32_t* rng_key, int32_t* rng_pos) {
  if (!c || !n_out) return fail(SC_ERR_ARG, "null argument");
  if (!c->snap_pending) return fail(SC_ERR_STATE, "sc_checkpoint_begin first");
  HIPCHK(hipEventSynchronize(c->snap_ready));  // the counters' copy rode on the context's stream up to here
  HIPCHK(hipEventSynchronize(c->snap_done));
  c->snap_pending = false;
  const int64_t stored = std::min<int64_t>(c->snap_counters_h[C_NS], c->snap_n_bound);
  std::vector<int> order;
  order.reserve(stored);
  for (int64_t k = 0; k < stored; ++k)
    if (std::isfinite(c->snap_h[0][k])) order.push_back((int)k);  // slab mode leaves dead ghost copies (x = +inf) behind
  std::sort(order.begin(), order.end(), [&](int a, int b) { return c->snap_id_h[a] < c->snap_id_h[b]; });
  const int64_t n = (int64_t)order.size();
  *n_out = n;
  if (tick) *tick = c->snap_tick;
  if (next_id) *next_id = c->snap_counters_h[C_NEXT_ID];
  if (rng_pos) *rng_pos = c->snap_has_rng ? c->snap_rng_h->pos : -1;
  if (rng_key && c->snap_has_rng) std::memcpy(rng_key, c->snap_rng_h->mt, sizeof c->snap_rng_h->mt);
  if (n > room) return fail(SC_ERR_CAPACITY, "host arrays hold %lld, the checkpoint has %lld particles", (long long)room, (long long)n);
  for (int64_t k = 0; k < n; ++k) {
    const int s = order[k];
    if (xy) {
      xy[2 * k] = c->snap_h[0][s];
      xy[2 * k + 1] = c->snap_h[1][s];
    }
    if (vxy) {
      vxy[2 * k] = c->snap_h[2][s];
      vxy[2 * k + 1] = c->snap_h[3][s];
    }
    if (ids) ids[k] = c->snap_id_h[s];
  }
  return SC_OK;
}

int sc_restore_counters(sc_ctx* c, int64_t tick, int64_t next_id) {
  if (!c || tick < 0 || next_id < 0 || next_id > std::numeric_limits<int>::max()) return fail(SC_ERR_ARG, "bad tick / next id");
  if (c->in_step) return fail(SC_ERR_STATE, "sc_restore_counters inside a tick");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->prebinned) {  // a promised tick is abandoned: forget its bucket counts
    HIPCHK(hipMemsetAsync(c->cellCount, 0, c->cellAlloc * sizeof(int), c->stream));
    c->prebinned = false;
  }
  c->tick = tick;
  c->halo_ring_from = tick;
  c->live_hint_from = tick;
  c->bigHintHost[1] = (int)tick;  // "ticks finished": nothing of the new numbering is queued
  c->next_id = std::max<int64_t>(c->next_id, next_id);
  const int nid = (int)c->next_id;
  HIPCHK(hipMemcpyAsync(c->counters + C_NEXT_ID, &nid, sizeof nid, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return SC_OK;
}

// ---- NumPy's global MT19937 stream on the device (sc_rng.h) -------------------------------------

int sc_rng_set_state(sc_ctx* c, const uint32_t* key, int32_t pos) {
  if (!c || !key || pos < 0 || pos > kMtN) return fail(SC_ERR_ARG, "an MT19937 state is 624 words and a position in [0, 624]");
  if (c->in_step) return fail(SC_ERR_STATE, "the generator cannot change inside a tick");
  HIPCHK(hipSetDevice(c->device));
  if (!c->rng) HIPCHK(dalloc(&c->rng, 1));
  RngState h;
  std::memcpy(h.mt, key, sizeof h.mt);
  h.pos = pos;
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(c->rng, &h, sizeof h, hipMemcpyHostToDevice));
  return SC_OK;
}

int sc_rng_get_state(sc_ctx* c, uint32_t* key, int32_t* pos) {
  if (!c || !key || !pos) return fail(SC_ERR_ARG, "null argument");
  if (!c->rng) return fail(SC_ERR_STATE, "sc_rng_set_state has not been called");
  if (c->in_step) return fail(SC_ERR_STATE, "sc_rng_get_state inside a tick");
  RngState h;
  HIPCHK(hipMemcpyAsync(&h, c->rng, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::memcpy(key, h.mt, sizeof h.mt);
  *pos = h.pos;
  return SC_OK;
}

int sc_emit_particles(sc_ctx* c, const sc_source* sources, int32_t n_sources, double dt, int64_t max_particles) {
  if (!c || n_sources < 0 || (n_sources > 0 && !sources)) return fail(SC_ERR_ARG, "bad sources");
  if (!c->rng) return fail(SC_ERR_STATE, "sc_rng_set_state has not been called");
  if (c->in_step) return fail(SC_ERR_STATE, "particles cannot change between sc_step_begin and sc_step_finish");
  if (c->prebinned) return fail(SC_ERR_STATE, "particles cannot be emitted after sc_set_next_inputs promised the next tick");
  if (n_sources > kMaxSources) return fail(SC_ERR_CAPACITY, "%d particle sources, at most %d", n_sources, kMaxSources);
  if (n_sources == 0) return SC_OK;
  SourcesK k;
  std::memset(&k, 0, sizeof k);
  k.n = n_sources;
  int64_t most = 0;
  for (int i = 0; i < n_sources; ++i) {
    const sc_source& s = sources[i];
    const double p = dt;
    // the legacy binomial for p <= 0.5: inversion up to n p = 30 (both YAML scenes: n p = 4 and 14), BTPE beyond;
    // p > 0.5 (a time step above one half) is not on the device
    if (!(p > 0.0 && p <= 0.5) || s.flow < 1)
      return fail(SC_ERR_DOMAIN, "binomial(%lld, %g): the device draws NumPy's legacy binomial for 0 < p <= 0.5 only",
                  (long long)s.flow, p);
    SourceK& d = k.src[i];
    d.radius = s.radius; d.px = s.position_x; d.py = s.position_y; d.vx = s.velocity_x; d.vy = s.velocity_y;
    d.noise = s.noise; d.flow = s.flow; d.p = p;
    d.q = 1.0 - p;
    d.qn = std::exp((double)s.flow * std::log(d.q));
    const double np_ = (double)s.flow * p;
    d.bound = (long long)std::min((double)s.flow, np_ + 10.0 * std::sqrt(np_ * d.q + 1));
    d.btpe = np_ > 30.0 ? 1 : 0;
    if (d.btpe) {  // randomkit's rk_binomial_btpe set-up, in its operation order (r = p, q = 1 - p here)
      const double n = (double)s.flow, r = p, q = d.q, fm = n * r + r;
      d.m = (long long)std::floor(fm);
      d.p1 = std::floor(2.195 * std::sqrt(n * r * q) - 4.6 * q) + 0.5;
      d.xm = (double)d.m + 0.5;
      d.xl = d.xm - d.p1;
      d.xr = d.xm + d.p1;
      d.c = 0.134 + 20.5 / (15.3 + (double)d.m);
      double a = (fm - d.xl) / (fm - d.xl * r);
      d.laml = a * (1.0 + a / 2.0);
      a = (d.xr - fm) / (d.xr * q);
      d.lamr = a * (1.0 + a / 2.0);
      d.p2 = d.p1 * (1.0 + 2.0 * d.c);
      d.p3 = d.p2 + d.c / d.laml;
      d.p4 = d.p3 + d.c / d.lamr;
      d.nrq = n * r * q;
    }
    most += d.bound;
  }
  // host-side bounds of the stored count and of the ids: at most `bound` particles per source; the live count a
  // recent tick published (progress block) keeps the bound from drifting away without any synchronisation
  // (the device writes the tick number last: the three words belong together when it reads the same before and after)
  int64_t done = *(volatile int*)(c->bigHintHost + 1);
  const int64_t live = *(volatile int*)(c->bigHintHost + 2), published_ids = *(volatile int*)(c->bigHintHost + 3);
  std::atomic_thread_fence(std::memory_order_acquire);
  if (*(volatile int*)(c->bigHintHost + 1) != done) done = -1;  // a tick finished in between: no hint this time
  int64_t upper = c->upper + most;
  if (done > c->live_hint_from && c->tick >= done && c->tick - done <= 8)
    upper = std::min(upper, live + (c->tick - done + 1) * most);
  upper = std::min<int64_t>(upper, std::max<int64_t>(max_particles, c->upper));
  if (upper > c->cap) {
    int h[C_COUNT];
    int rc = read_counters(c, h);  // rare: the bound reached the capacity, look at the real count
    if (rc) return rc;
    upper = std::min<int64_t>(h[C_NS] + most, std::max<int64_t>(max_particles, h[C_NS]));
    if (upper > c->cap) return fail(SC_ERR_CAPACITY, "%lld particles may exceed the context capacity %lld", (long long)upper, (long long)c->cap);
  }
  // The host's id counter is a bound too (the device hands out the real ids): every call adds the binomial's restart
  // bound, several times the particles actually emitted, and the id tables of SC_NOISE_HOST are sized and scanned by
  // it every tick.  The count the device published with a recent tick pulls it back, like `upper` above.
  c->emit_most = std::max(c->emit_most, most);
  if (done > c->live_hint_from && c->tick >= done && c->tick - done <= 8) {
    if (published_ids > 0) c->next_id = std::min(c->next_id, published_ids + (c->tick - done + 1) * c->emit_most);
  }
  if (c->next_id + most > std::numeric_limits<int>::max()) return fail(SC_ERR_CAPACITY, "particle ids exhausted");
  HIPCHK(hipSetDevice(c->device));
  {
    Bracket br(c, K_APPEND);
    hipLaunchKernelGGL(k_rng_emit, dim3(1), dim3(64), 0, c->stream, k, (long long)max_particles, c->rng, c->counters, c->x[0],
                       c->y[0], c->vx[0], c->vy[0], c->id[0], (int)c->cap);
  }
  HIPCHK(hipGetLastError());
  c->upper = upper;
  c->next_id += most;  // an upper bound from here on: the device counts the ids it hands out (C_NEXT_ID)
  return SC_OK;
}

#ifdef SC_STAMPS
// diagnostic build: copies the stamp buffer (kStampKernels x 65536 waves x kStampSlots slots, int64) to the host
int sc_debug_stamps(sc_ctx* c, long long* out) {
  if (!c || !out) return fail(SC_ERR_ARG, "null argument");
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(sc::g_stamps), sizeof(long long) * sc::kStampKernels * sc::kStampWaves * sc::kStampSlots));
  return SC_OK;
}
#endif

#ifdef SC_TIMELINE
// diagnostic build: [kTlKernels][65536][4] = (start, end) on the 100 MHz clock, HW_ID, XCC_ID of every wave of the last pass A / pass B
int sc_debug_timeline(sc_ctx* c, long long* out) {
  if (!c || !out) return fail(SC_ERR_ARG, "null argument");
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(sc::g_timeline), sizeof(long long) * sc::kTlKernels * sc::kTlWaves * 4));
  return SC_OK;
}
#endif

// ---- timing -----------------------------------------------------------------------------------

static int harvest(sc_ctx* c) {
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto& e : c->ev_used) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      c->ms[e.k] += ms;
      c->launches[e.k] += 1;
    }
    c->ev_free.push_back(e);
  }
  c->ev_used.clear();
  return SC_OK;
}

int sc_enable_timing(sc_ctx* c, int on) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  if (!on && c->timing) harvest(c);
  c->timing = on != 0;
  return SC_OK;
}

int sc_reset_timing(sc_ctx* c) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  int rc = harvest(c);
  for (int k = 0; k < SC_NUM_KERNELS; ++k) {
    c->ms[k] = 0;
    c->launches[k] = 0;
  }
  return rc;
}

int sc_get_timing(sc_ctx* c, double* ms, int64_t* launches) {
  if (!c) return fail(SC_ERR_ARG, "null context");
  int rc = harvest(c);
  for (int k = 0; k < SC_NUM_KERNELS; ++k) {
    if (ms) ms[k] = c->ms[k];
    if (launches) launches[k] = c->launches[k];
  }
  return rc;
}

}  // extern "C"
