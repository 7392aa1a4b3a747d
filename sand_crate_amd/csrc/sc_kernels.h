// HIP kernels of the SandCrate particle update for gfx950 (MI355X).  Included once by
// sandcrate_hip.hip.  Built with -ffp-contract=off: every decision the reference makes in
// float64 (row index, wall contact, neighbor predicate, segment crossing) is evaluated with the
// reference's operation order and without fused multiply-add, so it comes out bit-identical.
#pragma once
#include "sc_device.h"

namespace sc {

// np.clip(t, 0, 1): NaN passes through, like NumPy's minimum/maximum.
__device__ __forceinline__ double clip01(double t) { return t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t); }

// ------------------------------------------------------------------------------------------
// K0  append: crate.py:138-147 (create_new_particles).  Host arrays are P x 2 interleaved.
// ------------------------------------------------------------------------------------------
__global__ void k_append(const double* __restrict__ xy, const double* __restrict__ vxy, int m, int first_id,
                         int* __restrict__ counters, double* __restrict__ x, double* __restrict__ y,
                         double* __restrict__ vx, double* __restrict__ vy, int* __restrict__ id, int reset) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  int base = reset ? 0 : counters[C_NS];
  if (k < m) {
    x[base + k] = xy[2 * k];
    y[base + k] = xy[2 * k + 1];
    vx[base + k] = vxy[2 * k];
    vy[base + k] = vxy[2 * k + 1];
    id[base + k] = first_id + k;
  }
  // every block reads `base` before any block may bump the counter: the bump happens in a
  // separate one-thread launch (k_bump) ordered after this kernel on the stream.
}

__global__ void k_bump(int* counters, int m, int reset) { counters[C_NS] = (reset ? 0 : counters[C_NS]) + m; }

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
__global__ void __launch_bounds__(kBlock) k_wall_bin(World w, int* __restrict__ counters, double* __restrict__ x,
                                                     double* __restrict__ y, int* __restrict__ cellS,
                                                     int* __restrict__ wslotS, int* __restrict__ cellCount,
                                                     double* __restrict__ wrec) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counters[C_NS]) return;
  double px = x[i], py = y[i];
  if (px < w.lo || px > w.hi || py < w.lo || py > w.hi) {  // crate.py:152
    cellS[i] = -1;
    return;
  }
  // bounding-box reject (exact-safe: the boxes are inflated far beyond rounding error)
  unsigned cand = 0;
  bool far = true;
  for (int k = 0; k < w.nseg; ++k) {
    Seg s = w.seg[k];
    double ox = fmax(fmax(fmin(s.ax, s.bx) - px, px - fmax(s.ax, s.bx)), 0.0);
    double oy = fmax(fmax(fmin(s.ay, s.by) - py, py - fmax(s.ay, s.by)), 0.0);
    if (ox <= w.far_box && oy <= w.far_box) far = false;
    if (ox <= w.touch_box && oy <= w.touch_box) cand |= 1u << k;
  }
  int wslot = far ? -1 : -2;
  if (cand) {
    double cpx[kMaxSeg], cpy[kMaxSeg], ux[kMaxSeg], uy[kMaxSeg];
    unsigned touch = 0;
    int V = 0;
    for (int k = 0; k < w.nseg; ++k) {
      if (!(cand >> k & 1u)) continue;
      Seg s = w.seg[k];
      // geometry_utils.py:26-38, same operation order
      double abx = s.bx - s.ax, aby = s.by - s.ay;
      double apx = px - s.ax, apy = py - s.ay;
      double t = (apx * abx + apy * aby) / (abx * abx + aby * aby);
      t = clip01(t);
      double cx = abx * t + s.ax, cy = aby * t + s.ay;
      double dx = cx - px, dy = cy - py;
      double s2 = dx * dx + dy * dy;
      if (s2 <= w.t_wall) {  // == (sqrt(s2) <= r * 1.2), crate.py:229
        cpx[V] = cx;
        cpy[V] = cy;
        ux[V] = (px - cx) * 2;  // crate.py:234
        uy[V] = (py - cy) * 2;
        touch |= 1u << k;
        ++V;
      }
    }
    if (V > 0) {
      // crate.py:73-85: every body with n_b touching segments overwrites slots [0, n_b)
      double velx[kMaxSeg], vely[kMaxSeg];
      for (int k = 0; k < V; ++k) velx[k] = vely[k] = 0.0;
      int seg0 = 0;
      for (int b = 0; b < w.nbody; ++b) {
        BodyK bd = w.body[b];
        unsigned mask = (bd.nseg >= 32 ? 0xFFFFFFFFu : ((1u << bd.nseg) - 1u)) << seg0;
        int nb = __popc(touch & mask);
        seg0 += bd.nseg;
        for (int k = 0; k < nb; ++k) {
          velx[k] = bd.vx + (cpy[k] - bd.py) * bd.omega;
          vely[k] = bd.vy + (-(cpx[k] - bd.px)) * bd.omega;
        }
      }
      double Ux = 0, Uy = 0, Cx = 0, Cy = 0, fx = 0, fy = 0;
      for (int k = 0; k < V; ++k) {
        Ux += ux[k];
        Uy += uy[k];
        Cx += velx[k];
        Cy += vely[k];
        double rel = w.r / sqrt(ux[k] * ux[k] + uy[k] * uy[k]);  // crate.py:206-208
        if (rel < 0.5) rel = 0.5;
        fx += ux[k] * (rel - 0.5);
        fy += uy[k] * (rel - 0.5);
      }
      px += fx;  // crate.py:211
      py += fy;
      x[i] = px;
      y[i] = py;
      wslot = atomicAdd(&counters[C_WREC], 1);
      double* rec = wrec + 5 * (size_t)wslot;
      rec[0] = Ux;
      rec[1] = Uy;
      rec[2] = Cx;
      rec[3] = Cy;
      rec[4] = (double)V;
    }
  }
  if (!(px == px) || !(py == py)) {  // crate.py:206: distance 0 to a wall gives NaN
    atomicOr(&counters[C_FLAGS], F_NAN);
    cellS[i] = -1;
    return;
  }
  double fr = floor(py / w.d), fc = floor(px / w.d);  // collision_detector.py:126
  long long lr = (long long)fr - w.row0, lc = (long long)fc - w.col0;
  if (!(fabs(fr) < 9e15) || !(fabs(fc) < 9e15) || lr < 1 || lr > w.nrows - 2 || lc < 1 || lc > w.ncols - 2) {
    atomicOr(&counters[C_FLAGS], F_OUT_OF_GRID);
    cellS[i] = -1;
    return;
  }
  int c = (int)lr * w.ncols + (int)lc;
  cellS[i] = c;
  wslotS[i] = wslot;
  atomicAdd(&cellCount[c], 1);
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
  int incl = sum;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
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
  if (blockIdx.x == nblocks - 1 && threadIdx.x == 0) {
    int tot = off + blockSums[nblocks - 1];
    out[n] = tot;  // one-past-the-end entry: cellStart[ncells]
    if (total_out) *total_out = tot;
  }
}

// ------------------------------------------------------------------------------------------
// K3  scatter: a slot inside the particle's cell bucket, in arrival order.  The returning
// atomic counts the bucket back down to zero, so cellCount needs no clearing for the next tick.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_scatter(const int* __restrict__ counters, const int* __restrict__ cellS,
                                                    const int* __restrict__ cellStart, int* __restrict__ cellCount,
                                                    int* __restrict__ perm) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counters[C_NS]) return;
  int c = cellS[i];
  if (c < 0) return;
  int pos = cellStart[c] + atomicSub(&cellCount[c], 1) - 1;
  perm[pos] = i;
}

// ------------------------------------------------------------------------------------------
// K4  reorder: final slot = bucket start + rank of (x, id) inside the bucket, which makes the
// whole array sorted by (row, x, id) = np.lexsort((x, y_floored)) with its stable tie-break
// (collision_detector.py:127).  Moves the particle's state to the sorted arrays.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_reorder(const int* __restrict__ counters, const int* __restrict__ perm, const int* __restrict__ cellS,
              const int* __restrict__ cellStart, const int* __restrict__ wslotS, const double* __restrict__ xS,
              const double* __restrict__ yS, const double* __restrict__ vxS, const double* __restrict__ vyS,
              const int* __restrict__ idS, double* __restrict__ xT, double* __restrict__ yT, double* __restrict__ vxT,
              double* __restrict__ vyT, int* __restrict__ idT, int* __restrict__ cellT, int* __restrict__ wslotT) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= counters[C_NT]) return;
  int i = perm[s];
  int c = cellS[i];
  double xi = xS[i];
  int idi = idS[i];
  int b = cellStart[c], e = cellStart[c + 1];
  int rank = 0;
  for (int t = b; t < e; ++t) {
    int j = perm[t];
    double xj = xS[j];
    int idj = idS[j];
    rank += (xj < xi) || (xj == xi && idj < idi);
  }
  int dst = b + rank;
  xT[dst] = xi;
  yT[dst] = yS[i];
  vxT[dst] = vxS[i];
  vyT[dst] = vyS[i];
  idT[dst] = idi;
  cellT[dst] = c;
  wslotT[dst] = wslotS[i];
}

// ------------------------------------------------------------------------------------------
// K5  neighbor lists in the reference's canonical order (collision_detector.py:9-121):
//   [same row, to the right, x ascending] [row+1, x ascending]
//   [same row, to the left, x descending] [row-1, x descending], cut at 20.
// In the (row, x, id)-sorted array each of the three rows' candidates (columns c-1..c+1) is one
// contiguous range.  The forward predicates are the reference's own (:106-119, :75-80); a
// reverse edge j->i exists exactly when i is a forward candidate of j (:85-88).
// Lists are written slot-major (nbr[s*cap + i]) so that lanes store/load consecutive words.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_neighbors(World w, int* __restrict__ counters, const double* __restrict__ x, const double* __restrict__ y,
                const int* __restrict__ cell, const int* __restrict__ cellStart, int* __restrict__ nbr,
                unsigned char* __restrict__ cnt, int cap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int n = counters[C_NT];
  int count = 0;
  if (i < n) {
    int c = cell[i];
    double xi = x[i], yi = y[i];
    double xhi = xi + w.d, xlo = xi - w.d;
    int e0 = cellStart[c + 2];
    for (int j = i + 1; j < e0 && count < kMaxNbr; ++j) {  // same strip, after i (:106-109)
      double xj = x[j];
      if (xj > xhi) break;
      double dx = xj - xi, dy = y[j] - yi;
      if (dx * dx + dy * dy <= w.t_nbr) nbr[(size_t)count++ * cap + i] = j;
    }
    int b1 = cellStart[c + w.ncols - 1], e1 = cellStart[c + w.ncols + 2];
    for (int j = b1; j < e1 && count < kMaxNbr; ++j) {  // next strip (:112-119)
      double xj = x[j];
      if (xj > xhi) break;
      if (xj >= xlo) {
        double dx = xj - xi, dy = y[j] - yi;
        if (dx * dx + dy * dy <= w.t_nbr) nbr[(size_t)count++ * cap + i] = j;
      }
    }
    int b0 = cellStart[c - 1];
    for (int j = i - 1; j >= b0 && count < kMaxNbr; --j) {  // reverse edges from the same strip
      double xj = x[j];
      if (!(xi <= xj + w.d)) break;
      double dx = xj - xi, dy = y[j] - yi;
      if (dx * dx + dy * dy <= w.t_nbr) nbr[(size_t)count++ * cap + i] = j;
    }
    int bm = cellStart[c - w.ncols - 1], em = cellStart[c - w.ncols + 2];
    for (int j = em - 1; j >= bm && count < kMaxNbr; --j) {  // reverse edges from the previous strip
      double xj = x[j];
      if (!(xi <= xj + w.d)) break;
      if (xi >= xj - w.d) {
        double dx = xj - xi, dy = y[j] - yi;
        if (dx * dx + dy * dy <= w.t_nbr) nbr[(size_t)count++ * cap + i] = j;
      }
    }
    cnt[i] = (unsigned char)count;
  }
  int s = wave_sum(count), m = wave_max(count);
  if ((threadIdx.x & 63) == 0 && s) {
    unsigned old = atomicAdd((unsigned*)&counters[C_SUMC], (unsigned)s);
    if (old + (unsigned)s < old) atomicAdd(&counters[C_SUMC_HI], 1);
    atomicMax(&counters[C_MAXC], m);
  }
}

// host-noise mode: exclusive scan of C_i in particle-id order gives each particle's offset into
// the host's rand(sum C_i, 2) block (crate.py:165-170 draws particle by particle in index order).
__global__ void k_count_by_id(const int* __restrict__ counters, const int* __restrict__ id,
                              const unsigned char* __restrict__ cnt, int* __restrict__ cntById) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < counters[C_NT]) cntById[id[i]] = cnt[i];
}

// Collider offset eta_ij of crate.py:169 for (particle id, slot).
template <int NOISE>
__device__ __forceinline__ void collider_noise(const World& w, int id, int slot, const double* __restrict__ eta,
                                               int off, double& ex, double& ey) {
  if (NOISE == SC_NOISE_NONE) {
    ex = ey = 0.0;
  } else {
    double ux, uy;
    if (NOISE == SC_NOISE_HOST) {
      ux = eta[2 * ((size_t)off + slot)];
      uy = eta[2 * ((size_t)off + slot) + 1];
    } else {
      noise_u01(w.noise_key, id, slot, ux, uy);
    }
    ex = (ux - 0.5) * w.d * w.level;
    ey = (uy - 0.5) * w.d * w.level;
  }
}

// ------------------------------------------------------------------------------------------
// K6  pass A "density + normals": populate_colliders (crate.py:161-175), compute_particle_pressures
// (:261-275) and pass 1 of apply_tension (:337-342).  Writes P_i and s_i.
// ------------------------------------------------------------------------------------------
template <int NOISE>
__global__ void __launch_bounds__(kBlock)
    k_density(World w, const int* __restrict__ counters, const double* __restrict__ x, const double* __restrict__ y,
              const int* __restrict__ id, const int* __restrict__ nbr, const unsigned char* __restrict__ cnt, int cap,
              const double* __restrict__ eta, const int* __restrict__ offById, double* __restrict__ P,
              double* __restrict__ sx, double* __restrict__ sy) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counters[C_NT]) return;
  int C = cnt[i];
  double xi = x[i], yi = y[i];
  int idi = (NOISE == SC_NOISE_NONE) ? 0 : id[i];
  int off = (NOISE == SC_NOISE_HOST) ? offById[idi] : 0;
  double sumw = 0, ax = 0, ay = 0;
  for (int s = 0; s < C; ++s) {
    int j = nbr[(size_t)s * cap + i];
    double ex, ey;
    collider_noise<NOISE>(w, idi, s, eta, off, ex, ey);
    double rx = xi - (x[j] + ex), ry = yi - (y[j] + ey);  // crate.py:167-171
    double dist = sqrt(rx * rx + ry * ry);
    double nx = rx / dist, ny = ry / dist;                 // crate.py:174
    double ov = 1 - clip01(dist / w.d);                    // crate.py:270
    sumw += ov;
    double t = (1 - ov) * ov;                              // crate.py:342
    ax += t * nx;
    ay += t * ny;
  }
  P[i] = C ? fmax(0.0, sumw - w.ignored) : 0.0;  // crate.py:265-273
  sx[i] = ax;
  sy[i] = ay;
}

// ------------------------------------------------------------------------------------------
// K7  pass B "force + integrate", fused: pass 2 of apply_tension (crate.py:343-353), apply_gravity
// (:309-310), apply_pressure (:295-307), apply_viscosity (:316-323), apply_wall_bounce (:245-259),
// apply_continuous_collision_velocity_fix (:177-200; geometry_utils.py:136-143, :182-222) and
// apply_particles_velocity (:360-361).  Reads the sorted arrays, writes the storage arrays in
// the same (sorted) order: that is the next tick's input.
// ------------------------------------------------------------------------------------------
template <int NOISE>
__global__ void __launch_bounds__(kBlock)
    k_force(World w, int* __restrict__ counters, const double* __restrict__ x, const double* __restrict__ y,
            const double* __restrict__ vx, const double* __restrict__ vy, const int* __restrict__ id,
            const int* __restrict__ wslot, const int* __restrict__ nbr, const unsigned char* __restrict__ cnt, int cap,
            const double* __restrict__ eta, const int* __restrict__ offById, const double* __restrict__ P,
            const double* __restrict__ sx, const double* __restrict__ sy, const double* __restrict__ wrec,
            double* __restrict__ xo, double* __restrict__ yo, double* __restrict__ vxo, double* __restrict__ vyo,
            int* __restrict__ ido) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int n = counters[C_NT];
  if (i == 0) counters[C_NS] = n;  // the storage arrays now hold the n live particles
  if (i >= n) return;
  int C = cnt[i];
  double xi = x[i], yi = y[i];
  double vxi = vx[i], vyi = vy[i];
  double Pi = P[i], sxi = sx[i], syi = sy[i];
  int idi = id[i];
  int off = (NOISE == SC_NOISE_HOST) ? offById[idi] : 0;
  double tx = 0, ty = 0, qx = 0, qy = 0, ux = 0, uy = 0;
  for (int s = 0; s < C; ++s) {
    int j = nbr[(size_t)s * cap + i];
    double ex, ey;
    collider_noise<NOISE>(w, idi, s, eta, off, ex, ey);
    double rx = xi - (x[j] + ex), ry = yi - (y[j] + ey);
    double dist = sqrt(rx * rx + ry * ry);
    double nx = rx / dist, ny = ry / dist;
    double Pj = P[j];
    double align = ((sxi - sx[j]) * nx + (syi - sy[j]) * ny) * w.ss;  // crate.py:347-349
    double fix = Pj + Pi - 2 * w.tp;                                   // crate.py:351
    double k = align + fix;
    tx += k * nx;
    ty += k * ny;
    double pp = Pi + Pj;  // crate.py:301-304
    qx += nx * pp;
    qy += ny * pp;
    ux += vx[j];  // crate.py:175 snapshot of the neighbors' start-of-tick velocities
    uy += vy[j];
  }
  vxi += w.dt * tx;  // crate.py:352
  vyi += w.dt * ty;
  vxi += w.dt * w.gx;  // crate.py:310
  vyi += w.dt * w.gy;
  int ws = wslot[i];
  double Ux = 0, Uy = 0, Cx = 0, Cy = 0, V = 0;
  if (ws >= 0) {
    const double* rec = wrec + 5 * (size_t)ws;
    Ux = rec[0];
    Uy = rec[1];
    Cx = rec[2];
    Cy = rec[3];
    V = rec[4];
    qx += Ux * Pi;  // wall colliders carry pressure 0 and are not normalised (crate.py:286-293)
    qy += Uy * Pi;
  }
  double dpa = w.dt * w.pamp;
  vxi += dpa * qx;  // crate.py:306
  vyi += dpa * qy;
  double dv = w.dt * w.visc;  // crate.py:319-323: sum_j (v0_j - v_i) with v_i the current velocity
  vxi += dv * (ux - C * vxi);
  vyi += dv * (uy - C * vyi);
  if (ws >= 0) {  // crate.py:245-259
    double nx = Ux / V, ny = Uy / V;
    double nn = sqrt(nx * nx + ny * ny);
    nx /= nn;
    ny /= nn;
    double cvx = Cx / V, cvy = Cy / V;
    double q = (vxi - cvx) * nx + (vyi - cvy) * ny;
    if (q < 0) {
      double cx = -1 * q * nx, cy = -1 * q * ny;
      vxi += cx;
      vyi += cy;
      vxi += cx * w.decay;
      vyi += cy * w.decay;
    }
  }
  // continuous collision: movement p -> p + v*dt against the 2S padded segments
  double mx = vxi * w.dt, my = vyi * w.dt;
  if (!(ws == -1 && mx * mx + my * my < w.ccd_skip2)) {
    double bx = xi + mx, by = yi + my;   // crate.py:183-184
    double abx = bx - xi, aby = by - yi;  // geometry_utils.py:205 uses (b - a)
    double fac = 1.0;
    for (int m = 0; m < 2 * w.nseg; ++m) {
      Seg s = w.pad[m];
      double dcx = s.bx - s.ax, dcy = s.by - s.ay;
      if (!(dcy * abx + (-dcx) * aby < 0)) continue;  // opposite_direction_map (:205)
      // orientation(p,q,r) = sign((q.y-p.y)*(r.x-q.x) - (q.x-p.x)*(r.y-q.y))  (:212-222)
      double o1 = (by - yi) * (s.ax - bx) - (bx - xi) * (s.ay - by);  // (a,b,c)
      double o2 = (by - yi) * (s.bx - bx) - (bx - xi) * (s.by - by);  // (a,b,d)
      double o3 = (s.by - s.ay) * (xi - s.bx) - (s.bx - s.ax) * (yi - s.by);  // (c,d,a)
      double o4 = (s.by - s.ay) * (bx - s.bx) - (s.bx - s.ax) * (by - s.by);  // (c,d,b)
      int g1 = (o1 > 0) - (o1 < 0), g2 = (o2 > 0) - (o2 < 0), g3 = (o3 > 0) - (o3 < 0), g4 = (o4 > 0) - (o4 < 0);
      bool n1 = o1 != o1, n2 = o2 != o2, n3 = o3 != o3, n4 = o4 != o4;  // np.sign(nan) = nan, nan != x
      if ((g1 != g2 || n1 || n2) && (g3 != g4 || n3 || n4)) {
        // calc_collision_point(a, ab = v*dt, c, cd): cross(a-c, cd) / cross(cd, ab)  (:141-143)
        double acx = xi - s.ax, acy = yi - s.ay;
        double f = (acx * dcy - acy * dcx) / (dcx * my - dcy * mx);
        if (f < fac) fac = f;  // crate.py:198-199
      }
    }
    vxi *= fac;  // crate.py:200
    vyi *= fac;
  }
  xo[i] = xi + w.dt * vxi;  // crate.py:361
  yo[i] = yi + w.dt * vyi;
  vxo[i] = vxi;
  vyo[i] = vyi;
  ido[i] = idi;
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
