/*
 * sandcrate_hip.h -- C ABI of libsandcrate_hip.so: the MI355X (gfx950) implementation of
 * SandCrate's per-timestep particle update.
 *
 * The reference has no FFI (it is pure Python/NumPy); this header is the boundary a
 * maintainer binds with ctypes (INTEGRATION.md shows the stub).  Each entry point names the
 * reference code it replaces as  file:line  relative to the reference repository.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SC_ERR_* otherwise; the message of the
 *     last failure on the calling thread is sc_last_error().  Nothing throws across the ABI and
 *     nothing calls back into the host language.
 *   - host arrays are caller-allocated, C-contiguous, float64 / int64 / int32 exactly as NumPy
 *     holds them (particles are P x 2 interleaved x,y like crate.py:24-25); the library owns all
 *     device memory.  Device state is float64 SoA (x, y, vx, vy) in cell-sorted order plus a
 *     per-particle id that remembers the reference's particle index order.
 *   - one context per GPU, one calling thread per context.  Calls are enqueued on the context's
 *     HIP stream and return before the GPU finishes unless the description says "synchronises".
 */
#ifndef SANDCRATE_HIP_H
#define SANDCRATE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 5
#define SC_MAX_NEIGHBORS 20 /* collision_detector.py:6  MAX_ALLOWED_NEIGHBORS */
#define SC_MAX_SEGMENTS 16  /* wall segments of all rigid bodies together (scenes use 6 and 8) */
#define SC_MAX_BODIES 8

enum {
  SC_OK = 0,
  SC_ERR_ARG = -1,      /* bad argument */
  SC_ERR_HIP = -2,      /* HIP runtime failure, see sc_last_error() */
  SC_ERR_CAPACITY = -3, /* more particles / cells than the context was created for */
  SC_ERR_STATE = -4,    /* call order violated (e.g. sc_step_finish without sc_step_begin) */
  SC_ERR_DOMAIN = -5    /* a particle fell outside the cell grid (NaN or runaway position) */
};

/* Collider noise of crate.py:169 ((rand(C_i,2) - 0.5) * diameter * collider_noise_level). */
enum {
  SC_NOISE_NONE = 0,    /* eta = 0 (what collider_noise_level = 0 gives) */
  SC_NOISE_HOST = 1,    /* uniforms supplied per tick by sc_set_noise_host: the host's MT19937 stream */
  SC_NOISE_COUNTER = 2  /* counter-based hash of (seed, tick, particle id, slot); no host traffic */
};

/* Live-editable coefficients: the YAML keys of config/ *.yaml:10-22 that the tick reads
 * (crate.py:55-57).  spring_* are inert in the reference (crate.py:117-118) and absent here. */
typedef struct sc_params {
  double dt;
  double particle_radius;
  double wall_collision_decay;
  double pressure_amplifier;
  double ignored_pressure;
  double collider_noise_level;
  double viscosity;
  double surface_smoothing;
  double target_pressure;
  double gravity_x;
  double gravity_y;
} sc_params;

/* One rigid body after RigidBody.apply_velocity (rigid_body.py:42-46, :64-68): what
 * calc_body_points_velocities (rigid_body.py:28-34) needs, and how many of the stacked
 * segments (crate.py:69-71) belong to it. */
typedef struct sc_body {
  double position_x, position_y;
  double center_velocity_x, center_velocity_y;
  double angular_clockwise_velocity;
  int32_t n_segments;
  int32_t reserved;
} sc_body;

typedef struct sc_stats {
  int64_t particles;      /* P after remove_particles (crate.py:149-159) */
  int64_t neighbor_slots; /* sum of C_i: how many (rand, rand) pairs crate.py:169 draws this tick */
  int32_t max_neighbors;  /* max C_i */
  int32_t wall_particles; /* particles with at least one wall contact (crate.py:229) */
  int32_t flags;          /* nonzero: SC_ERR_DOMAIN condition seen on the device */
  int32_t reserved;
} sc_stats;

typedef struct sc_ctx sc_ctx;

const char* sc_last_error(void);
int sc_abi_version(void);

/* Lifetime.  capacity = most particles the context will ever hold (Crate: max_particles). */
int sc_create(int device, int64_t capacity, sc_ctx** out);
int sc_destroy(sc_ctx* ctx);
/* Run on this hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) instead of the context's
 * own stream; NULL is HIP's default stream, as everywhere in HIP.  sc_use_own_stream goes back to
 * the private non-blocking stream the context was created with.  Both synchronise the old stream. */
int sc_set_stream(sc_ctx* ctx, void* hip_stream);
int sc_use_own_stream(sc_ctx* ctx);

/* State in/out.  Replaces direct assignment of Crate.particles / particle_velocities
 * (crate.py:24-25).  Particle i gets id i; ids order ties exactly like the reference's
 * array index does (collision_detector.py:127 lexsort is stable). */
int sc_upload_state(sc_ctx* ctx, const double* xy, const double* vxy, int64_t n);
/* crate.py:138-147 create_new_particles: appended particles get the next ids. */
int sc_append_particles(sc_ctx* ctx, const double* xy, const double* vxy, int64_t n);
/* Synchronises.  Number of live particles. */
int sc_count(sc_ctx* ctx, int64_t* n);
/* Synchronises.  Writes live particles in id order (= the reference's array order): xy, vxy are
 * n x 2, pressure (crate.py:275 particles_pressure) and ids are n.  Any pointer may be NULL.
 * n_capacity is the room in the host arrays; *n_out is what was written. */
int sc_download_state(sc_ctx* ctx, double* xy, double* vxy, double* pressure, int64_t* ids,
                      int64_t n_capacity, int64_t* n_out);

/* Per-tick inputs.  Coefficients are re-sent every tick because the viewer edits them live
 * (playback.py:221-226); segments because bodies move (crate.py:363-365). */
int sc_set_params(sc_ctx* ctx, const sc_params* p);
/* segments: n_segments x 2 x 2 (crate.py:69-71); padded: 2*n_segments x 2 x 2 from pad_segments
 * (geometry_utils.py:146-172), computed on the host because it is O(S). */
int sc_set_segments(sc_ctx* ctx, const double* segments, const double* padded, int32_t n_segments,
                    const sc_body* bodies, int32_t n_bodies);
int sc_set_noise_mode(sc_ctx* ctx, int mode, uint64_t seed);

/* The tick: crate.py:91-129 from remove_particles on.
 *   sc_step_begin  : remove_particles (:149-159), calc_virtual_colliders + hard wall fix (:97-99,
 *                    :202-243), strip sort + neighbor lists (collision_detector.py:9-49)
 *   sc_step_stats  : synchronises; P and sum C_i, so the host can draw rand(sum C_i, 2)
 *   sc_set_noise_host: those uniforms, (n_pairs x 2) in particle-index order, slot-minor (:169)
 *   sc_step_finish : populate_colliders ... apply_particles_velocity (:103-125)
 * sc_step(ctx, k) = k x (begin, finish) with no synchronisation; not valid in SC_NOISE_HOST mode. */
/* Look-ahead for back-to-back ticks (optional; between sc_step_begin and sc_step_finish).  Declares the
 * coefficients and walls of the tick AFTER the one being finished.  sc_step_finish then also performs
 * that next tick's remove_particles / calc_virtual_colliders / apply_hard_wall_fix and the bucket counts
 * (crate.py:93, :97-99) in the epilogue of the force kernel, while the new position is still in
 * registers, and the next sc_step_begin skips them: one launch and one pass over the positions less per
 * tick.  The promise is binding: the next tick must be started with exactly these inputs and without
 * appending particles in between, otherwise sc_step_begin / sc_append_particles return SC_ERR_STATE.
 * (Crate.physics_tick() never promises -- the viewer may edit coefficients between ticks; Crate.run()
 * and sc_step(ctx, k > 1) do.)  Not used with slabs. */
int sc_set_next_inputs(sc_ctx* ctx, const sc_params* p, const double* segments, int32_t n_segments,
                       const sc_body* bodies, int32_t n_bodies);
int sc_step_begin(sc_ctx* ctx);
int sc_step_stats(sc_ctx* ctx, sc_stats* out);
int sc_set_noise_host(sc_ctx* ctx, const double* u01, int64_t n_pairs);
int sc_step_finish(sc_ctx* ctx);
int sc_step(sc_ctx* ctx, int32_t n_ticks);
/* One whole tick in ONE call, for drivers whose per-call overhead matters (ctypes: ~4 us per call):
 *   sc_set_params(now) + sc_set_segments(now) + sc_step_begin + [sc_set_next_inputs(next)] + sc_step_finish.
 * `next` may be NULL (no look-ahead).  Same errors as the calls it stands for; in SC_NOISE_HOST mode
 * valid only when the device holds the stream (sc_rng_set_state) -- otherwise the host has to draw the noise
 * between begin and finish (crate.py:169). */
typedef struct sc_tick_inputs {
  sc_params params;
  const double* segments; /* n_segments x 2 x 2 (crate.py:69-71) */
  const double* padded;   /* 2 n_segments x 2 x 2 (geometry_utils.py:146-172) */
  const sc_body* bodies;
  int32_t n_segments, n_bodies;
} sc_tick_inputs;
int sc_tick(sc_ctx* ctx, const sc_tick_inputs* now, const sc_tick_inputs* next);
int sc_synchronize(sc_ctx* ctx);

/* The bucket scan is one pass with decoupled look-back: a workgroup waits for the totals of the workgroups before it,
 * which the dispatcher starts first.  The wait is bounded -- `polls` attempts per predecessor (default 2^22; negative:
 * give up at once, for tests) -- and a workgroup that gives up abandons the TICK: its later kernels do nothing, the particles
 * stay as the tick found them, and the next synchronising call returns SC_ERR_HIP.  (A knob for tests of that path.) */
int sc_set_scan_patience(sc_ctx* ctx, int64_t polls);

/* Parity taps, valid between sc_step_begin and sc_step_finish.  Synchronise.  All arrays have one
 * entry per sorted slot k = 0..P-1 (the order of collision_detector.py:127):
 *   y_floored[k]  row index floor(y/d) (collision_detector.py:126)
 *   ids[k]        particle id in that slot (= sorted_indices when ids are array indices)
 *   counts[k]     C of that particle, neighbors[k*20 + s] the id of its s-th neighbor or -1
 *   fixed_xy[k*2] position after apply_hard_wall_fix (crate.py:202-211) */
int sc_download_sort(sc_ctx* ctx, int64_t* y_floored, int64_t* ids, int64_t n_capacity, int64_t* n_out);
int sc_download_neighbors(sc_ctx* ctx, int64_t* ids, int32_t* counts, int64_t* neighbors, double* fixed_xy,
                          int64_t n_capacity, int64_t* n_out);
/* After sc_step_finish: surface normals s_i of apply_tension pass 1 (crate.py:337-342) in id order. */
int sc_download_normals(sc_ctx* ctx, double* sxy, int64_t n_capacity, int64_t* n_out);

/* Stand-alone forms of two reference functions (its tests/test_distance.py pins both).
 * sc_neighbor_search = detect_particle_collisions (collision_detector.py:9-49) on arbitrary
 * coordinates: y_floored/sorted_indices per sorted slot, counts[i] and table[i*20+s] per ORIGINAL
 * index i, -1 padded.  sc_points_to_segments = points_to_segments_distance
 * (geometry_utils.py:7-39): nearest is n x s x 2, distances n x s. */
int sc_neighbor_search(int device, const double* xy, int64_t n, double diameter, int64_t* y_floored,
                       int64_t* sorted_indices, int32_t* counts, int64_t* table);
int sc_points_to_segments(int device, const double* xy, int64_t n, const double* segments, int32_t n_segments,
                          double* nearest, double* distances);
/* pad_segments (geometry_utils.py:146-172) on the host, the reference's operations in its order: `padded` receives
 * 2 * n_segments segments, first every (a + o, b + o), then every (b - o, a - o), o = cw90(b - a) * pad_distance / |b - a|.
 * No GPU involved: the padded twins of a moving wall are a kernel argument of every tick (sc_set_segments). */
int sc_pad_segments(const double* segments, int32_t n_segments, double pad_distance, double* padded);

/* Kernel timing with HIP events on the context's stream.  While enabled every kernel launch is
 * bracketed by two events; sc_get_timing synchronises and returns, per kernel, the summed
 * milliseconds and the number of launches since sc_reset_timing.  Names: sc_kernel_name(i). */
#define SC_NUM_KERNELS 12
int sc_enable_timing(sc_ctx* ctx, int on);
int sc_reset_timing(sc_ctx* ctx);
int sc_get_timing(sc_ctx* ctx, double* ms /*[SC_NUM_KERNELS]*/, int64_t* launches /*[SC_NUM_KERNELS]*/);
const char* sc_kernel_name(int index);

/* Multi-GPU slabs (of columns, or of rows: sc_set_slab_axis) (no reference counterpart).  One context per GPU owns the
 * grid columns [col_lo, col_hi), column = floor(x / diameter) of the position a particle has when
 * the tick starts.  Particles within `halo` columns outside the slab are ghosts: they take part in
 * the wall fix, the neighbor search and pass A exactly like owned particles, are never integrated,
 * and are dropped at the end of the tick.  Ids are global (sc_upload_state_ids), so tie-breaks and
 * the counter-based noise are the same as on one GPU.  Not available with SC_NOISE_HOST.
 *
 * Per tick:  sc_halo_pack -> exchange the two buffers with the neighbors (RCCL send/recv on
 * the stream given to sc_set_stream, or any transport) -> sc_halo_unpack of the two received buffers
 * (either pointer may be NULL at a domain edge) -> the tick.  Buffers are DEVICE memory of
 * (capacity_records + 1) * 5 doubles, caller-owned (e.g. torch tensors) and zero-initialised: record 0 is
 * a header whose first 32-bit word is the record count, records 1.. are (x, y, vx, vy, id).
 * sc_halo_pack writes every stored particle within `halo` columns of the left / right edge, including
 * particles that have already moved out of the slab on that side (migrants: the receiver owns them from
 * this tick on).  sc_halo_unpack also re-arms the headers of the send buffers, which must therefore have
 * been sent (in stream order) by then.  Nothing here synchronises.
 *
 * Look-ahead: the buffers given to the last sc_halo_pack stay bound to the context.  A tick whose successor
 * was promised (sc_set_next_inputs / sc_tick with `next`) packs the successor's halo message in the epilogue
 * of its force kernel, and the following sc_halo_unpack runs the removal / wall pass for what it appends; the
 * steady-state slab tick is then:  exchange -> sc_halo_unpack -> sc_tick(now, next).  Calling sc_halo_pack
 * for a tick that was packed this way is refused (SC_ERR_STATE). */
int sc_set_slab(sc_ctx* ctx, int64_t col_lo, int64_t col_hi, int32_t halo, int32_t has_left, int32_t has_right);
/* Which way the domain is cut: axis 0 (the default) -- slabs are ranges of COLUMNS floor(x / d), "left" / "right"
 * are the neighbors towards smaller / larger x; axis 1 -- ranges of ROWS floor(y / d), neighbors towards smaller /
 * larger y (sc_set_slab's col_lo / col_hi, the histogram of sc_column_histogram and `halo` then count rows).  The
 * sorted order is row-major, so with rows the halo bands are the first and last few blocks of it: the blocks
 * that may pack halo records (sc_set_halo_overlap) are a percent of all instead of a third.  Call before
 * sc_set_slab; results do not depend on the axis. */
int sc_set_slab_axis(sc_ctx* ctx, int32_t axis);
int sc_upload_state_ids(sc_ctx* ctx, const double* xy, const double* vxy, const int64_t* ids, int64_t n);
/* crate.py:138-147 under slabs: every rank draws the same new particles (same host stream) and appends the ones whose
 * column / row it owns, under their global ids. */
int sc_append_particles_ids(sc_ctx* ctx, const double* xy, const double* vxy, const int64_t* ids, int64_t n);
int sc_halo_pack(sc_ctx* ctx, double* dev_left, double* dev_right, int64_t capacity_records);
/* Message sizes.  A message need not carry the whole buffer: sc_halo_sizes gives, for the exchange of the coming
 * tick, the number of records (after the header record) to send to / receive from each side -- the count the same
 * direction had six ticks earlier plus 50 % and 1024 records, in steps of 256, at most capacity_records.  Sender
 * and receiver of a message derive it from the same number (what was packed = what the received header said;
 * sc_halo_unpack publishes both in host-mapped memory), so the two ends agree without talking; the lag exceeds
 * the number of ticks the host may run ahead of the device, so nothing synchronises.  Whole buffers for the
 * first ticks after an upload or sc_set_slab.  sc_halo_unpack is told how many records each message carried; a
 * header that announces more sets the halo-overflow condition (SC_ERR_CAPACITY at the next synchronising call). */
int sc_halo_sizes(sc_ctx* ctx, int64_t capacity_records, int64_t* send_left_records, int64_t* recv_left_records,
                  int64_t* send_right_records, int64_t* recv_right_records);
int sc_halo_unpack(sc_ctx* ctx, const double* dev_from_left, int64_t left_records, const double* dev_from_right,
                   int64_t right_records);
/* Slab re-balancing: stored live particles per grid column floor(x / diameter), columns clamped into
 * [col0, col0 + n_columns).  Every particle is stored live on exactly one rank, so the ranks' histograms add up
 * to the global one, from which all ranks derive the same new cuts (sc_set_slab; the next halo exchange moves
 * the particles that changed owner).  Synchronises. */
int sc_column_histogram(sc_ctx* ctx, int64_t col0, int32_t n_columns, int64_t* histogram);
/* RCCL transport for the exchange step (optional: any transport that moves the buffers between the
 * calls above will do; sand_crate_amd.slab falls back to torch.distributed P2P ops).  librccl is dlopen()ed
 * on first use -- the copy already loaded in the process if any, else `rccl_path`, else the default search
 * path -- so the library itself has no link-time dependency on it.
 *   sc_comm_available  0 when librccl can be loaded in this process: every rank checks (and the ranks agree on
 *                      the answer) BEFORE any of them enters the collective sc_comm_init
 *   sc_comm_unique_id  rank 0: 128 bytes to hand to every rank (ncclGetUniqueId)
 *   sc_comm_init       collective over the `world` contexts of the slab chain (ncclCommInitRank); rank = slab index
 *   sc_halo_exchange   on the context's stream, one group: send `send_left` to / receive `recv_left` from
 *                      rank `left_rank`, the same on the right; a negative rank means no neighbor on that side.
 *                      Each message is (records + 1) * 5 doubles from the start of its buffer (sc_halo_sizes).
 * A failing RCCL call returns SC_ERR_HIP with RCCL's message in sc_last_error(). */
int sc_comm_available(const char* rccl_path);
int sc_comm_unique_id(const char* rccl_path, void* id_128_bytes);
int sc_comm_init(sc_ctx* ctx, const char* rccl_path, const void* id_128_bytes, int32_t rank, int32_t world);
int sc_comm_destroy(sc_ctx* ctx);
int sc_halo_exchange(sc_ctx* ctx, const double* send_left, int64_t send_left_records, double* recv_left,
                     int64_t recv_left_records, int32_t left_rank, const double* send_right, int64_t send_right_records,
                     double* recv_right, int64_t recv_right_records, int32_t right_rank);

/* Force monitor: the reference's HUD shows the mean |dv| of each force phase (force_monitor.py:13-37 around
 * crate.py:110-123).  While enabled, the force kernel also sums |dv| per particle and phase -- tension, gravity,
 * pressure, viscosity, wall_bounce, continuous_collision, in this order -- without changing any result (ticks are
 * then never fused with their successor's wall pass).  sc_get_force_monitor returns the six sums and the number
 * of particles summed since the last call and clears them; it synchronises. */
int sc_enable_force_monitor(sc_ctx* ctx, int on);
int sc_get_force_monitor(sc_ctx* ctx, double* sums_6, int64_t* particles);

/* Checkpoint.  sc_checkpoint_begin copies the stored state (positions, velocities, ids, counters, the MT19937
 * stream if the device holds it) device-to-device on the context's stream and sends the copy to pinned host memory
 * on a side stream; it returns at once and later ticks overlap the transfer.  sc_checkpoint_finish waits for that
 * transfer only and delivers the particles in particle-index order with the tick they belong to, the id the next
 * emitted particle gets, and the generator state (rng_position = -1: the host holds the stream).  One checkpoint
 * at a time.  sc_restore_counters, after sc_upload_state_ids on a fresh context, puts tick and next id back. */
int sc_checkpoint_begin(sc_ctx* ctx);
int sc_checkpoint_finish(sc_ctx* ctx, double* xy, double* vxy, int64_t* ids, int64_t room, int64_t* n_out, int64_t* tick,
                         int64_t* next_id, uint32_t* rng_key_624, int32_t* rng_position);
int sc_restore_counters(sc_ctx* ctx, int64_t tick, int64_t next_id);

/* NumPy's legacy global generator on the device (the reference draws particle sources and collider noise from
 * `np.random`, seeded in Crate.__init__, crate.py:22).  sc_rng_set_state hands the stream to the context -- the
 * 624-word key and the position of `np.random.get_state()` -- and from then on
 *   sc_emit_particles  runs ParticleSource.generate_particles (particle_source.py:17-24) for the given sources
 *                      on the device: binomial(flow, dt) new particles each (legacy inversion branch; SC_ERR_DOMAIN
 *                      if flow * dt > 30 or dt > 0.5), rand(n, 2) position jitter, rand(n, 2) velocity noise,
 *                      capped at max_particles minus the stored count, appended with the next ids;
 *   sc_step_finish     in SC_NOISE_HOST mode without a sc_set_noise_host call draws the tick's rand(sum C_i, 2)
 *                      block on the device,
 * bit for bit the numbers NumPy would have produced, with no count readback and no upload.  sc_rng_get_state
 * (synchronises) returns the stream to the host, e.g. for `np.random.set_state`. */
typedef struct sc_source {
  double radius, position_x, position_y, velocity_x, velocity_y, noise;
  int64_t flow;
} sc_source;
int sc_rng_set_state(sc_ctx* ctx, const uint32_t* key_624, int32_t position);
int sc_rng_get_state(sc_ctx* ctx, uint32_t* key_624, int32_t* position);
int sc_emit_particles(sc_ctx* ctx, const sc_source* sources, int32_t n_sources, double dt, int64_t max_particles);

/* Halo overlap (BASELINE.json configs[4]: "halo overlap on side HIP stream").  With it on, a tick whose successor
 * was promised runs its force kernel in two launches: first the blocks that hold a particle within the halo band
 * plus two columns of a cut -- the only ones that can pack halo records -- then the interior blocks; the exchange
 * of the coming tick waits for the first launch only and runs on the context's side stream next to the second.
 *   sc_halo_exchange        does both waits itself (RCCL calls go to the side stream);
 *   sc_halo_overlap_begin   for other transports: the side stream waits for what the coming message depends on
 *                           (`peer`, optional: and for what that context's message depends on -- in-process chains);
 *                           the caller then enqueues its copies / sends on sc_side_stream;
 *   sc_halo_overlap_end     the context's stream waits for the side stream; sc_halo_unpack follows as usual.
 * A particle of an interior block that ends the tick inside a band after all (it moved more than the margin: two
 * columns / eight rows) is not silently lost: SC_ERR_DOMAIN at the next synchronising call.
 *
 * sc_set_band_flag (slabs of rows only): instead of two launches the force kernel runs as ONE whose first workgroups
 * take the blocks at both ends of the sorted order (where the band blocks are); the last of them to finish publishes
 * a flag, and the side stream waits for it with a one-thread polling kernel (hipStreamWaitValue32 is not usable
 * here).  Costs ~4 us per tick instead of ~24 -- PROVIDED the side stream has a hardware queue of its own: a process
 * with more streams than hardware queues may put the polling kernel in front of the very kernel it waits for, which
 * then costs the poll's time-out (50 ms, reported as SC_ERR_HIP).  Off by default; bench.py tries it and keeps it when
 * it is faster. */
int sc_set_halo_overlap(sc_ctx* ctx, int on);
int sc_set_band_flag(sc_ctx* ctx, int on);
int sc_side_stream(sc_ctx* ctx, void** hip_stream);
int sc_halo_overlap_begin(sc_ctx* ctx, sc_ctx* peer);
int sc_halo_overlap_end(sc_ctx* ctx);

/* Synchronises.  Live particles stored in this context (dead ghost copies excluded); summed over
 * the ranks this is the global particle count. */
int sc_owned_count(sc_ctx* ctx, int64_t* n);

#ifdef __cplusplus
}
#endif
#endif /* SANDCRATE_HIP_H */
