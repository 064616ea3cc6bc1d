/*
 * target_batch_c.h -- batched extension of the target_manager C boundary (not in the reference).
 *
 * The reference steps one target per call under a manager-wide mutex
 * (src/target_manager.cpp:190-225).  At 10^4..10^6 targets the per-id calls cannot be the fast
 * path, so this header adds (a) array-of-ids entry points with host buffers and (b) a
 * device-resident dense path whose inputs already live in HBM (no host staging; this is what
 * bench.py times and what a multi-GPU caller uses, one manager per rank).
 *
 * Plain C ABI: pointers and sizes only.  "dev" pointers are HIP device pointers (e.g.
 * torch.Tensor.data_ptr()).  Unless stated, functions return 0 on success and a negative
 * error code (message on stderr) on failure; they never throw.
 */
#ifndef TARGET_ESTIMATION_AMD_TARGET_BATCH_C_H
#define TARGET_ESTIMATION_AMD_TARGET_BATCH_C_H

#include "target_manager_c.h"

typedef void target_batch_c; /* all targets of one (model, P layout) inside a manager */

/* model ids = the reference enum TargetManager::target_t (target_manager.hpp:38) */
#define TARGET_ANGULAR_RATES 0
#define TARGET_ANGULAR_VELOCITIES 1
#define TARGET_UNIFORM_ACCELERATION 2
#define TARGET_UNIFORM_VELOCITY 3
/* add to lanes_per_target (1, 2, 3 or 6) to store P symmetric-packed (upper triangle) in HBM */
#define TARGET_LAYOUT_SYMMETRIC_PACKED 100
/* lanes_per_target = 1 + this: store only the entries of P inside an axis group (exact when Q, R and
 * P0 do not couple different axes, as in every shipped model file; refused otherwise) */
#define TARGET_LAYOUT_AXIS_SEPARABLE 200
/* lanes_per_target = 1 + this: axis-separable with each group block stored as its upper triangle */
#define TARGET_LAYOUT_AXIS_SEPARABLE_PACKED 300
#define TARGET_DTYPE_F64 0
#define TARGET_DTYPE_F32 1

#ifdef __cplusplus
extern "C" {
#endif

/* ---- construction ---------------------------------------------------------------------- */
/* file may be NULL (no default model: use the *_typed initialisers, as the reference's
 * default-constructed TargetManager, target_manager.cpp:106-109).  dtype: TARGET_DTYPE_*.
 * lanes_per_target: 0 = automatic, checked per init call:
 *   - Q, R and P0 do not couple different axis groups: the axis-separable layout -- with each group block
 *     stored as its upper triangle (1 + TARGET_LAYOUT_AXIS_SEPARABLE_PACKED) when the matrices are also
 *     exactly symmetric, with full group blocks (1 + TARGET_LAYOUT_AXIS_SEPARABLE) otherwise;
 *   - coupled but exactly symmetric: the dense kernel on the upper triangle of P
 *     (G + TARGET_LAYOUT_SYMMETRIC_PACKED, G = lanes per target: 1 or 3 depending on the model);
 *   - anything else: the dense kernel on the full P with the tuned G of (model, dtype).
 * Explicit: 1, 2, 3 or 6 (dense, full P), G + TARGET_LAYOUT_SYMMETRIC_PACKED with G = 1, 2, 3 or 6 where
 * the model allows it (upper triangle of P only in HBM: 44 % less traffic), or one of the two separable
 * codes.  With a packed layout P is symmetric by construction, whereas the reference's (I-KC)P is
 * symmetric only to rounding; the full layouts reproduce that rounding-level asymmetry (the
 * axis-separable full-block layout is bit-identical to the dense kernel). */
target_manager_c* target_manager_new_ex(const char* file, int dtype, int lanes_per_target);
/* all launches of this manager go to `hip_stream` (a hipStream_t; NULL = default stream) */
int target_manager_set_stream(target_manager_c* self, void* hip_stream);
int target_manager_synchronize(target_manager_c* self);
/* rt_logger-equivalent snapshots (the reference publishes measurement / pose / twist / acceleration / covariance per
 * target, src/target_interface.cpp:32-40): with a directory set (here or by env TARGET_ESTIMATION_LOG_DIR) every
 * target_manager_log() appends one row per selected target to <dir>/time_<id>, meas_pose_<id>, est_pose_<id>,
 * est_twist_<id> (the files of the reference's test and plot script, test/target_manager_test.cpp:164-168,
 * matlab/plot_target_manager_test.m:9-13) and pose_<id> ([xyz rpy]), est_acc_<id>, covariance_<id> (P, row-major), in
 * the text format of the reference's writeTxtFile (utils.hpp:96-120).  Selected = target_manager_set_log_targets, or
 * every target while there are at most 64; otherwise one <channel>_all file per channel with the id in front of every
 * row.  Files stay open between calls; one buffered write per file per call.  NULL or "" switches logging off (log()
 * is then a no-op, as the reference without LOGGER_ON).  A directory switches the measured-pose rows on. */
int target_manager_set_log_directory(target_manager_c* self, const char* dir);
int target_manager_set_log_targets(target_manager_c* self, const unsigned int* ids, long n);   /* n == 0: automatic */
/* TargetInterface::getMeasuredPose (include/target_estimation/target_interface.hpp:130, src/target_interface.cpp:117-121,
 * :142-146): the last measurement of a target, [0 0 0 0 0 0 1] before the first.  The filters never read it back, so it is
 * kept only on request: one row of 7 doubles per target beside the records, written by a small kernel behind every
 * step (56 B per measured target per tick; off by default).  The getter returns false for an unknown id or when the
 * rows are not kept.  Linear models fed through target_batch_step_host (x y z only) report the identity orientation. */
int target_manager_set_keep_measurement(target_manager_c* self, int on);
bool target_manager_get_measured_pose(target_manager_c* self, unsigned int id, double* pose7);
/* TargetInterface::getPeriodEstimate (target_interface.hpp:94, src/target_interface.cpp:80-87): 2 pi / |omega| of the
 * current twist, -1 if the target is not rotating */
bool target_manager_get_period_estimate(target_manager_c* self, unsigned int id, double* period);
/* TargetInterface::getEstimatedTransform (target_interface.hpp:106, src/target_interface.cpp:95-98): the isometry T_ as
 * a row-major 4x4 matrix [R t; 0 0 0 1] */
bool target_manager_get_estimated_transform(target_manager_c* self, unsigned int id, double* T16);
/* TargetInterface::getN / getM (target_interface.hpp:142,148); 0 for an unknown id */
int target_manager_get_n(target_manager_c* self, unsigned int id);
int target_manager_get_m(target_manager_c* self, unsigned int id);
/* getTarget(id)->getEstimator()->getQ() / getR() / getP0() (include/target_estimation/kalman.hpp:74,79,89): the matrices
 * the target was created with, row-major doubles (Q, P0: n*n; R: m*m); any pointer may be NULL.  false: unknown id, or
 * (P0 only) the manager saw more than 4096 distinct P0 matrices for this model and stopped mirroring them. */
bool target_manager_get_model_matrices(target_manager_c* self, unsigned int id, double* Q, double* R, double* P0);
const char* target_manager_last_error(void);

/* TargetManager::init(type,id,dt0,t0,Q,R,P0,p0,v0,a0), target_manager.hpp:85-87.
 * Q, P0: n*n row-major; R: m*m; v0, a0 may be NULL (zeros). */
int target_manager_init_typed(target_manager_c* self, int type, unsigned int id, double dt0, double t0,
                              const double* Q, const double* R, const double* P0, const double* p0,
                              const double* v0, const double* a0);
/* n targets at once with the manager's default model; p0 [n][7], v0/a0 [n][6] or NULL.
 * Existing ids are skipped.  Returns the number created (or < 0). */
long target_manager_init_batch(target_manager_c* self, const unsigned int* ids, long n, double dt0, double t0,
                               const double* p0, const double* v0, const double* a0);
long target_manager_init_batch_typed(target_manager_c* self, int type, const unsigned int* ids, long n, double dt0,
                                     double t0, const double* Q, const double* R, const double* P0,
                                     int per_target_P0, const double* p0, const double* v0, const double* a0);
/* n targets whose model parameters come from a table of n_classes sets: Q [n_classes][ns*ns], R [n_classes][m*m],
 * P0 [n_classes][ns*ns] (row-major), class_of [n] = the row of target i.  The reference lets every target carry its own
 * Q, R, P0 (TargetManager::init, target_manager.hpp:85-87); here all classes of one layout live in ONE batch -- a (Q, R)
 * table in HBM plus a class index per target -- so a tick is still one launch per motion model, however many classes
 * there are.  (The one-set initialisers above do the same: a new (Q, R) for a model joins that model's batch as a new
 * class.)  Existing ids are skipped.  Returns the number created (or < 0). */
long target_manager_init_batch_classes(target_manager_c* self, int type, const unsigned int* ids, long n, double dt0, double t0,
                                       long n_classes, const double* Q, const double* R, const double* P0,
                                       const unsigned int* class_of, const double* p0, const double* v0, const double* a0);
/* TargetManager::erase, target_manager.cpp:227-241.  1 = erased, 0 = unknown id. */
int target_manager_erase(target_manager_c* self, unsigned int id);
/* the same for n ids in one call: one compaction launch per batch instead of one launch per target
 * (a timeout storm in the ingest stage).  Unknown or repeated ids print the reference's message and are
 * skipped.  Returns the number erased (or < 0). */
long target_manager_erase_batch(target_manager_c* self, const unsigned int* ids, long n);
long target_manager_size(target_manager_c* self);
/* TargetManager::getAvailableTargets (ascending ids), target_manager.cpp:126-133 */
long target_manager_get_available_targets(target_manager_c* self, unsigned int* ids_out, long capacity);

/* ---- array-of-ids calls, host buffers ---------------------------------------------------- */
/* for i < n: has_meas[i] ? update(ids[i], dt, meas[i]) : update(ids[i], dt); meas [n][7] may be
 * NULL (predict only), has_meas may be NULL (all measured).  An id that appears twice is stepped twice,
 * in order, as the reference's loop would.
 * Calls of node-tick size (up to 1024 ids, every batch of the manager at most 16384 targets) are queued like the
 * reference's one-target calls and run as one launch at the next read; target_manager_get_est_batch of that size
 * reads the host-resident rows that launch wrote (40 targets: 23 us per update + read-back, 400: 45 us; larger calls
 * stage their arrays and resolve ids on the device from 8192; TE_SMALL_BATCH_QUEUE).  Same results.
 * Returns the number of known ids stepped. */
long target_manager_update_meas_batch(target_manager_c* self, const unsigned int* ids, long n, double dt,
                                      const double* meas, const unsigned char* has_meas);
/* TargetManager::update(dt): predict every target, target_manager.cpp:220-225 */
int target_manager_update_all(target_manager_c* self, double dt);
/* any of pose [n][7], twist [n][6], acceleration [n][6], found [n] may be NULL */
long target_manager_get_est_batch(target_manager_c* self, const unsigned int* ids, long n, double* pose,
                                  double* twist, double* acceleration, unsigned char* found);
/* TargetInterface::getEstimatedPose(t1) / Twist(t1) / Acceleration(t1) (src/types, target_interface.cpp:123-140) */
long target_manager_get_est_at_batch(target_manager_c* self, const unsigned int* ids, long n, double t1,
                                     double* pose, double* twist, double* acceleration, unsigned char* found);
/* getTarget(id)->getEstimator()->getState()/getP() for ids of ONE model: x [n][ns], P [n][ns*ns]
 * row-major.  Returns the state size ns, or < 0. */
long target_manager_get_state_batch(target_manager_c* self, const unsigned int* ids, long n, double* x, double* P);
/* TargetInterface::getTime() */
int target_manager_get_time(target_manager_c* self, unsigned int id, double* t);

/* ---- sphere intersection (IntersectionSolver, src/intersection_solver.cpp:42-104) -------------- */
/* Time from t1 (absolute) to the first crossing of the sphere |p - origin| = radius along the
 * extrapolated trajectory p + v d + a d^2/2 (smallest real root of the quartic; -1 if none, if
 * it is negative, if the target has zero acceleration -- uniform_velocity and angular_velocities
 * targets never intersect in the reference -- or if the id is unknown).
 * Replaces IntersectionSolver::getIntersectionTimeWithSphere, intersection_solver.cpp:42-89. */
double target_manager_get_intersection_time_with_sphere(target_manager_c* self, unsigned int id, double t1,
                                                         const double* origin, double radius);
/* Pose at t1 + delta ([0 0 0 0 0 0 1] if none); returns whether an intersection exists.  Replaces
 * IntersectionSolver::getIntersectionPoseWithSphere, intersection_solver.cpp:91-104, WITHOUT its
 * moving-average convergence gate (:105-120; the gated form is target_manager_intersect_sphere_converged_batch below).
 * delta may be NULL. */
bool target_manager_get_intersection_pose_with_sphere(target_manager_c* self, unsigned int id, double t1,
                                                      const double* origin, double radius, double* pose,
                                                      double* delta);
/* the same for n ids: delta [n], pose [n][7] or NULL, found [n] or NULL */
long target_manager_intersect_sphere_batch(target_manager_c* self, const unsigned int* ids, long n, double t1,
                                           const double* origin, double radius, double* delta, double* pose,
                                           unsigned char* found);

/* The query above followed by the reference's convergence gate, kept PER TARGET (the reference keeps
 * one per IntersectionSolver object): two moving averages of window filters_length (<= 0: the reference
 * default 250, intersection_solver.hpp:63) over the distance and the quaternion angle between
 * consecutive intersection poses; converged[i] = both filtered errors <= their thresholds
 * (IntersectionSolver::getIntersectionPoseWithSphere, intersection_solver.cpp:91-124).  A query with no
 * intersection leaves the gate untouched and reports 0.  filtered_errors [n][2] may be NULL.
 * Costs 2 * filters_length doubles of HBM per target (allocated on first use). */
long target_manager_intersect_sphere_converged_batch(target_manager_c* self, const unsigned int* ids, long n, double t1,
                                                     double pos_th, double ang_th, const double* origin, double radius,
                                                     int filters_length, double* delta, double* pose,
                                                     unsigned char* converged, unsigned char* found, double* filtered_errors);

/* The reference's IntersectionSolver as an OBJECT (include/target_estimation/intersection_solver.hpp:56-126): a manager
 * handle plus ONE convergence gate -- two moving averages of window filters_length (default 250, :63) and the previous
 * intersection pose -- shared by every id queried through it, as the reference's members are (src/intersection_solver.cpp:
 * 19-40).  ..._get_time_with_sphere = getIntersectionTimeWithSphere (:73, .cpp:42-89); ..._get_pose_with_sphere =
 * getIntersectionPoseWithSphere (:86, .cpp:91-124): pose7 = the pose at t1 + delta, [0 0 0 0 0 0 1] if there is no
 * intersection; returns whether both filtered errors are within their thresholds.  The solver does not own the manager,
 * which must outlive it.  ..._last_errors: the filtered errors of the last call that found an intersection. */
typedef void target_intersection_solver_c;
target_intersection_solver_c* target_intersection_solver_new(target_manager_c* manager, unsigned int filters_length);
void target_intersection_solver_delete(target_intersection_solver_c* solver);
double target_intersection_solver_get_time_with_sphere(target_intersection_solver_c* solver, unsigned int id, double t1,
                                                        const double* origin, double radius);
bool target_intersection_solver_get_pose_with_sphere(target_intersection_solver_c* solver, unsigned int id, double t1, double pos_th,
                                                     double ang_th, const double* origin, double radius, double* pose7);
void target_intersection_solver_last_errors(target_intersection_solver_c* solver, double* pos_error_filtered, double* ang_error_filtered);

/* ---- device-resident dense path ---------------------------------------------------------- */
int target_manager_num_batches(target_manager_c* self);
target_batch_c* target_manager_get_batch(target_manager_c* self, int index);
target_batch_c* target_manager_get_batch_of_type(target_manager_c* self, int type);
long target_batch_size(target_batch_c* b);
int target_batch_type(target_batch_c* b);
int target_batch_dtype(target_batch_c* b);
int target_batch_state_dim(target_batch_c* b);
int target_batch_meas_dim(target_batch_c* b);
int target_batch_lanes_per_target(target_batch_c* b);
int target_batch_is_symmetric_packed(target_batch_c* b);
/* 0 full P, 1 symmetric-packed, 2 axis-separable, 3 axis-separable with symmetric-packed groups */
int target_batch_layout(target_batch_c* b);
/* number of distinct (Q, R) parameter classes among the batch's targets */
int target_batch_num_classes(target_batch_c* b);
/* bytes one predict+update cycle of one target must move ((2n + 2n^2 + 7 (+6)) * w, or
 * (2n + n(n+1) + 7 (+6)) * w for a symmetric-packed batch, (2n + 2 sum(group^2) + 7 (+6)) * w for an
 * axis-separable one) */
long target_batch_algorithmic_bytes(target_batch_c* b);
/* HBM bytes actually allocated per target (record incl. tile padding) */
double target_batch_resident_bytes_per_target(target_batch_c* b);
/* slot -> id of the dense order (slot i is row i of every dense array) */
long target_batch_slot_ids(target_batch_c* b, unsigned int* ids_out, long capacity);
/* One predict(+update) tick over every target of the batch, asynchronous on the manager's stream.
 *   meas_dev    : SoA [7][ld] in the batch precision (row c = component c of [x y z qx qy qz qw]
 *                 for all slots), or NULL for predict-only.  Linear models read rows 0..2 only.
 *   has_meas_dev: per-slot bytes, or NULL (every slot has a measurement). */
int target_batch_step(target_batch_c* b, double dt, const void* meas_dev, long ld, const unsigned char* has_meas_dev);
/* One tick of every slot from HOST measurements in SoA form and in the batch precision: row c of meas_soa_host
 * (ld_host elements per row) = component c of [x y z qx qy qz qw] for slots 0..size-1.  Only the rows the model
 * reads are copied (3 for the linear models, 7 for the angular ones): 12-24 B per target over PCIe instead of the
 * 56 B of target_manager_update_meas_batch's double[n][7], and no conversion kernel.  has_meas_host [size] or
 * NULL.  Returns after the step has been enqueued and the host arrays have been consumed. */
int target_batch_step_host(target_batch_c* b, double dt, const void* meas_soa_host, long ld_host, const unsigned char* has_meas_host);
/* n_ticks consecutive ticks = n_ticks launches of the step kernel, enqueued in one call: tick s
 * reads meas_dev + s * tick_stride elements (and has_meas_dev + s * has_stride bytes).  With
 * use_graph != 0 the launches are recorded once into a hipGraph (keyed by the arguments) and
 * replayed: for replaying recorded streams at small batch sizes, where the tick is launch-bound.
 * use_graph == 2 records the graph and launches nothing (set-up before a timed region). */
int target_batch_step_sequence(target_batch_c* b, long n_ticks, double dt, const void* meas_dev, long tick_stride, long ld,
                               const unsigned char* has_meas_dev, long has_stride, int use_graph);
/* The same with the measurements (and masks) held as a RING of ring_ticks ticks: tick s reads ring entry
 * s % ring_ticks, so one recorded graph may hold several passes over the ring (fewer graph launches). */
int target_batch_step_sequence_ring(target_batch_c* b, long n_ticks, double dt, const void* meas_dev, long tick_stride, long ld,
                                    const unsigned char* has_meas_dev, long has_stride, long ring_ticks, int use_graph);
/* The same n_ticks ticks in ONE launch: each target's state stays in registers across the ticks and
 * only the measurements are read per tick (temporal fusion; identical results).  For replaying
 * recorded streams; its throughput is an "effective" figure, not comparable with one launch per tick.
 * Batches that have no fused kernel -- several (Q, R) classes in the batch, or one of the few (model, precision, layout)
 * combinations whose fused form would spill -- are served tick by tick inside the call: same results, one launch per tick. */
int target_batch_step_fused(target_batch_c* b, long n_ticks, double dt, const void* meas_dev, long tick_stride, long ld,
                            const unsigned char* has_meas_dev, long has_stride);
/* RESIDENT ("live") mode for small batches -- a live stream without a dispatch per tick.  A 10^4-target tick is shorter than
 * the 1.5-2 us a dependent kernel launch costs, so at BASELINE.json configs[1] / configs[2] the per-tick launch IS the tick.
 * Here ONE launch stays on the device with the batch's state in registers and serves tick after tick as the host posts them:
 *   ..._live_start   launch.  meas_ring_dev / has_ring_dev: a ring of ring_ticks ticks in device memory, laid out as for
 *                    target_batch_step_sequence; tick k of the session reads entry (first_entry + k) % ring_ticks, which must
 *                    hold tick k's measurements BEFORE tick k is posted (written by a copy on another stream, or in advance;
 *                    the kernel reads them past the caches).  max_ticks bounds the session; idle_limit_s (<= 0: 10 s) is how
 *                    long a wavefront waits without news before it gives up by itself.
 *   ..._live_post    n more ticks are in the ring: one store to the doorbell word the resident kernel polls -- in device memory,
 *                    written through the PCIe BAR, where the system has a large BAR (the GPU then polls a local word instead of
 *                    reading host memory over PCIe every round: a paced tick of 10^5 targets 8.0 -> 6.6 us from C++), in host-mapped
 *                    memory otherwise or with TE_LIVE_DOORBELL=host.  Returns at once.
 *   ..._live_done    ticks that EVERY wavefront has finished;  ..._live_wait: spin until that reaches `tick` (0) or timeout (1)
 *   ..._live_stop    finish the posted ticks, write the records back, end the launch; returns the ticks served
 * Results equal those of single ticks bit for bit.  While a session is open the records in HBM are stale: any other call on
 * the batch (steps, getters, erase, init) ends the session first.  Needs the automatic layout of the shipped models
 * (axis-separable, packed groups), one (Q, R) class and a batch small enough to be fully resident: ..._live_capacity targets
 * (on an MI355X: 3.1 * 10^5 targets for the linear models, 1.8 to 2.5 * 10^5 for the angular ones in either precision -- the fp64
 * angular kernels park part of the record in LDS; a session with the per-tick query or pose output runs a larger kernel: 1.1 to 3.1 * 10^5;
 * profiles/r04_live_capacity.txt).  ..._live_start returns once the resident kernel
 * is known to run (its last workgroup says so) and fails after 2 s otherwise; the kernel runs on a high-priority stream of the
 * library's own, so that no stream of the caller shares its hardware queue (work queued behind an endless kernel waits for its end).  Posting faster than the device serves is fine: a wavefront that is behind catches up
 * without polling.
 * Sessions of one PROCESS share the device: the library counts the resident sessions of the process (all managers) and refuses a
 * start whose wavefronts do not fit next to the ones already resident ("do not fit the device together") instead of launching a
 * grid that can only start in part.  A session of ANOTHER process is not seen: there the 2 s start check is what reports it.
 * While any session of the process is resident, calls that would FREE device memory (a batch that grows, measured-pose rows
 * switched off, large get_state requests, a destroyed manager) do not call hipFree -- it synchronises the device and would block
 * until that session ends -- but put the memory on a list that is freed when the last session has left. */
int target_batch_live_start(target_batch_c* b, double dt, const void* meas_ring_dev, long tick_stride, long ld,
                            const unsigned char* has_ring_dev, long has_stride, long ring_ticks, long first_entry, long max_ticks,
                            double idle_limit_s);
/* Per-tick pose output of the sessions started AFTER this call (also through target_manager_live_start_all): pose_soa_dev =
 * device memory, SoA [7][ld] doubles (row c = component c of [x y z qx qy qz qw] for slots 0..size-1), NULL = off.  The resident
 * kernel writes the estimated pose of every target after every tick -- what the reference's node publishes every tick
 * (src/target_manager_ros.cpp:78-87) -- through the caches and before the tick counts as done: copy the buffer on another stream
 * after ..._live_done reached tick k and it holds the poses of a tick >= k (exactly k when one tick is posted at a time). */
int target_batch_live_set_pose_output(target_batch_c* b, double* pose_soa_dev, long ld);
int target_batch_live_post(target_batch_c* b, long n_ticks);
/* n_ticks doorbells of ONE tick each, back to back (what a caller's loop of ..._live_post(b, 1) does, without its call overhead) */
int target_batch_live_post_each(target_batch_c* b, long n_ticks);
long target_batch_live_done(target_batch_c* b);
int target_batch_live_wait(target_batch_c* b, long tick, double timeout_s);
long target_batch_live_stop(target_batch_c* b);
long target_batch_live_capacity(target_batch_c* b);
/* 1 while the session's resident kernel is there (serving or waiting), 0 once it has left -- after ..._live_stop, or by itself after
 * idle_limit_s without news (the records are back in HBM then; ..._live_stop still has to be called to close the session) */
int target_batch_live_running(target_batch_c* b);
/* n_ticks ticks of EVERY batch of the manager in one call (BASELINE.json configs[3]/[4]: several motion
 * models per GPU, optionally with the per-tick sphere query "fused on-GPU").  per_batch[i] describes
 * batch i (target_manager_get_batch order): measurements as for target_batch_step_sequence, plus the
 * device outputs of the query (delta [size], pose [size][7] or NULL; overwritten every tick) when
 * query != 0.  The query is the own-time one (t1 = each target's current time).  Batches are
 * independent.  When every non-empty batch is a one-class batch in the axis-separable layout with packed
 * groups (the automatic choice for the shipped model files) and there are at least two, a tick of ALL of
 * them is ONE launch (csrc/kf_step_sep.hpp kf_step_population_kernel; target_manager_population_tick says
 * whether that holds): nothing then depends on which hardware queues the runtime gives to concurrent
 * streams.  Otherwise there is a launch per batch per tick; with use_graph != 0 the batches' chains are
 * recorded as parallel branches of ONE hipGraph (use_graph == 2: record only).  Results are identical to
 * calling target_batch_step (+ target_batch_intersect_sphere_dev with a NaN t1) per batch per tick. */
typedef struct target_batch_sequence_c {
  const void* meas_dev; long tick_stride; long ld;
  const unsigned char* has_meas_dev; long has_stride;
  double* delta_dev; double* pose_dev;
  long ring_ticks;   /* > 0: meas_dev / has_meas_dev hold a ring of that many ticks, tick s reads entry s % ring_ticks; 0: linear */
} target_batch_sequence_c;
int target_manager_step_sequence_all(target_manager_c* m, long n_ticks, double dt, const target_batch_sequence_c* per_batch,
                                     long n_batches, int query, const double* origin, double radius, int use_graph);
/* 1 if target_manager_step_sequence_all currently steps all batches with one launch per tick, 0 if with a launch per
 * batch, -1 on error (semantics served: every target of every model each tick, src/target_manager.cpp:190-225) */
int target_manager_population_tick(target_manager_c* m);
/* Resident ("live") mode (target_batch_live_* above) for EVERY batch of a manager at once (BASELINE.json configs[3] / configs[4]: two motion models per GPU, whose per-GPU
 * share is launch-bound): one resident kernel per batch, each on a stream of its own so that they are on the device together;
 * per_batch[i] describes batch i's ring as for target_manager_step_sequence_all (ring_ticks > 0); with query != 0 the own-time
 * sphere query of every target runs after every tick inside the resident kernels, into per_batch[i].delta_dev / pose_dev
 * (overwritten every tick), as target_manager_step_sequence_all's fused query does; like the pose output above they are written
 * THROUGH the caches before the tick counts as done: copy them on another stream after ..._live_done_all reached tick k and they
 * hold the results of a tick >= k (exactly k when one tick is posted at a time).  The
 * sessions' wavefronts must fit the device together.  ..._post_all: n_ticks to every batch (one_doorbell_per_tick != 0: as
 * n_ticks separate doorbells); ..._done_all: ticks finished by every wavefront of every batch; ..._stop_all: ticks served. */
int target_manager_live_start_all(target_manager_c* m, double dt, const target_batch_sequence_c* per_batch, long n_batches, long first_entry,
                                  long max_ticks, double idle_limit_s, int query, const double* origin, double radius);
int target_manager_live_post_all(target_manager_c* m, long n_ticks, int one_doorbell_per_tick);
long target_manager_live_done_all(target_manager_c* m);
int target_manager_live_wait_all(target_manager_c* m, long tick, double timeout_s);
long target_manager_live_stop_all(target_manager_c* m);
/* derived outputs of every slot into device arrays of doubles ([size][7], [size][6], [size][6];
 * any may be NULL); at_time != 0 extrapolates to t1 */
int target_batch_get_est_dev(target_batch_c* b, double* pose_dev, double* twist_dev, double* acc_dev, int at_time, double t1);
/* every slot of the batch: delta_dev [size], pose_dev [size][7] or NULL (device doubles); a NaN t1
 * means "each target's own current time" (the per-tick fused query of BASELINE.json configs[4]);
 * origin is a host array of 3 */
int target_batch_intersect_sphere_dev(target_batch_c* b, double t1, const double* origin, double radius,
                                      double* delta_dev, double* pose_dev);
/* the gated form for every slot; delta_dev and pose_dev are required (the gate reads them) */
int target_batch_intersect_sphere_converged_dev(target_batch_c* b, double t1, double pos_th, double ang_th,
                                                const double* origin, double radius, int filters_length,
                                                double* delta_dev, double* pose_dev, unsigned char* converged_dev);
/* The convergence gate alone (IntersectionSolver::getIntersectionPoseWithSphere, src/intersection_solver.cpp:102-120),
 * fed with query results already on the device -- e.g. the delta / pose arrays the per-tick query of
 * target_manager_step_sequence_all leaves behind: delta_dev [size], pose_dev [size][7] -> converged_dev [size];
 * filtered_dev [size][2] (filtered position / angle error) and variance_dev [size][2]
 * (MovingAvgFilter::getVariance, utils.hpp:241-251; O(filters_length) per target, NULL skips it) are optional. */
int target_batch_gate_update_dev(target_batch_c* b, const double* delta_dev, const double* pose_dev, double pos_th, double ang_th,
                                 int filters_length, unsigned char* converged_dev, double* filtered_dev, double* variance_dev);
/* AoS doubles [n][7] (host layout of the reference) -> SoA [7][ld] in the batch precision, on device */
int target_batch_pack_meas_dev(target_batch_c* b, const double* meas_aos_dev, long n, void* meas_soa_dev, long ld);

/* ---- synthetic measurement streams (SURVEY 8d "Synthetic inputs") ------------------------------------------ */
/* Counter-based generator: every number is a pure function of (seed, target, tick, component) -- splitmix64 keys, Box-Muller
 * normals with fixed-sequence log / sin / cos -- so the GPU fills its measurement ring with one kernel and a CPU checker
 * regenerates the same doubles bit for bit (csrc/stream_gen.hpp has the definition; it widens the generator of the
 * reference's integration test, test/target_manager_test.cpp:82-115, to a population: per-target start, velocity,
 * acceleration (uniform_acceleration), body rate; xyz + N(0, 0.01^2); quaternion = Qtran(dt, omega)^(tick+1) applied to the
 * identity).  Target i of a call is target first_target + i of the stream (ranks of a sharded run pass their offset).
 *   availability < 1 : a target has a measurement on a tick with that probability (has_meas_dev receives the bytes)
 *   rpy_noise  > 0   : the measured orientation is the true one times a small random rotation (rad, per axis) */
typedef struct target_stream_c {
  int model;                 /* TARGET_* motion model */
  unsigned long long seed;
  long first_target;
  double dt;                 /* tick length: tick s is time (s + 1) dt */
  double availability;       /* 1 = every target measured on every tick */
  double rpy_noise;          /* 0 = noiseless quaternion, as in the reference's test */
} target_stream_c;
/* ticks first_tick .. first_tick + n_ticks - 1 for n_targets targets: meas_dev = SoA ring [n_ticks][7][ld] (tick_stride
 * elements between ticks) in precision dtype (TARGET_DTYPE_*; values are generated in double and rounded once);
 * has_meas_dev [n_ticks][has_stride] bytes or NULL.  Asynchronous on hip_stream. */
int target_stream_fill_dev(const target_stream_c* spec, long n_targets, long first_tick, long n_ticks, int dtype, void* meas_dev,
                           long tick_stride, long ld, unsigned char* has_meas_dev, long has_stride, void* hip_stream);
/* pose0_dev [n][7]: the pose to create target i with (start position + measurement noise, identity orientation);
 * truth_dev [n][12]: p, v, a, omega of the generating motion.  Either may be NULL.  Device doubles. */
int target_stream_truth_dev(const target_stream_c* spec, long n_targets, double* pose0_dev, double* truth_dev, void* hip_stream);

/* ---- multi-GPU: gather of the estimated poses to one rank over xGMI (RCCL) ---------------------------- */
/* One process per GPU, every rank owns a shard of the targets (no collective in the predict/update path).  What the
 * reference's node does with the filtered poses every tick is publish them (src/target_manager_ros.cpp:78-87); across
 * GPUs that is a gather of pose7 rows to one rank.  It is a direct gather (one ncclSend per rank, one ncclRecv per peer
 * on the root: xGMI is point-to-point) on the communicator's own stream behind an event, so the step kernels of the
 * following ticks run while the poses travel.
 *   target_comm_unique_id : rank 0 creates the 128-byte id; the caller hands it to every rank (any channel)
 *   target_comm_new       : ncclCommInitRank on the calling thread's current HIP device (collective over all ranks)
 *   ..._gather_pose_begin : enqueue.  counts [world] = rows per rank (counts[rank] == target_manager_size);
 *                           recv_dev (root only) [sum(counts)][7] doubles, rank r's rows at row sum(counts[:r]), batch
 *                           order then slot order within a rank.  Returns immediately.
 *   ..._gather_pose_wait  : block the host until the last gather has finished; device_ms (may be NULL) = its duration
 * RCCL is looked up at run time (the copy already in the process, else librccl.so.1). */
typedef void target_comm_c;
int target_comm_unique_id(char* out128);
target_comm_c* target_comm_new(const char* id128, int rank, int world);
void target_comm_delete(target_comm_c* comm);
int target_manager_gather_pose_begin(target_manager_c* self, target_comm_c* comm, int root, const long* counts, double* recv_dev);
int target_manager_gather_pose_wait(target_comm_c* comm, float* device_ms);
/* the same with a deadline (the done event is polled with hipEventQuery): 0 finished, 1 still in flight after timeout_s
 * seconds (a peer that never posted its send / recv; nothing is cancelled), < 0 error */
int target_manager_gather_pose_wait_for(target_comm_c* comm, double timeout_s, float* device_ms);

/* ---- measurement ingest: the ROS node's mailbox / has-measurement / expiry policy --------------- */
/* Transport-agnostic restatement of class Measurement (target_manager_ros.hpp:74-134) and
 * RosTargetManager::update (src/target_manager_ros.cpp:41-92).  The caller pushes (id | frame name,
 * stamp, pose); one tick = create the targets seen for the first time, predict+update those whose
 * mailbox holds a new measurement, predict the others, erase those whose last measurement is older
 * than the expiration time; as one batched init + one batched step.
 * Q == NULL: new targets use the manager's default model (then type, R, P0 are ignored). */
typedef void target_ingest_c;
target_ingest_c* target_ingest_new(target_manager_c* manager, int type, const double* Q, const double* R, const double* P0);
void target_ingest_delete(target_ingest_c* ingest);
void target_ingest_set_expiration_time(target_ingest_c* ingest, double seconds);   /* setExpirationTime, :99-103 */
void target_ingest_set_token_name(target_ingest_c* ingest, const char* token);      /* setTargetTokenName, :94-97 */
int target_ingest_push(target_ingest_c* ingest, unsigned int id, double stamp, const double* pose);
/* 1 taken, 0 frame name without the token, -1 token present but not "<name>_<id>" (the reference then
 * drops the rest of the message, target_manager_ros.cpp:34-35) */
int target_ingest_push_named(target_ingest_c* ingest, const char* child_frame_id, double stamp, const double* pose);
/* one node tick at wall time `now`; ids_out / poses_out ([capacity], [capacity][7]) receive the live
 * targets in ascending id order with their filtered poses; returns their number */
long target_ingest_tick(target_ingest_c* ingest, double dt, double now, unsigned int* ids_out, double* poses_out, long capacity);

#ifdef __cplusplus
}
#endif
#endif
