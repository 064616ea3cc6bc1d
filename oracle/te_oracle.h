/*
 * te_oracle.h -- CPU restatement of the reference's per-target Kalman path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under target_estimation_amd/ (the product)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / reported CPU
 * baseline.
 *
 * PARITY STATUS: "parity unpinned" at step level.  The reference cannot be
 * compiled in this environment (Eigen3, yaml-cpp, gtest, ROS absent) and ships
 * no golden vectors for kalman.cpp / src/types; what pins this restatement is
 *   (1) the reference's own test assertions, restated in tests/test_oracle_harness.py
 *       (test/target_manager_test.cpp:179-189,223-233,268-281,321-340),
 *   (2) the identities of test/geometry_test.cpp (1e-4),
 *   (3) the hand-derived uniform-velocity step-1 known answer,
 *   (4) an independent NumPy restatement (oracle/np_twin.py).
 *
 * Every function cites the reference file:line it follows.  Arithmetic follows
 * the reference's association order: dense (A*P)*A^T + Q, (P*C^T)*inv(S),
 * (I - K*C)*P, inverse by partial-pivot LU (what Eigen's dynamic .inverse() does),
 * accumulation k-ascending, no FMA contraction (build with -ffp-contract=off).
 *
 * Two instantiations: _f64 (the reference's precision) and _f32.
 * Matrices are row-major here; Eigen's column-major storage does not change the
 * arithmetic.
 */
#ifndef TE_ORACLE_H
#define TE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NMAX 18
#define ORC_MMAX 6

/* include/target_estimation/target_manager.hpp:38 */
enum {
  ORC_ANGULAR_RATES = 0,
  ORC_ANGULAR_VELOCITIES = 1,
  ORC_UNIFORM_ACCELERATION = 2,
  ORC_UNIFORM_VELOCITY = 3
};

#define ORC_DECLARE(REAL, SFX)                                                                    \
  typedef struct orc_target_##SFX {                                                               \
    int model, n, m, initialized;                                                                 \
    unsigned id;                                                                                  \
    long long n_meas;                                                                             \
    double t;                                                                                     \
    /* estimator (include/target_estimation/kalman.hpp:95-150) */                                 \
    REAL A[ORC_NMAX * ORC_NMAX], C[ORC_MMAX * ORC_NMAX], Q[ORC_NMAX * ORC_NMAX];                  \
    REAL R[ORC_MMAX * ORC_MMAX], P0[ORC_NMAX * ORC_NMAX], P[ORC_NMAX * ORC_NMAX];                 \
    REAL K[ORC_NMAX * ORC_MMAX], Id[ORC_NMAX * ORC_NMAX];                                          \
    REAL x_hat[ORC_NMAX], x_hat_new[ORC_NMAX];                                                    \
    /* target (include/target_estimation/target_interface.hpp:193-282) */                         \
    REAL x[ORC_NMAX], P_tgt[ORC_NMAX * ORC_NMAX];                                                 \
    REAL T_lin[9], T_trans[3];                                                                    \
    REAL twist[6], acceleration[6], pose_internal[6], measured_pose[7], meas_rpy_internal[3];     \
    REAL f_dt; /* dt bound into f_ by std::bind (angular_velocities.cpp:98,110) */                \
  } orc_target_##SFX;                                                                             \
                                                                                                  \
  int orc_target_sizeof_##SFX(void);                                                              \
  int orc_target_init_##SFX(orc_target_##SFX* tg, int model, unsigned id, double dt0, double t0,  \
                            const double* Q, const double* R, const double* P0,                  \
                            const double* p0, const double* v0, const double* a0);               \
  void orc_target_add_measurement_##SFX(orc_target_##SFX* tg, double dt, const double* meas);     \
  void orc_target_update_##SFX(orc_target_##SFX* tg, double dt);                                  \
  void orc_target_get_state_##SFX(const orc_target_##SFX* tg, double* x, double* P);              \
  void orc_target_get_pose_##SFX(const orc_target_##SFX* tg, double* pose7);                      \
  void orc_target_get_twist_##SFX(const orc_target_##SFX* tg, double* twist6);                    \
  void orc_target_get_acceleration_##SFX(const orc_target_##SFX* tg, double* acc6);               \
  void orc_target_get_measured_pose_##SFX(const orc_target_##SFX* tg, double* pose7);             \
  void orc_target_get_pose6_##SFX(const orc_target_##SFX* tg, double* pose6);                     \
  void orc_target_get_transform_##SFX(const orc_target_##SFX* tg, double* T16);                   \
  double orc_target_get_period_estimate_##SFX(const orc_target_##SFX* tg);                        \
  void orc_target_get_pose_at_##SFX(const orc_target_##SFX* tg, double t1, double* pose7);        \
  void orc_target_get_twist_at_##SFX(const orc_target_##SFX* tg, double t1, double* twist6);      \
  void orc_target_get_acceleration_at_##SFX(const orc_target_##SFX* tg, double t1, double* a6);   \
  double orc_intersection_time_##SFX(const orc_target_##SFX* tg, double t1, const double* origin, \
                                     double radius);                                              \
  int orc_intersection_pose_##SFX(const orc_target_##SFX* tg, double t1, const double* origin,    \
                                  double radius, double* pose7, double* delta_t);                 \
  /* batch helpers: array of targets, OpenMP over targets (cpu_baseline) */                       \
  void orc_batch_step_##SFX(orc_target_##SFX* tgs, long n, double dt, const double* meas,         \
                            const unsigned char* has_meas, int nthreads);                         \
  void orc_batch_get_state_##SFX(const orc_target_##SFX* tgs, long n, double* x, double* P);      \
  /* geometry (include/target_estimation/geometry.hpp), exported for unit tests */                \
  REAL orc_constrain_angle_##SFX(REAL x);                                                         \
  REAL orc_angle_conv_##SFX(REAL a);                                                              \
  REAL orc_angle_diff_##SFX(REAL a, REAL b);                                                      \
  REAL orc_unwrap_##SFX(REAL prev, REAL now);                                                     \
  void orc_quat_normalize_##SFX(REAL* q);                                                         \
  void orc_quat_to_rpy_##SFX(const REAL* q, REAL* rpy);                                           \
  void orc_rpy_to_quat_##SFX(const REAL* rpy, REAL* q);                                           \
  void orc_rot_to_rpy_##SFX(const REAL* R, REAL* rpy);                                            \
  void orc_quat_to_rot_##SFX(const REAL* q, REAL* R);                                             \
  void orc_rot_to_quat_##SFX(const REAL* R, REAL* q);                                             \
  void orc_rpy_to_ear_base_##SFX(const REAL* rpy, REAL* E);                                       \
  void orc_rpy_to_ear_base_inv_##SFX(const REAL* rpy, REAL* E);                                   \
  void orc_ear_base_inv_jac_rpy_##SFX(const REAL* rpy, const REAL* omega, REAL dt, REAL* J);      \
  void orc_ear_base_inv_jac_omega_##SFX(const REAL* rpy, REAL dt, REAL* J);                       \
  void orc_qtran_##SFX(REAL dt, const REAL* omega, REAL* M);                                      \
  int orc_inverse_##SFX(int n, const REAL* A, REAL* Ainv);

ORC_DECLARE(double, f64)
ORC_DECLARE(float, f32)

/* MovingAvgFilter (include/target_estimation/utils.hpp:206-265) and the convergence gate of
 * IntersectionSolver::getIntersectionPoseWithSphere (src/intersection_solver.cpp:105-120): one gate per
 * solver object in the reference (= per target here). */
typedef struct orc_moving_avg {
  int n, idx, complete;
  double sum, variance;
  double window[1024];
} orc_moving_avg;
void orc_moving_avg_init(orc_moving_avg* f, int n);
double orc_moving_avg_update(orc_moving_avg* f, double value);
typedef struct orc_gate {
  orc_moving_avg pos, ang;
  double prev_pose[7];
} orc_gate;
void orc_gate_init(orc_gate* g, int filters_length);
/* feeds one query result (exists = delta > -1, pose at the intersection) through the gate; returns converged */
int orc_gate_update(orc_gate* g, int exists, const double* pose7, double pos_th, double ang_th,
                    double* pos_err_filt, double* ang_err_filt);
int orc_gate_sizeof(void);

/* src/intersection_solver.cpp:4-17; coefficients lowest order first */
double orc_lowest_real_root(const double* coeffs, int ncoeffs);
/* all complex roots (re,im interleaved) of a polynomial, for tests */
int orc_poly_roots(const double* coeffs, int ncoeffs, double* roots_re_im);

int orc_max_threads(void);

/* one target through the reference test's loop (test/target_manager_test.cpp:125-146), est_pose [n][7], est_twist [n][6] */
void orc_harness_run_f64(int model, const double* Q, const double* R, const double* P0, const double* meas, long n, double dt,
                         double* est_pose, double* est_twist);

/* CPU twin of the product's counter-based stream generator (te_stream.c; definition: csrc/stream_gen.hpp) */
void orc_stream_truth(int model, unsigned long long seed, long target, double* truth12, double* pose0);
int orc_stream_measurement(int model, unsigned long long seed, long target, long tick, double dt, double availability,
                           double rpy_noise, double* meas7);
void orc_stream_fill(int model, unsigned long long seed, long first_target, long n_targets, long first_tick, long n_ticks,
                     double dt, double availability, double rpy_noise, int f32, double* meas, unsigned char* has,
                     double* pose0, double* truth);

#ifdef __cplusplus
}
#endif
#endif
