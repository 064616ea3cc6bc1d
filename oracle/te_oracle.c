/*
 * te_oracle.c -- CPU oracle for the per-target Kalman path (TEST INFRASTRUCTURE ONLY).
 * See te_oracle.h for the parity status ("parity unpinned" at step level) and for who may
 * use this.  Build: oracle/Makefile.
 */
#include "te_oracle.h"

#include <complex.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Polynomial roots.  The reference uses Eigen::PolynomialSolver (unsupported/Eigen/Polynomials,
 * not under /root/reference, version unpinned): eigenvalues of the companion matrix, then
 * smallestRealRoot(hasRealRoot, 1e-10) = the root with the smallest REAL PART among those with
 * |imag| < threshold.  Here the roots come from an Aberth-Ehrlich iteration in long double
 * (simple roots agree with any backward-stable eigen-solver to a few ulp; the classification of
 * near-multiple roots against the 1e-10 threshold is inherently solver-specific: unpinned). */
/* ------------------------------------------------------------------------- */
int orc_poly_roots(const double* coeffs, int ncoeffs, double* roots_re_im) {
  int deg = ncoeffs - 1;
  while (deg > 0 && coeffs[deg] == 0.0) --deg;
  if (deg <= 0) return 0;
  long double complex z[8];
  long double a[9];
  for (int i = 0; i <= deg; ++i) a[i] = (long double)coeffs[i] / (long double)coeffs[deg];
  long double rad = 0;
  for (int i = 0; i < deg; ++i) {
    long double v = fabsl(a[i]);
    if (v > rad) rad = v;
  }
  rad = 1.0L + rad;
  /* start on a circle of half the Cauchy bound, irrational phase */
  for (int i = 0; i < deg; ++i) {
    long double ang = 2.0L * 3.14159265358979323846264338327950288L * i / deg + 0.4L;
    z[i] = 0.5L * rad * (cosl(ang) + I * sinl(ang));
  }
  for (int it = 0; it < 500; ++it) {
    long double maxstep = 0;
    for (int i = 0; i < deg; ++i) {
      long double complex p = 1, dp = 0;
      for (int k = deg - 1; k >= 0; --k) {
        dp = dp * z[i] + p;
        p = p * z[i] + a[k];
      }
      if (cabsl(p) == 0) continue;
      long double complex ratio = p / dp;
      long double complex sum = 0;
      for (int j = 0; j < deg; ++j)
        if (j != i) sum += 1.0L / (z[i] - z[j]);
      long double complex step = ratio / (1.0L - ratio * sum);
      z[i] -= step;
      long double s = cabsl(step) / (1.0L + cabsl(z[i]));
      if (s > maxstep) maxstep = s;
    }
    if (maxstep < 1e-19L) break;
  }
  for (int i = 0; i < deg; ++i) {
    roots_re_im[2 * i] = (double)creall(z[i]);
    roots_re_im[2 * i + 1] = (double)cimagl(z[i]);
  }
  return deg;
}

/* Solver::lowestRealRoot, src/intersection_solver.cpp:4-17 */
double orc_lowest_real_root(const double* coeffs, int ncoeffs) {
  double roots[16];
  if (!(fabs(coeffs[ncoeffs - 1]) > 0.0)) return -1;
  int deg = orc_poly_roots(coeffs, ncoeffs, roots);
  const double imThreshold = 1e-10;
  int found = 0;
  double best = 0;
  for (int i = 0; i < deg; ++i) {
    if (fabs(roots[2 * i + 1]) < imThreshold) {
      if (!found || roots[2 * i] < best) { best = roots[2 * i]; found = 1; }
    }
  }
  if (!found) return -1;
  return best;
}

/* ------------------------------------------------------------------------- */
/* MovingAvgFilter, include/target_estimation/utils.hpp:206-265                  */
/* ------------------------------------------------------------------------- */
void orc_moving_avg_init(orc_moving_avg* f, int n) {   /* ctor :212-220 */
  f->n = n; f->idx = 0; f->complete = 0; f->sum = 0.0; f->variance = 0.0;
  for (int i = 0; i < n; ++i) f->window[i] = 0.0;
}

double orc_moving_avg_update(orc_moving_avg* f, double value) {   /* update :222-251 */
  const int n = f->n;
  f->sum -= f->window[f->idx];
  f->sum += value;
  f->window[f->idx] = value;
  if (!f->complete && f->idx == n - 1) f->complete = 1;
  int num = n;
  if (!f->complete) num = f->idx + 1;
  const double res = f->sum / num;
  f->idx = (f->idx + 1) % n;
  double variance_sum = 0.0;
  for (int i = 0; i < n; ++i) variance_sum += pow(f->window[i] - res, 2);
  f->variance = variance_sum / num;
  return res;
}

/* wrapMax / wrapMinMax, geometry.hpp:79-88 */
static double wrap_max(double x, double max) { return fmod(max + fmod(x, max), max); }
static double wrap_min_max(double x, double min, double max) { return min + wrap_max(x - min, max - min); }

void orc_gate_init(orc_gate* g, int filters_length) {   /* IntersectionSolver ctor, intersection_solver.cpp:19-40 */
  orc_moving_avg_init(&g->pos, filters_length);
  orc_moving_avg_init(&g->ang, filters_length);
  for (int i = 0; i < 7; ++i) g->prev_pose[i] = (i == 6) ? 1.0 : 0.0;   /* initPose */
}

int orc_gate_sizeof(void) { return (int)sizeof(orc_gate); }

/* src/intersection_solver.cpp:102-123 */
int orc_gate_update(orc_gate* g, int exists, const double* pose7, double pos_th, double ang_th,
                    double* pos_err_filt, double* ang_err_filt) {
  if (!exists) return 0;
  double dx = pose7[0] - g->prev_pose[0], dy = pose7[1] - g->prev_pose[1], dz = pose7[2] - g->prev_pose[2];
  const double pos_error = sqrt(dx * dx + dy * dy + dz * dz);
  double q1[4] = {pose7[3], pose7[4], pose7[5], pose7[6]};
  double q2[4] = {g->prev_pose[3], g->prev_pose[4], g->prev_pose[5], g->prev_pose[6]};
  orc_quat_normalize_f64(q1);
  orc_quat_normalize_f64(q2);
  /* computeQuaternionError, geometry.hpp:630-651: q_e = q_des * q.inverse(), normalised
   * (Eigen: inverse = conjugate / squaredNorm; product = Hamilton product), [x y z w] */
  const double n2 = q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2] + q2[3] * q2[3];
  const double ix = -q2[0] / n2, iy = -q2[1] / n2, iz = -q2[2] / n2, iw = q2[3] / n2;
  double qe[4];
  qe[3] = q1[3] * iw - q1[0] * ix - q1[1] * iy - q1[2] * iz;
  qe[0] = q1[3] * ix + q1[0] * iw + q1[1] * iz - q1[2] * iy;
  qe[1] = q1[3] * iy + q1[1] * iw + q1[2] * ix - q1[0] * iz;
  qe[2] = q1[3] * iz + q1[2] * iw + q1[0] * iy - q1[1] * ix;
  orc_quat_normalize_f64(qe);
  /* computeQuaternionErrorAngle :653-657, then abs(wrapMinMax(., -pi, pi)) :110 */
  const double ang_error = fabs(wrap_min_max(2 * acos(qe[3]), -M_PI, M_PI));
  const double pf = orc_moving_avg_update(&g->pos, pos_error);
  const double af = orc_moving_avg_update(&g->ang, ang_error);
  for (int i = 0; i < 7; ++i) g->prev_pose[i] = pose7[i];
  if (pos_err_filt) *pos_err_filt = pf;
  if (ang_err_filt) *ang_err_filt = af;
  return (pf <= pos_th && af <= ang_th) ? 1 : 0;
}

/* ------------------------------------------------------------------------- */
/* precision instantiations                                                   */
/* ------------------------------------------------------------------------- */
#define REAL double
#define SFX f64
#define RSIN sin
#define RCOS cos
#define RATAN2 atan2
#define RASIN asin
#define RSQRT sqrt
#define RFMOD fmod
#define RFABS fabs
#include "te_oracle_impl.h"
#undef REAL
#undef SFX
#undef RSIN
#undef RCOS
#undef RATAN2
#undef RASIN
#undef RSQRT
#undef RFMOD
#undef RFABS

#define REAL float
#define SFX f32
#define RSIN sinf
#define RCOS cosf
#define RATAN2 atan2f
#define RASIN asinf
#define RSQRT sqrtf
#define RFMOD fmodf
#define RFABS fabsf
#include "te_oracle_impl.h"

/* ------------------------------------------------------------------------- */
/* The reference's integration-test loop for ONE target, in C (for timing configs[0] without a Python loop around it):
 * _manager.init(type, id, dt, 0.0, Q, R, P, meas.row(0)) then n x { update(id, dt, meas_i); getTargetPose; getTargetTwist }
 * (generateEstimation, test/target_manager_test.cpp:125-146). */
/* ------------------------------------------------------------------------- */
void orc_harness_run_f64(int model, const double* Q, const double* R, const double* P0, const double* meas, long n, double dt,
                         double* est_pose, double* est_twist) {
  orc_target_f64 tg;
  orc_target_init_f64(&tg, model, 0u, dt, 0.0, Q, R, P0, meas, 0, 0);
  for (long i = 0; i < n; ++i) {
    orc_target_add_measurement_f64(&tg, dt, meas + 7 * i);
    orc_target_get_pose_f64(&tg, est_pose + 7 * i);
    orc_target_get_twist_f64(&tg, est_twist + 6 * i);
  }
}
