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
