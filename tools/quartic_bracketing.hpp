// quartic_bracketing.hpp -- the earlier device solver (every monotone piece bracketed in turn), kept for A/B runs of tools/quartic_bench.hip
//
// Reference semantics (src/intersection_solver.cpp:4-17, Eigen::PolynomialSolver from
// unsupported/Eigen/Polynomials, not under the reference tree): if the leading coefficient is
// zero return -1; otherwise take all roots, keep those with |imag| < 1e-10, return the one with
// the smallest real part, or -1 if none.  (The caller maps a negative result to -1 as well,
// src/intersection_solver.cpp:83.)
//
// Eigen finds the roots as eigenvalues of the companion matrix.  Only the real roots matter, so
// here the line is split at the real critical points (roots of the derivative, themselves found the
// same way from the roots of the second derivative, a quadratic); on each of the resulting monotone
// intervals a sign change brackets exactly one simple real root, which a safeguarded Newton
// iteration converges to full double precision.  The first root from the left is the answer.
// Cost: a few hundred flops, independent of how badly scaled the coefficients are (a tiny
// leading coefficient, i.e. a nearly unaccelerated target, sends two roots towards infinity; an
// iteration on all four complex roots then needs hundreds of steps).
// A multiple root (trajectory tangent to the sphere: p = p' = 0) has no sign change and is
// reported as "no real root"; an eigen-solver returns such a pair with imaginary parts of order
// sqrt(eps), also beyond the reference's 1e-10 threshold.  Solver-specific either way: unpinned.
#pragma once
#include <hip/hip_runtime.h>

namespace te_bracketing {

// value and derivative of c[0] + c[1] x + ... + c[deg] x^deg (Horner)
__device__ __forceinline__ void poly_eval(const double* c, int deg, double x, double* f, double* df) {
  double v = c[deg], d = 0.0;
  for (int k = deg - 1; k >= 0; --k) {
    d = d * x + v;
    v = v * x + c[k];
  }
  *f = v;
  *df = d;
}

// 1/d to about 2^-28 relative: v_rcp_f64 and one Newton step (a full IEEE division is ~15 instructions)
__device__ __forceinline__ double rcp_approx(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  return fma(fma(-d, r, 1.0), r, r);
}

// the root in (lo, hi) of a polynomial that is monotone there, given f(lo) = flo with the opposite sign of
// f(hi): Newton steps, replaced by bisection whenever they would leave the bracket or converge slowly
__device__ inline double poly_root_in(const double* c, int deg, double lo, double hi, double flo) {
  // Brackets can span many orders of magnitude (the Cauchy bound of a nearly degenerate quartic is
  // huge): first shrink them geometrically, so that the Newton phase starts within a factor 4.
  double f, df;
  if (lo < 0.0 && hi > 0.0) {                 // split at zero: f(0) = c[0]
    if (c[0] == 0.0) return 0.0;
    if ((c[0] < 0.0) == (flo < 0.0)) { lo = 0.0; flo = c[0]; } else hi = 0.0;
  }
  for (int it = 0; it < 16; ++it) {                  // halves the binary-exponent range: <= 12 rounds in fp64
    const double al = fabs(lo), ah = fabs(hi);
    const double mn = fmin(al, ah), mx = fmax(al, ah);
    if (mx <= 4.0 * mn || mx < 1e-300) break;
    // a point strictly between mn and mx in magnitude: 2^ceil((e_mn + e_mx)/2) (integer exponent arithmetic,
    // no square roots); with mn = 0 step down by 2^-10
    double x = mn > 0.0 ? ldexp(1.0, (ilogb(mn) + ilogb(mx) + 1) >> 1) : ldexp(mx, -10);
    if ((lo + hi) < 0.0) x = -x;
    poly_eval(c, deg, x, &f, &df);
    if (f == 0.0) return x;
    if ((f < 0.0) == (flo < 0.0)) { lo = x; flo = f; } else hi = x;
  }
  double xl = flo < 0.0 ? lo : hi, xh = flo < 0.0 ? hi : lo;   // f(xl) < 0 < f(xh)
  double x = 0.5 * (lo + hi);
  double dxold = fabs(hi - lo), dx = dxold;
  poly_eval(c, deg, x, &f, &df);
  for (int it = 0; it < 200; ++it) {
    if (f == 0.0) break;
    if (f < 0.0) xl = x; else xh = x;
    const bool outside = ((x - xh) * df - f) * ((x - xl) * df - f) > 0.0;
    if (outside || fabs(2.0 * f) > fabs(dxold * df)) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      x = xl + dx;
    } else {
      dxold = dx;
      dx = f * rcp_approx(df);   // a 2^-28 reciprocal costs the iteration nothing: its error scales with dx
      x -= dx;
    }
    if (fabs(dx) <= 2.0 * 2.220446049250313e-16 * fabs(x) || dx == 0.0) break;
    poly_eval(c, deg, x, &f, &df);
  }
  return x;
}

// ascending simple real roots of a polynomial of degree `deg` inside (-B, B), given the ascending real roots
// `crit` of its derivative (which split the line into monotone pieces); returns their number
__device__ inline int roots_between(const double* c, int deg, const double* crit, int ncrit, double B, double* out,
                                    int max_roots) {
  int n = 0;
  double lo = -B, flo, d;
  poly_eval(c, deg, lo, &flo, &d);
  for (int k = 0; k <= ncrit; ++k) {
    const double hi = (k < ncrit) ? crit[k] : B;
    if (!(hi > lo)) continue;
    double fhi;
    poly_eval(c, deg, hi, &fhi, &d);
    if ((flo < 0.0 && fhi > 0.0) || (flo > 0.0 && fhi < 0.0)) {
      out[n++] = poly_root_in(c, deg, lo, hi, flo);
      if (n >= max_roots) return n;
    }
    lo = hi;
    if (fhi != 0.0) flo = fhi;   // exactly zero at a critical point = multiple root: not a simple real root
  }
  return n;
}

// coefficients lowest order first: c[0] + c[1] x + ... + c[4] x^4
__device__ inline double lowest_real_root_quartic(const double* c) {
  if (!(fabs(c[4]) > 0.0)) return -1.0;
  // every root (and, by Gauss-Lucas, every root of the derivatives) lies in (-B, B)
  double B = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) B = fmax(B, fabs(c[i]));
  B = B * rcp_approx(fabs(c[4])) * 1.000001 + 1.0;   // Cauchy bound (slightly inflated: approximate reciprocal)
  const double d1[4] = {c[1], 2.0 * c[2], 3.0 * c[3], 4.0 * c[4]};        // p'
  const double d2[3] = {d1[1], 2.0 * d1[2], 3.0 * d1[3]};                  // p''
  // roots of the quadratic p'' (stable form), ascending
  double r2[2];
  int n2 = 0;
  {
    const double qa = d2[2], qb = d2[1], qc = d2[0];
    const double disc = qb * qb - 4.0 * qa * qc;
    if (disc > 0.0) {
      const double qq = -0.5 * (qb + (qb >= 0.0 ? sqrt(disc) : -sqrt(disc)));
      double x0 = qq / qa, x1 = (qq != 0.0) ? qc / qq : x0;
      if (x0 > x1) { const double t = x0; x0 = x1; x1 = t; }
      r2[0] = x0; r2[1] = x1; n2 = 2;
    }
  }
  double r1[3], r0[4];
  const int n1 = roots_between(d1, 3, r2, n2, B, r1, 3);   // critical points of p
  const int n0 = roots_between(c, 4, r1, n1, B, r0, 1);    // the leftmost real root of p is enough
  return n0 > 0 ? r0[0] : -1.0;
}

}  // namespace te_bracketing
