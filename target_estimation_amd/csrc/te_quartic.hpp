// te_quartic.hpp -- smallest real root of the sphere-intersection quartic, on the device.
//
// Reference semantics (src/intersection_solver.cpp:4-17, Eigen::PolynomialSolver from
// unsupported/Eigen/Polynomials, not under the reference tree): if the leading coefficient is
// zero return -1; otherwise take all roots, keep those with |imag| < 1e-10, return the one with
// the smallest real part, or -1 if none.  (The caller maps a negative result to -1 as well,
// src/intersection_solver.cpp:83.)
//
// Eigen finds the roots as eigenvalues of the companion matrix; here they come from an
// Aberth-Ehrlich iteration in complex double followed by a real Newton polish, so simple real
// roots get an exactly zero imaginary part (as a real Schur form gives them) and conjugate pairs
// keep theirs.  Near-multiple roots (tangent trajectories) are classified against the 1e-10
// threshold by whatever error the solver leaves -- solver-specific in the reference too, unpinned.
#pragma once
#include <hip/hip_runtime.h>

namespace te {

struct Cplx { double re, im; };
__device__ __forceinline__ Cplx cadd(Cplx a, Cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cplx csub(Cplx a, Cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cplx cdiv(Cplx a, Cplx b) {
  // Smith's algorithm
  if (fabs(b.re) >= fabs(b.im)) {
    const double r = b.im / b.re, d = b.re + b.im * r;
    return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
  }
  const double r = b.re / b.im, d = b.re * r + b.im;
  return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
__device__ __forceinline__ double cabs1(Cplx a) { return hypot(a.re, a.im); }

// coefficients lowest order first: c[0] + c[1] x + ... + c[4] x^4
__device__ inline double lowest_real_root_quartic(const double* c) {
  if (!(fabs(c[4]) > 0.0)) return -1.0;
  double a[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) a[i] = c[i] / c[4];
  double rad = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) rad = fmax(rad, fabs(a[i]));
  rad = 1.0 + rad;
  Cplx z[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double s, co;
    sincos(2.0 * 3.14159265358979323846 * i / 4.0 + 0.4, &s, &co);
    z[i] = {0.5 * rad * co, 0.5 * rad * s};
  }
  for (int it = 0; it < 200; ++it) {
    double maxstep = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Cplx p = {1.0, 0.0}, dp = {0.0, 0.0};
#pragma unroll
      for (int k = 3; k >= 0; --k) {
        dp = cadd(cmul(dp, z[i]), p);
        p = cadd(cmul(p, z[i]), Cplx{a[k], 0.0});
      }
      if (p.re == 0.0 && p.im == 0.0) continue;
      const Cplx ratio = cdiv(p, dp);
      Cplx sum = {0.0, 0.0};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j != i) sum = cadd(sum, cdiv(Cplx{1.0, 0.0}, csub(z[i], z[j])));
      const Cplx step = cdiv(ratio, csub(Cplx{1.0, 0.0}, cmul(ratio, sum)));
      z[i] = csub(z[i], step);
      maxstep = fmax(maxstep, cabs1(step) / (1.0 + cabs1(z[i])));
    }
    if (maxstep < 1e-16) break;
  }
  bool found = false;
  double best = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double re = z[i].re, im = z[i].im;
    if (fabs(im) <= 1e-7 * (1.0 + fabs(re))) {
      // real Newton polish: a simple real root converges and is then exactly real
      double x = re;
      bool ok = false;
      for (int k = 0; k < 8; ++k) {
        const double p = (((x + a[3]) * x + a[2]) * x + a[1]) * x + a[0];
        const double dp = ((4.0 * x + 3.0 * a[3]) * x + 2.0 * a[2]) * x + a[1];
        const double mag = (((fabs(x) + fabs(a[3])) * fabs(x) + fabs(a[2])) * fabs(x) + fabs(a[1])) * fabs(x) + fabs(a[0]);
        if (fabs(p) <= 64.0 * 2.220446049250313e-16 * mag) { ok = true; break; }
        if (dp == 0.0) break;
        x -= p / dp;
      }
      if (ok && fabs(x - re) <= 1e-6 * (1.0 + fabs(re))) { re = x; im = 0.0; }
    }
    if (fabs(im) < 1e-10) {
      if (!found || re < best) { best = re; found = true; }
    }
  }
  return found ? best : -1.0;
}

}  // namespace te
