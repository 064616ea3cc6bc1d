// te_quartic.hpp -- first crossing time of the sphere-intersection quartic, on the device.
//
// Reference semantics (src/intersection_solver.cpp:4-17, Eigen::PolynomialSolver from
// unsupported/Eigen/Polynomials, not under the reference tree): if the leading coefficient is
// zero return -1; otherwise take all roots, keep those with |imag| < 1e-10, return the one with
// the smallest real part, or -1 if none.  The only caller maps a negative result to -1 as well
// (src/intersection_solver.cpp:83), so what the query needs is
//     first_crossing_quartic(c) = the leftmost real root if it is >= 0, else -1.
//
// Eigen finds the roots as eigenvalues of the companion matrix.  Only the leftmost real root matters.
// Round 4: most quartics never get as far as a root finder -- Sturm's chain counts the real roots on either side of
// zero (quartic_sturm_classify below), which settles "-1" for a trajectory that misses the sphere or has left it behind,
// and tells when the answer is simply the smallest positive root (first_positive_root).  What neither settles
// (cancellation, multiple roots, exact zeros) is located rigorously from the shape of the curve, as since round 2:
//   1. the critical points of p (real roots of the cubic p') split the line into monotone pieces.
//      They come from W. Kahan's cubic algorithm ("To Solve a Real Cubic Equation", 1986): ONE
//      Newton iteration that starts outside the outermost root and converges monotonically, then
//      deflation (forward or backward, whichever is stable) to a quadratic solved in closed form;
//   2. with the leading coefficient made positive the leftmost root sits on the first decreasing
//      piece that reaches below zero: (-inf, m0) or (m1, m2).  A piece that ends at or below x = 0,
//      or on which p(0) < 0, holds a negative root: the answer is -1 without computing it;
//   3. otherwise the inflection points of p (closed form) cut the piece into parts of constant
//      convexity; on the part with the sign change Newton's iteration started from the correct end
//      converges monotonically, without bracketing bookkeeping.
// Cost: one cubic and one quartic Newton iteration of a handful of steps each plus three quadratic
// formulas -- a few hundred fp64 instructions, against several thousand for bracketing every
// monotone piece in turn (tools/quartic_bracketing.hpp, the earlier version); uniform across a wavefront.
// A multiple root (trajectory tangent to the sphere: p = p' = 0) has no sign change and is reported
// as "no real root"; an eigen-solver returns such a pair with imaginary parts of order sqrt(eps),
// also beyond the reference's 1e-10 threshold.  Solver-specific either way: unpinned.
//
// The file also compiles on the host (TE_QUARTIC_HOST) for the CPU-side solver tests.
#pragma once
#ifdef TE_QUARTIC_HOST
#include <cmath>
#define TE_QDEV inline
namespace te { namespace qdetail {
using std::fabs; using std::fma; using std::fmax; using std::sqrt; using std::frexp; using std::ldexp;
inline double rcp(double d) { return 1.0 / d; }
inline float cbrt_f(float x) { return std::cbrt(x); }
}}
#else
#include <hip/hip_runtime.h>
#define TE_QDEV __device__ __forceinline__
namespace te { namespace qdetail {
// 1/d to about 2^-50: v_rcp_f64 and two Newton steps, no scaling / fix-up (operands here are normal)
__device__ __forceinline__ double rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  return fma(fma(-d, r, 1.0), r, r);
}
__device__ __forceinline__ float cbrt_f(float x) { return ::cbrtf(x); }
}}
#endif

#ifndef TE_QTS
#define TE_QTS(i)   // phase marks for tools/quartic_bench.hip
#endif

namespace te {
#ifdef TE_QUARTIC_HOST
using namespace qdetail;
#else
#pragma clang fp contract(off)   // one rounding sequence wherever the solver is inlined (te_device_math.hpp); the explicit fma calls stay
#endif

// value and derivative of the quartic c[0] + c[1] x + ... + c[4] x^4 (Horner)
TE_QDEV void quartic_eval(const double* c, double x, double* f, double* df) {
  double v = c[4], d = 0.0;
#pragma unroll
  for (int k = 3; k >= 0; --k) {
    d = fma(d, x, v);
    v = fma(v, x, c[k]);
  }
  *f = v;
  *df = d;
}

// real roots of a x^2 + b x + c (a != 0), ascending; returns 0 or 2 (a double root counts as none: the
// callers only want points where the sign changes)
TE_QDEV int quadratic_roots(double a, double b, double c, double* r) {
  const double disc = fma(b, b, -4.0 * a * c);
  if (!(disc > 0.0)) return 0;
  const double sq = sqrt(disc);
  const double qq = -0.5 * (b + (b >= 0.0 ? sq : -sq));
  double x0 = qq * qdetail::rcp(a), x1 = (qq != 0.0) ? c * qdetail::rcp(qq) : x0;
  if (x0 > x1) { const double t = x0; x0 = x1; x1 = t; }
  r[0] = x0; r[1] = x1;
  return 2;
}

// upper bound (within ~1e-6 above) of |t|^(1/3): float cube root of the mantissa, exponent handled exactly
TE_QDEV double cbrt_upper(double t) {
  t = fabs(t);
  if (!(t > 0.0)) return 0.0;
  int e;
  const double m = frexp(t, &e);             // t = m 2^e, m in [0.5, 1)
  int k = e / 3, j = e - 3 * k;              // e = 3k + j
  if (j < 0) { j += 3; k -= 1; }
  const float cr = qdetail::cbrt_f((float)ldexp(m, j));
  return ldexp((double)cr * 1.000001, k);
}

// Real roots of A x^3 + B x^2 + C x + D (A != 0), ascending, after W. Kahan's QBC; returns 1 or 3
// (a double root of the quotient quadratic is dropped: the sign of the cubic does not change there).
TE_QDEV int cubic_real_roots(double A, double B, double C, double D, double* out) {
  double X, b1, c2;
  if (D == 0.0) {
    X = 0.0; b1 = B; c2 = C;
  } else {
    const double rA = qdetail::rcp(A);
    X = -(B * rA) * (1.0 / 3.0);                           // the inflection point of the cubic
    double q0 = A * X;
    b1 = q0 + B; c2 = fma(b1, X, C);
    double dq = fma(q0 + b1, X, c2), q = fma(c2, X, D);
    double t = q * rA;
    const double s = t > 0.0 ? 1.0 : (t < 0.0 ? -1.0 : 0.0);
    TE_QTS(5);
    double r = cbrt_upper(t);
    t = -dq * rA;
    if (t > 0.0) r = 1.324717957244746 * 1.000001 * fmax(r, sqrt(t));
    double x0 = X - s * r;                                 // outside the outermost root on the side of s
    TE_QTS(6);
    if (x0 != X) {
      for (int it = 0; it < 80; ++it) {                    // monotone from outside: stops when it no longer advances
        X = x0;
        q0 = A * X;
        b1 = q0 + B; c2 = fma(b1, X, C);
        dq = fma(q0 + b1, X, c2); q = fma(c2, X, D);
        x0 = (dq == 0.0) ? X : X - (q * qdetail::rcp(dq)) * (1.0 - 1.0e-15);
        if (!(s * x0 > s * X)) break;
      }
      // deflate backwards when the root found is the large one (forward deflation would cancel)
      if (fabs(A) * X * X > fabs(D * qdetail::rcp(X))) {
        const double rX = qdetail::rcp(X);
        c2 = -D * rX; b1 = (c2 - C) * rX;
      }
    }
  }
  TE_QTS(7);
  double r2[2];
  const int n2 = quadratic_roots(A, b1, c2, r2);
  TE_QTS(8);
  if (n2 == 0) { out[0] = X; return 1; }
  // merge X into the ascending pair
  if (X <= r2[0]) { out[0] = X; out[1] = r2[0]; out[2] = r2[1]; }
  else if (X <= r2[1]) { out[0] = r2[0]; out[1] = X; out[2] = r2[1]; }
  else { out[0] = r2[0]; out[1] = r2[1]; out[2] = X; }
  return 3;
}

// the root of the quartic in [lo, hi], 0 <= lo < hi, p(lo) > 0 > p(hi), p decreasing and of constant
// convexity there: Newton from the end it converges monotonically from (convex: lo, concave: hi),
// the bracket kept only as a guard against rounding
TE_QDEV double quartic_root_monotone(const double* c, double lo, double hi, bool convex) {
  double x = convex ? lo : hi;
  double dx = hi - lo;
  bool prev_pos = convex;                                  // sign of p at the starting end
  for (int it = 0; it < 100; ++it) {
    double f, df;
    quartic_eval(c, x, &f, &df);
    if (f == 0.0) return x;
    const bool pos = f > 0.0;
    // the approach is one-sided; a sign flip after a tiny step is the rounding noise of p itself: converged
    if (pos != prev_pos && dx <= 1e-9 * fabs(x)) return x;
    prev_pos = pos;
    if (pos) lo = x; else hi = x;
    double xn = (df != 0.0) ? x - f * qdetail::rcp(df) : 0.5 * (lo + hi);
    if (!(xn >= lo && xn <= hi)) xn = 0.5 * (lo + hi);     // left the bracket (rounding, or df ~ 0): bisect
    dx = fabs(xn - x);
    x = xn;
    if (dx <= 4.0 * 2.220446049250313e-16 * fabs(x)) break;
  }
  return x;
}

// How many real roots lie below zero and how many above it?  Sturm's chain of p (leading coefficient > 0) counts them without
// locating any: p0 = p, p1 = p', p(k+1) = -rem(p(k-1), pk) -- here each multiplied by a positive factor (16 c4, b2^2, e1^2) so
// that no division is needed:
//     p2 = b2 x^2 + b1 x + b0,   b2 = 3 c3^2 - 8 c2 c4,  b1 = 2 c2 c3 - 12 c1 c4,  b0 = c1 c3 - 16 c0 c4
//     p3 = e1 x + e0,            g = 3 c3 b2 - 4 c4 b1,  e1 = 4 c4 b2 b0 + g b1 - 2 c2 b2^2,  e0 = g b0 - c1 b2^2
//     p4 = -(b2 e0^2 - b1 e0 e1 + b0 e1^2)          (evaluated in a factored form, below)
// and (distinct real roots in (a, b]) = V(a) - V(b), V = sign changes along the chain.  A root below zero, or none above it: the
// answer is -1, which is what most targets get most of the time (a trajectory that misses the sphere, or has left it behind) --
// some 70 instructions without a branch instead of the critical points' iteration, and a wavefront whose targets are all
// settled this way skips the rest.  No root below zero and at least one above: the crossing exists and is the smallest positive
// root, which first_positive_root() finds without the critical points.
// Every value v is computed next to a magnitude M >= |v| (the same expression over absolute values); its rounding error is at
// most rho M with rho a small multiple of the unit roundoff that follows from the expression alone (first order; the test uses
// four times that).  A sign that is not certain -- cancellation, a shortened chain, a multiple root, under- or overflow --
// settles nothing and the target takes the long road as before (tools/quartic_bench.hip compares the two roads on recorded
// quartics, tests/host/quartic_host_test.cpp holds the classification's claims to the oracle's roots).
//   returns 0: not settled   1: the answer is -1   2: no root below zero, at least one above
TE_QDEV int quartic_sturm_classify(const double* c) {
  const double u = 2.220446049250313e-16;
  const double tiny = 1e-280;                               // (below this the relative bounds no longer hold)
  const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4];
  // (Descartes: with p(0) > 0 and no negative coefficient there is no root at or above zero, whatever lies below)
  const bool descartes = (c0 > 0.0) && (c1 >= 0.0) && (c2 >= 0.0) && (c3 >= 0.0);
  const bool sc0 = c0 > 0.0, sc1 = c1 > 0.0;
  bool sure = (c0 != 0.0) && (c1 != 0.0);
  const double d3 = 4.0 * c4, d2 = 3.0 * c3, d1 = 2.0 * c2;
  // p2: rho 4 u
  const double b2b = (8.0 * c4) * c2, b1b = (12.0 * c4) * c1, b0b = (16.0 * c4) * c0;
  const double b2 = fma(d2, c3, -b2b), Mb2 = fma(fabs(d2), fabs(c3), fabs(b2b));
  const double b1 = fma(d1, c3, -b1b), Mb1 = fma(fabs(d1), fabs(c3), fabs(b1b));
  const double b0 = fma(c1, c3, -b0b), Mb0 = fma(fabs(c1), fabs(c3), fabs(b0b));
  // p4 = -b2 (e0^2 - e1 h),  h = 4 c4 (b2 (3 c1^2 - 8 c0 c2) - b0^2): the same polynomial with b1 e0 - b0 e1 = b2 h taken out by
  // hand.  Term by term the leading orders of b1 e0 e1 and b0 e1^2 cancel when the acceleration is small (c4 ~ a^2, c3 ~ a): the
  // literal form loses |a| / |v| of its digits there and stops being certain below |a| ~ 1e-8; this one does not.
  // Only the sign of p4 is used: -sign(b2) sign(D), D = e0^2 - e1 h.   rho: h 13 u, D 30 u
  const double k = fma(3.0 * c1, c1, -(8.0 * c0) * c2), Mk = fma(3.0 * c1, c1, fabs((8.0 * c0) * c2));
  const double h = d3 * fma(b2, k, -(b0 * b0)), Mh = d3 * fma(Mb2, Mk, Mb0 * Mb0);
  const bool sb2 = b2 > 0.0, sb0 = b0 > 0.0;
  sure = sure && (fabs(b2) > fma(16.0 * u, Mb2, tiny)) && (fabs(b0) > fma(16.0 * u, Mb0, tiny));
  // p3: g rho 7 u, b2^2 rho 9 u, e1 rho 14 u, e0 rho 13 u
  const double g = fma(d2, b2, -(d3 * b1)), Mg = fma(fabs(d2), Mb2, d3 * Mb1);
  const double bb = b2 * b2, Mbb = Mb2 * Mb2;
  const double e1 = fma(d3 * b2, b0, fma(g, b1, -(d1 * bb))), Me1 = fma(d3 * Mb2, Mb0, fma(Mg, Mb1, fabs(d1) * Mbb));
  const double e0 = fma(g, b0, -(c1 * bb)), Me0 = fma(Mg, Mb0, fabs(c1) * Mbb);
  const bool se1 = e1 > 0.0, se0 = e0 > 0.0;
  sure = sure && (fabs(e1) > fma(56.0 * u, Me1, tiny)) && (fabs(e0) > fma(52.0 * u, Me0, tiny));
  const double D = fma(e0, e0, -(e1 * h)), MD = fma(Me0, Me0, Me1 * Mh);
  sure = sure && (fabs(D) > fma(120.0 * u, MD, tiny));
  if (descartes) return 1;
  if (!sure) return 0;                                      // (NaN and infinities compare false: not sure)
  const bool sp4 = (D > 0.0) != sb2;
  // sign sequences: at -inf (+, -, sb2, -se1, sp4); at 0 (sc0, sc1, sb0, se0, sp4); at +inf (+, +, sb2, se1, sp4)
  const int v_minus = 1 + (int)sb2 + (int)(sb2 == se1) + (int)(se1 == sp4);
  const int v_zero = (int)(sc0 != sc1) + (int)(sc1 != sb0) + (int)(sb0 != se0) + (int)(se0 != sp4);
  const int v_plus = (int)(!sb2) + (int)(sb2 != se1) + (int)(se1 != sp4);
  const int below = v_minus - v_zero, above = v_zero - v_plus;
  return (below > 0 || above == 0) ? 1 : 2;
}

// Newton's iteration on a CONVEX piece of p from its left end x, where p(x) > 0: while p' < 0 the tangent's zero lies to the left
// of the first root of the piece, if there is one, so the iterates never pass it and converge to it; a piece without a root is
// left through its minimum (p' >= 0) or its right end.  Returns the root, or -1 if the piece holds none (-2: no decision).
TE_QDEV double quartic_root_convex_from_left(const double* c, double x, double right_end) {
  for (int it = 0; it < 100; ++it) {
    double f, df;
    quartic_eval(c, x, &f, &df);
    if (!(f > 0.0)) {
      // at or below zero: converged if that is the rounding noise of p's evaluation (Horner: a few units of roundoff of the
      // sum of the terms' magnitudes); anything else is not the situation described above
      const double ax = fabs(x);
      const double terms = fma(fma(fma(fma(fabs(c[4]), ax, fabs(c[3])), ax, fabs(c[2])), ax, fabs(c[1])), ax, fabs(c[0]));
      return (-f <= 16.0 * 2.220446049250313e-16 * terms) ? x : -2.0;
    }
    if (!(df < 0.0)) return -1.0;
    const double xn = x - (f * qdetail::rcp(df)) * (1.0 - 1.0e-15);
    if (!(xn <= right_end)) return -1.0;
    if (!(xn - x > 4.0 * 2.220446049250313e-16 * fabs(xn))) return xn;
    x = xn;
  }
  return -2.0;
}

// The smallest positive root of p when p(0) > 0, no root lies below zero and at least one above (quartic_sturm_classify == 2),
// without the critical points: the inflection points (closed form) cut [0, inf) into at most three pieces of constant convexity.
// A convex piece that starts above zero is searched from its left end (above); a concave one that starts above zero holds a
// root only if it ends below zero, and then exactly one.  Returns -2 when rounding leaves no clean decision (the caller falls
// back on the critical points).
TE_QDEV double first_positive_root(const double* c) {
  double infl[2];
  const int ni = quadratic_roots(6.0 * c[4], 3.0 * c[3], c[2], infl);
  const double inf = 1.0e308 * 10.0;
  double a = 0.0;                                           // left end of the current piece; p(a) > 0
  if (ni == 2 && infl[1] > 0.0) {
    double f, df;
    if (infl[0] > 0.0) {                                    // [0, i0] convex
      quartic_eval(c, infl[0], &f, &df);
      if (f == 0.0) return -2.0;
      if (f < 0.0) return quartic_root_monotone(c, 0.0, infl[0], true);
      const double r = quartic_root_convex_from_left(c, 0.0, infl[0]);
      if (r != -1.0) return r;
      a = infl[0];
    }
    quartic_eval(c, infl[1], &f, &df);                      // [a, i1] concave
    if (f == 0.0) return -2.0;
    if (f < 0.0) return quartic_root_monotone(c, a, infl[1], false);
    a = infl[1];
  }
  const double r = quartic_root_convex_from_left(c, a, inf);   // [a, inf) convex: the root is here
  return r == -1.0 ? -2.0 : r;
}

// coefficients lowest order first: c[0] + c[1] x + ... + c[4] x^4.
// Leftmost real root if it is >= 0, else -1 (also -1 for a zero leading coefficient and for no real root).
TE_QDEV double first_crossing_quartic(const double* cin) {
  if (!(fabs(cin[4]) > 0.0)) return -1.0;
  double c[5];
  const double sg = cin[4] < 0.0 ? -1.0 : 1.0;             // same roots, leading coefficient > 0
#pragma unroll
  for (int i = 0; i < 5; ++i) c[i] = sg * cin[i];
#ifndef TE_QUARTIC_NO_STURM   // (defined: the critical points for every target, as until round 3 -- the comparison in profiles/)
  const int cls = quartic_sturm_classify(c);
  TE_QTS(9);
  if (cls == 1) return -1.0;
  if (cls == 2) {
    const double r = first_positive_root(c);
    if (r >= 0.0) return r;
  }
#endif
  // critical points of p, ascending
  double m[3];
  TE_QTS(0);
  const int k = cubic_real_roots(4.0 * c[4], 3.0 * c[3], 2.0 * c[2], c[1], m);
  TE_QTS(1);
  double f, df;
  // the first decreasing piece (L, H) with p(L) > 0 > p(H): (-inf, m0) or (m1, m2)
  double L, H;
  bool have_L;
  quartic_eval(c, m[0], &f, &df);
  if (f < 0.0) { have_L = false; L = 0.0; H = m[0]; }
  else {
    if (k < 3) return -1.0;
    quartic_eval(c, m[2], &f, &df);
    if (!(f < 0.0)) return -1.0;
    have_L = true; L = m[1]; H = m[2];
  }
  if (!(H > 0.0)) return -1.0;                              // the root is below H <= 0
  if (!have_L || L < 0.0) {                                 // 0 lies on the piece: p(0) = c[0] decides the side
    if (c[0] < 0.0) return -1.0;
    if (c[0] == 0.0) return 0.0;
    L = 0.0;
  }
  TE_QTS(2);
  // inflection points inside (L, H) narrow the piece to constant convexity
  double infl[2];
  const int ni = quadratic_roots(6.0 * c[4], 3.0 * c[3], c[2], infl);
  bool convex = true;
  if (ni == 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double x = infl[j];
      if (x > L && x < H) {
        quartic_eval(c, x, &f, &df);
        if (f == 0.0) return x;
        if (f > 0.0) L = x; else H = x;
      }
    }
    const double mid = 0.5 * (L + H);
    convex = !(mid > infl[0] && mid < infl[1]);            // p'' < 0 exactly between the inflection points
  }
  TE_QTS(3);
  return quartic_root_monotone(c, L, H, convex);
}

#ifndef TE_QUARTIC_HOST
#pragma clang fp contract(fast)
#endif

}  // namespace te
