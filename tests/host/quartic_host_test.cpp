// quartic_host_test.cpp -- the device solver of the sphere-intersection quartic
// (target_estimation_amd/csrc/te_quartic.hpp, compiled for the host) against the oracle's
// long-double Aberth roots (oracle/te_oracle.c, orc_lowest_real_root).  Test code: links the oracle.
#define TE_QUARTIC_HOST
#include "../../target_estimation_amd/csrc/te_quartic.hpp"

#include <cstdio>
#include <cstdlib>
#include <random>

extern "C" double orc_lowest_real_root(const double* coeffs, int ncoeffs);

static long n_cases = 0, n_class = 0, n_val = 0, n_hits = 0;
static double worst = 0;

static long n_settled = 0, n_direct = 0, n_long_road = 0, n_claim = 0;

static void check(const double* c) {
  double want = orc_lowest_real_root(c, 5);
  if (want < 0) want = -1;                      // the caller's mapping (src/intersection_solver.cpp:83)
  const double got = te::first_crossing_quartic(c);
  ++n_cases;
  // what the Sturm classification claims must hold by the oracle's roots, whatever the solver does with it afterwards
  if (std::fabs(c[4]) > 0.0) {
    double cc[5];
    for (int k = 0; k < 5; ++k) cc[k] = c[4] < 0 ? -c[k] : c[k];
    const int cls = te::quartic_sturm_classify(cc);
    if (cls == 1) { ++n_settled; if (want != -1) { ++n_claim; printf("claim 1: c = %.17g %.17g %.17g %.17g %.17g want %.17g\n", c[0], c[1], c[2], c[3], c[4], want); } }
    else if (cls == 2) { ++n_direct; if (want == -1) { ++n_claim; printf("claim 2: c = %.17g %.17g %.17g %.17g %.17g want -1\n", c[0], c[1], c[2], c[3], c[4]); } }
    else ++n_long_road;
  }
  if ((want == -1) != (got == -1)) { ++n_class; printf("class: c = %.17g %.17g %.17g %.17g %.17g want %.17g got %.17g\n", c[0], c[1], c[2], c[3], c[4], want, got); return; }
  if (want == -1) return;
  ++n_hits;
  const double rel = std::fabs(got - want) / std::fmax(std::fabs(want), 1e-300);
  if (rel > worst) worst = rel;
  if (rel > 1e-9) { ++n_val; printf("value: c = %.17g %.17g %.17g %.17g %.17g want %.17g got %.17g\n", c[0], c[1], c[2], c[3], c[4], want, got); }
}

// Grazing trajectories: whether the pair of roots at the closest approach is real is decided in the last digits of the
// coefficients, and a solver is free to differ from the long-double roots there (te_quartic.hpp: "solver-specific either way").
// What must hold: a crossing reported is a root (p changes sign around it within rounding), and the classification's claims
// are never contradicted by roots the oracle finds CLEARLY (well separated from a double root).
static long n_graze = 0, n_graze_bad = 0;
static void check_grazing(const double* c) {
  ++n_graze;
  const double got = te::first_crossing_quartic(c);
  if (got == -1) return;
  auto pv = [&](double x) { return (((c[4] * x + c[3]) * x + c[2]) * x + c[1]) * x + c[0]; };
  const double h = 1e-6 * std::fmax(1.0, std::fabs(got));
  const double scale = std::fabs(c[0]) + std::fabs(c[1] * got) + std::fabs(c[2] * got * got) + std::fabs(c[3] * got * got * got) + std::fabs(c[4] * got * got * got * got);
  if (!(got >= 0) || std::fabs(pv(got)) > 1e-9 * scale || !(pv(got - h) >= -1e-12 * scale)) {
    ++n_graze_bad;
    printf("grazing: c = %.17g %.17g %.17g %.17g %.17g got %.17g p %.3g\n", c[0], c[1], c[2], c[3], c[4], got, pv(got));
  }
}

// quartic_host_test [seed [size multiplier]]   (the test runs it without arguments: seed 7, multiplier 1)
int main(int argc, char** argv) {
  const unsigned long long seed = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 7ull;
  const long mult = argc > 2 ? std::atol(argv[2]) : 1;
  std::mt19937_64 g(seed);
  std::normal_distribution<double> N(0, 1);
  std::uniform_real_distribution<double> U(0, 1);
  // sphere scenes, acceleration scale swept over 14 decades (tiny leading coefficients included)
  for (double asc : {1e2, 1.0, 1e-2, 1e-4, 1e-6, 1e-9, 1e-12})
    for (long i = 0; i < 1500 * mult; ++i) {
      double p[3], v[3], a[3];
      for (int k = 0; k < 3; ++k) { p[k] = -10 + 20 * U(g); v[k] = 3 * N(g); a[k] = asc * N(g); }
      const double R = 0.5 + 8 * U(g);
      const double c[5] = {p[0] * p[0] + p[1] * p[1] + p[2] * p[2] - R * R, 2 * (p[0] * v[0] + p[1] * v[1] + p[2] * v[2]),
                           v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + p[0] * a[0] + p[1] * a[1] + p[2] * a[2],
                           v[0] * a[0] + v[1] * a[1] + v[2] * a[2], 0.25 * (a[0] * a[0] + a[1] * a[1] + a[2] * a[2])};
      check(c);
    }
  // the shape the fused query meets in BASELINE configs[4] (|p| ~ 10, |v| ~ 0.3, |a| ~ 1e-2 .. 1e-3, R = 1), and grazing
  // trajectories: the closest approach within R (1 +- 1e-3 .. 1e-12) of the sphere, ahead of or behind the target
  for (long i = 0; i < 20000 * mult; ++i) {
    double p[3], v[3], a[3];
    const double asc = (i & 1) ? 1e-2 : 1e-3;
    for (int k = 0; k < 3; ++k) { p[k] = 6 * N(g); v[k] = 0.2 * N(g); a[k] = asc * N(g); }
    const double c[5] = {p[0] * p[0] + p[1] * p[1] + p[2] * p[2] - 1.0, 2 * (p[0] * v[0] + p[1] * v[1] + p[2] * v[2]),
                         v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + p[0] * a[0] + p[1] * a[1] + p[2] * a[2],
                         v[0] * a[0] + v[1] * a[1] + v[2] * a[2], 0.25 * (a[0] * a[0] + a[1] * a[1] + a[2] * a[2])};
    check(c);
  }
  for (long i = 0; i < 6000 * mult; ++i) {
    // straight line through the point n * d (|n| = 1, d = R (1 + eps)) along a direction orthogonal to n, a small acceleration on top
    double n[3], t[3], a[3];
    double nn = 0;
    for (int k = 0; k < 3; ++k) { n[k] = N(g); nn += n[k] * n[k]; }
    nn = std::sqrt(nn);
    for (int k = 0; k < 3; ++k) n[k] /= nn;
    double tn = 0;
    for (int k = 0; k < 3; ++k) { t[k] = N(g); tn += t[k] * n[k]; }
    for (int k = 0; k < 3; ++k) t[k] -= tn * n[k];
    const double eps = (U(g) < 0.5 ? -1 : 1) * std::pow(10.0, -3 - 9 * U(g));
    const double R = 1.0, d = R * (1 + eps), s0 = (U(g) < 0.5 ? -1 : 1) * (1 + 10 * U(g));   // the target is s0 before / after the closest point
    const double asc = std::pow(10.0, -8 + 6 * U(g));
    double p[3];
    for (int k = 0; k < 3; ++k) { p[k] = n[k] * d - t[k] * s0; a[k] = asc * N(g); }
    const double c[5] = {p[0] * p[0] + p[1] * p[1] + p[2] * p[2] - R * R, 2 * (p[0] * t[0] + p[1] * t[1] + p[2] * t[2]),
                         t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + p[0] * a[0] + p[1] * a[1] + p[2] * a[2],
                         t[0] * a[0] + t[1] * a[1] + t[2] * a[2], 0.25 * (a[0] * a[0] + a[1] * a[1] + a[2] * a[2])};
    check_grazing(c);
  }
  // random coefficients over 12 decades, both leading signs
  for (long i = 0; i < 8000 * mult; ++i) {
    double c[5];
    for (int k = 0; k < 5; ++k) c[k] = N(g) * std::pow(10.0, -6 + 12 * U(g));
    check(c);
  }
  // prescribed roots (four real / two real + a complex pair / two pairs), well separated
  for (long i = 0; i < 8000 * mult; ++i) {
    const double lead = (U(g) < 0.5 ? -1 : 1) * std::pow(10.0, -3 + 6 * U(g));
    double q1[3], q2[3];
    for (double* q : {q1, q2}) {
      if (U(g) < 0.6) { const double r1 = 10 * N(g), r2 = r1 + (0.05 + std::fabs(10 * N(g))) * (U(g) < 0.5 ? -1 : 1); q[0] = r1 * r2; q[1] = -(r1 + r2); }
      else { const double re = 5 * N(g), im = 3 * U(g) + 0.05; q[0] = re * re + im * im; q[1] = -2 * re; }
      q[2] = 1;
    }
    double c[5] = {0, 0, 0, 0, 0};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) c[a + b] += q1[a] * q2[b];
    for (int k = 0; k < 5; ++k) c[k] *= lead;
    check(c);
  }
  // semantics
  const double zero_lead[5] = {1, 2, 1, 0, 0};                         // leading coefficient 0 -> -1
  const double none[5] = {1, 0, 1, 0, 1};                              // x^4 + x^2 + 1: no real root
  const double neg_first[5] = {-24 * 5, 2 * 5, 23 * 5 - 20, 8 - 5, 1};  // (x+5)(x-1)(x+... ) has a negative leftmost root
  int bad = 0;
  if (te::first_crossing_quartic(zero_lead) != -1) { printf("zero leading coefficient\n"); ++bad; }
  if (te::first_crossing_quartic(none) != -1) { printf("no real root\n"); ++bad; }
  {  // (x - 1)(x - 2)(x - 3)(x - 4) -> 1;  (x + 5)(x - 2)(x - 3)(x - 4) -> -1 (leftmost root negative)
    const double a[5] = {24, -50, 35, -10, 1}, b[5] = {-120, 106, -9, -4, 1};
    if (std::fabs(te::first_crossing_quartic(a) - 1.0) > 1e-12) { printf("1234\n"); ++bad; }
    if (te::first_crossing_quartic(b) != -1) { printf("-5 2 3 4\n"); ++bad; }
    const double z[5] = {0, -6, 11, -6, 1};                            // x (x-1)(x-2)(x-3): root exactly at 0
    if (te::first_crossing_quartic(z) != 0.0) { printf("root at 0: %g\n", te::first_crossing_quartic(z)); ++bad; }
  }
  (void)neg_first;
  printf("cases %ld (with a crossing: %ld), class mismatches %ld, value mismatches %ld, worst rel %.3g, semantics failures %d\n",
         n_cases, n_hits, n_class, n_val, worst, bad);
  printf("Sturm classification: settled as -1 %ld, crossing found without the critical points %ld, the long road %ld; claims contradicted %ld\n",
         n_settled, n_direct, n_long_road, n_claim);
  printf("grazing trajectories %ld, bad crossings %ld\n", n_graze, n_graze_bad);
  const bool ok = n_class == 0 && n_val == 0 && bad == 0 && n_hits > 3000 && n_claim == 0 && n_graze_bad == 0 && n_settled > 10000 && n_direct > 3000 &&
                  n_long_road > 0;
  printf(ok ? "quartic host test ok\n" : "QUARTIC HOST TEST FAILED\n");
  return ok ? 0 : 1;
}
