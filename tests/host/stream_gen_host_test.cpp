// The product's stream generator (csrc/stream_gen.hpp) compiled for the host, held to BIT equality with the oracle's
// C twin (oracle/te_stream.c), and its log / sin / cos sequences held to libm within a few ulp.
// Build: g++ -std=c++17 -O2 -ffp-contract=off (tests/test_stream_gen.py).
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../oracle/te_oracle.h"
#include "../../target_estimation_amd/csrc/stream_gen.hpp"

static bool same_bits(double a, double b) { return std::memcmp(&a, &b, 8) == 0; }

int main() {
  using namespace te::sg;
  long bad = 0, checked = 0;
  // elementary sequences against libm
  double worst_log = 0, worst_sc = 0;
  for (int i = 0; i < 200000; ++i) {
    const double u = uniform01(key(11, i, 0, 0));
    const double l = log_det(u), lr = std::log(u);
    const double e = std::fabs(l - lr) / std::fabs(lr);
    if (e > worst_log) worst_log = e;
    const double a = (i % 2 ? 6.283185307179586 : 3000.0) * uniform01(key(12, i, 0, 1));
    double s, c;
    sincos_det(a, &s, &c);
    const double es = std::fabs(s - std::sin(a)), ec = std::fabs(c - std::cos(a));
    if (es > worst_sc) worst_sc = es;
    if (ec > worst_sc) worst_sc = ec;
  }
  std::printf("log_det: worst relative error %.3g; sincos_det: worst absolute error %.3g\n", worst_log, worst_sc);
  if (!(worst_log < 1e-15) || !(worst_sc < 1e-15)) { std::printf("FAIL: elementary sequences\n"); return 1; }
  // full streams against the oracle twin, every model, both variants
  for (int model = 0; model < 4; ++model) {
    for (int variant = 0; variant < 2; ++variant) {
      const double avail = variant ? 0.8 : 1.0, rn = variant ? 0.1 : 0.0, dt = 0.004;
      const unsigned long long seed = 20240000ull + model * 7 + variant;
      for (long i = 0; i < 300; ++i) {
        const long target = 1000003 * i + (variant ? 123456789012l : 0);
        const Truth tr = truth_of(model, seed, (uint64_t)target);
        double t12[12], p0o[7], p0[7];
        orc_stream_truth(model, seed, target, t12, p0o);
        init_pose(tr, seed, (uint64_t)target, p0);
        for (int c = 0; c < 3; ++c) {
          bad += !same_bits(tr.p[c], t12[c]) + !same_bits(tr.v[c], t12[3 + c]) + !same_bits(tr.a[c], t12[6 + c]) + !same_bits(tr.w[c], t12[9 + c]);
        }
        for (int c = 0; c < 7; ++c) bad += !same_bits(p0[c], p0o[c]);
        for (long s = 0; s < 40; ++s) {
          const long tick = s < 20 ? s : 1000 * s;
          double m[7], mo[7];
          const bool got = measurement(tr, seed, (uint64_t)target, (uint32_t)tick, dt, avail, rn, m);
          const int goto_ = orc_stream_measurement(model, seed, target, tick, dt, avail, rn, mo);
          bad += (got ? 1 : 0) != goto_;
          for (int c = 0; c < 7; ++c) bad += !same_bits(m[c], mo[c]);
          checked += 8;
          const double qn = std::sqrt(m[3] * m[3] + m[4] * m[4] + m[5] * m[5] + m[6] * m[6]);
          if (std::fabs(qn - 1.0) > 1e-14) { std::printf("FAIL: quaternion norm %.17g\n", qn); return 1; }
        }
      }
    }
  }
  std::printf("%ld values compared, %ld differ\n", checked, bad);
  if (bad) { std::printf("FAIL\n"); return 1; }
  std::printf("stream generator host test ok\n");
  return 0;
}
