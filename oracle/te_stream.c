/*
 * te_stream.c -- CPU twin of the synthetic stream generator.  TEST INFRASTRUCTURE ONLY (see te_oracle.h).
 *
 * The product fills its measurement rings on the GPU with a counter-based generator (the definition is in the header
 * comment of target_estimation_amd/csrc/stream_gen.hpp); this file restates that definition in plain C so that a CPU
 * checker regenerates the identical doubles without copying anything back from the device.  What is
 * generated widens the measurement generator of the reference's integration test (test/target_manager_test.cpp:82-115:
 * straight line + N(0, 0.01^2) on xyz :102-104, quaternion advanced by Qtran(dt, omega) and renormalised :106-113) to a
 * population of targets.
 *
 * Bit equality with the device needs the same rounding sequence: build with -ffp-contract=off (oracle/Makefile does, for
 * both libraries), only + - * / sqrt on doubles, log and sin/cos by the fixed polynomial sequences below.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "te_oracle.h"

#define SG_GOLDEN 0x9E3779B97F4A7C15ull
#define SG_STATIC 0xFFFFFFFFu

static uint64_t sg_mix(uint64_t z) { /* splitmix64 finaliser */
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static uint64_t sg_key(uint64_t seed, uint64_t target, uint32_t tick, uint32_t comp) {
  uint64_t z = sg_mix(seed + SG_GOLDEN * (target + 1));
  z = sg_mix(z + SG_GOLDEN * ((uint64_t)tick + 1));
  return sg_mix(z + SG_GOLDEN * ((uint64_t)comp + 1));
}

static double sg_u01(uint64_t k) { return ((double)(k >> 11) + 0.5) * 0x1.0p-53; }

/* ln x, x normal and positive: x = 2^k f, f in [sqrt(1/2), sqrt(2)); ln f = 2 atanh((f-1)/(f+1)) by its series to z^12 */
static double sg_log(double x) {
  static const double odd[12] = {1.0 / 25.0, 1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0,
                                 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0,  1.0 / 7.0,  1.0 / 5.0,  1.0 / 3.0};
  uint64_t b;
  memcpy(&b, &x, 8);
  int k = (int)((b >> 52) & 0x7FF) - 1023;
  const uint64_t m = b & 0x000FFFFFFFFFFFFFull;
  if (m >= 0x6A09E667F3BCDull) {
    k += 1;
    b = m | 0x3FE0000000000000ull;
  } else {
    b = m | 0x3FF0000000000000ull;
  }
  double f;
  memcpy(&f, &b, 8);
  const double s = (f - 1.0) / (f + 1.0);
  const double z = s * s;
  double p = odd[0];
  for (int i = 1; i < 12; ++i) p = p * z + odd[i];
  const double lf = 2.0 * s + 2.0 * s * (z * p);
  const double kd = (double)k;
  return kd * 0x1.62e42fee00000p-1 + (lf + kd * 0x1.a39ef35793c76p-33);
}

/* sin, cos of |a| < 2^20 pi/2: three-part pi/2 reduction, Taylor series on [-pi/4, pi/4] */
static void sg_sincos(double a, double* s_out, double* c_out) {
  static const double sc[10] = {-1.0 / 51090942171709440000.0, 1.0 / 121645100408832000.0, -1.0 / 355687428096000.0,
                                1.0 / 1307674368000.0,         -1.0 / 6227020800.0,        1.0 / 39916800.0,
                                -1.0 / 362880.0,               1.0 / 5040.0,               -1.0 / 120.0,
                                1.0 / 6.0};
  static const double cc[9] = {1.0 / 2432902008176640000.0, -1.0 / 6402373705728000.0, 1.0 / 20922789888000.0,
                               -1.0 / 87178291200.0,        1.0 / 479001600.0,         -1.0 / 3628800.0,
                               1.0 / 40320.0,               -1.0 / 720.0,              1.0 / 24.0};
  const double t = a * 0x1.45f306dc9c883p-1;
  const double qd = (double)(long long)(t < 0.0 ? t - 0.5 : t + 0.5);
  const long long q = (long long)qd;
  double r = a - qd * 0x1.921fb54400000p+0;
  r = r - qd * 0x1.0b4611a600000p-34;
  r = r - qd * 0x1.3198a2e037073p-69;
  const double z = r * r;
  double ps = sc[0], pc = cc[0];
  for (int i = 1; i < 10; ++i) ps = ps * z + sc[i];
  for (int i = 1; i < 9; ++i) pc = pc * z + cc[i];
  const double sn = r - r * (z * ps);
  const double cs = (1.0 - 0.5 * z) + z * (z * pc);
  switch ((int)(q & 3)) {
    case 0: *s_out = sn; *c_out = cs; break;
    case 1: *s_out = cs; *c_out = -sn; break;
    case 2: *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn; break;
  }
}

static double sg_normal(uint64_t seed, uint64_t target, uint32_t tick, uint32_t comp) { /* Box-Muller, cosine branch */
  const double u1 = sg_u01(sg_key(seed, target, tick, comp));
  const double u2 = sg_u01(sg_key(seed, target, tick, comp + 32));
  double sn, cs;
  sg_sincos(6.283185307179586 * u2, &sn, &cs);
  return sqrt(-2.0 * sg_log(u1)) * cs;
}

static double sg_uniform(uint64_t seed, uint64_t target, uint32_t tick, uint32_t comp, double lo, double hi) {
  return lo + (hi - lo) * sg_u01(sg_key(seed, target, tick, comp));
}

/* truth12 = p(3) v(3) a(3) omega(3) of target `target` */
void orc_stream_truth(int model, unsigned long long seed, long target, double* truth12, double* pose0) {
  const uint64_t tg = (uint64_t)target;
  for (int c = 0; c < 3; ++c) {
    truth12[c] = sg_uniform(seed, tg, SG_STATIC, c, -10.0, 10.0);
    truth12[3 + c] = sg_uniform(seed, tg, SG_STATIC, 3 + c, -1.0, 1.0);
    truth12[6 + c] = 0.0;
  }
  if (model == ORC_UNIFORM_ACCELERATION)
    for (int c = 0; c < 3; ++c) truth12[6 + c] = (c == 2 ? -9.81 : 0.0) + sg_uniform(seed, tg, SG_STATIC, 6 + c, -0.1, 0.1);
  truth12[9] = sg_uniform(seed, tg, SG_STATIC, 9, -3.0, 3.0);
  truth12[10] = sg_uniform(seed, tg, SG_STATIC, 10, -0.1, 0.1);
  truth12[11] = sg_uniform(seed, tg, SG_STATIC, 11, -0.1, 0.1);
  if (pose0) {
    for (int c = 0; c < 3; ++c) pose0[c] = truth12[c] + 0.01 * sg_normal(seed, tg, SG_STATIC, 12 + c);
    pose0[3] = 0.0; pose0[4] = 0.0; pose0[5] = 0.0; pose0[6] = 1.0;
  }
}

/* one measurement row [x y z qx qy qz qw] of tick `tick` (time (tick+1) dt); returns has_meas */
int orc_stream_measurement(int model, unsigned long long seed, long target, long tick, double dt, double availability,
                           double rpy_noise, double* meas7) {
  double tr[12];
  orc_stream_truth(model, seed, target, tr, 0);
  const uint64_t tg = (uint64_t)target;
  const uint32_t s = (uint32_t)tick;
  const double t = (double)(s + 1u) * dt;
  for (int c = 0; c < 3; ++c)
    meas7[c] = ((tr[c] + tr[3 + c] * t) + (0.5 * tr[6 + c]) * (t * t)) + 0.01 * sg_normal(seed, tg, s, c);
  const double* w = tr + 9;
  const double n2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
  double q[4] = {0.0, 0.0, 0.0, 1.0};
  if (n2 > 0.0) {
    /* Qtran(dt, omega)^(tick+1) on the identity, geometry.hpp:448-465,:493-504: rotation by |omega| t about omega */
    const double nw = sqrt(n2);
    double sn, cs;
    sg_sincos(0.5 * (nw * t), &sn, &cs);
    const double k = sn / nw;
    q[0] = k * w[0]; q[1] = k * w[1]; q[2] = k * w[2]; q[3] = cs;
    const double qn = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    for (int c = 0; c < 4; ++c) q[c] = q[c] / qn;
  }
  if (rpy_noise > 0.0) {
    double h[4];
    for (int c = 0; c < 3; ++c) h[c] = (0.5 * rpy_noise) * sg_normal(seed, tg, s, 3 + c);
    h[3] = 1.0;
    const double hn = sqrt(((h[0] * h[0] + h[1] * h[1]) + h[2] * h[2]) + h[3] * h[3]);
    for (int c = 0; c < 4; ++c) h[c] = h[c] / hn;
    const double x1 = q[0], y1 = q[1], z1 = q[2], w1 = q[3], x2 = h[0], y2 = h[1], z2 = h[2], w2 = h[3];
    q[0] = ((w1 * x2 + x1 * w2) + y1 * z2) - z1 * y2;
    q[1] = ((w1 * y2 - x1 * z2) + y1 * w2) + z1 * x2;
    q[2] = ((w1 * z2 + x1 * y2) - y1 * x2) + z1 * w2;
    q[3] = ((w1 * w2 - x1 * x2) - y1 * y2) - z1 * z2;
  }
  for (int c = 0; c < 4; ++c) meas7[3 + c] = q[c];
  if (availability >= 1.0) return 1;
  return sg_u01(sg_key(seed, tg, s, 6)) < availability ? 1 : 0;
}

/* the whole block: meas [n_ticks][n_targets][7] (the reference's row layout, doubles; f32 != 0 rounds every value to
 * float first, as a float ring on the device holds it), has [n_ticks][n_targets] or NULL, pose0 [n][7] / truth [n][12] or NULL */
void orc_stream_fill(int model, unsigned long long seed, long first_target, long n_targets, long first_tick, long n_ticks,
                     double dt, double availability, double rpy_noise, int f32, double* meas, unsigned char* has,
                     double* pose0, double* truth) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n_targets; ++i) {
    double tr[12];
    orc_stream_truth(model, seed, first_target + i, truth ? truth + 12 * i : tr, pose0 ? pose0 + 7 * i : 0);
  }
  if (!meas) return;
#pragma omp parallel for schedule(static)
  for (long s = 0; s < n_ticks; ++s) {
    for (long i = 0; i < n_targets; ++i) {
      double* row = meas + (s * n_targets + i) * 7;
      const int got = orc_stream_measurement(model, seed, first_target + i, first_tick + s, dt, availability, rpy_noise, row);
      if (f32)
        for (int c = 0; c < 7; ++c) row[c] = (double)(float)row[c];
      if (has) has[s * n_targets + i] = (unsigned char)got;
    }
  }
}
