// stream_gen.hpp -- counter-based synthetic measurement streams (SURVEY 8d "Synthetic inputs").
//
// What it produces is what the reference's integration test feeds its filters (test/target_manager_test.cpp:82-115:
// a straight line + N(0, 0.01^2) on xyz, a noiseless quaternion advanced by Qtran(dt, omega) :106-113), widened to a
// population: every target has its own start, velocity, (acceleration,) body rate.  Every number is a pure function of
// the KEY (seed, target, tick, component) -- no generator state -- so any (target, tick) can be produced anywhere, in any
// order, on any device: the GPU fills its measurement ring with one kernel, a CPU checker regenerates the identical
// doubles (tests/test_stream_gen.py holds the two implementations to bit equality).
//
// Bit equality across host and device needs arithmetic that rounds the same everywhere: only + - * / sqrt (IEEE, no
// contraction: the pragma below) and integer operations are used; log and sin/cos are evaluated by the fixed polynomial
// sequences below instead of libm / ocml (whose results differ in the last place).
//
//   mix(z)              splitmix64 finaliser
//   key(seed,t,s,c)     mix(mix(mix(seed + G (t+1)) + G (s+1)) + G (c+1)),  G = 0x9E3779B97F4A7C15
//   U(key)              ((key >> 11) + 0.5) 2^-53                      in (0, 1)
//   N(seed,t,s,c)       sqrt(-2 ln U(key(..,c))) cos(2 pi U(key(..,c+32)))  (Box-Muller, one output per pair)
//
// Per-target constants (tick index STATIC = 2^32-1): components 0-2 position p ~ U(-10,10)^3, 3-5 velocity v ~ U(-1,1)^3,
// 6-8 acceleration a = (0,0,-9.81) + U(-0.1,0.1)^3 (uniform_acceleration only, else 0), 9-11 body rate
// omega ~ U(-3,3) x U(-0.1,0.1)^2 (keeps pitch away from +-pi/2), 12-14 noise of the initial pose (N, sigma 0.01).
// Per tick s (time t = (s+1) dt): components 0-2 position noise (N, sigma 0.01), 3-5 orientation noise (half rotation
// vector, N, sigma rpy_noise / 2), 6 availability draw (U < availability -> the target has a measurement).
//   position    p + v t + a t^2 / 2 + noise
//   orientation q(t) = normalise([sin(|omega| t / 2) omega / |omega|, cos(|omega| t / 2)])  [x y z w]
//               = Qtran(dt, omega)^(s+1) applied to the identity (geometry.hpp:448-465, :493-504), in closed form
//               (x normalise([h, 1]) from the right when rpy_noise > 0)
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

#if defined(__HIPCC__)
#define TE_SG_FN __host__ __device__ inline
#else
#define TE_SG_FN inline
#endif

namespace te {
namespace sg {

#pragma clang fp contract(off)

constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
constexpr uint32_t kStatic = 0xFFFFFFFFu;   // "tick" of the per-target constants

TE_SG_FN uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
TE_SG_FN uint64_t key(uint64_t seed, uint64_t target, uint32_t tick, uint32_t comp) {
  uint64_t z = mix(seed + kGolden * (target + 1));
  z = mix(z + kGolden * ((uint64_t)tick + 1));
  return mix(z + kGolden * ((uint64_t)comp + 1));
}
TE_SG_FN double uniform01(uint64_t k) { return ((double)(k >> 11) + 0.5) * 0x1.0p-53; }

TE_SG_FN double bits_to_double(uint64_t b) {
  union { uint64_t u; double d; } v;
  v.u = b;
  return v.d;
}
TE_SG_FN uint64_t double_to_bits(double d) {
  union { uint64_t u; double d; } v;
  v.d = d;
  return v.u;
}

// ln x for a normal positive double: x = 2^k f with f in [sqrt(1/2), sqrt(2)), s = (f-1)/(f+1),
// ln f = 2 atanh s = 2 s (1 + z/3 + z^2/5 + ... + z^12/25), z = s^2 <= 0.0295 (truncation < 1e-20)
TE_SG_FN double log_det(double x) {
  uint64_t b = double_to_bits(x);
  int k = (int)((b >> 52) & 0x7FF) - 1023;
  uint64_t m = b & 0x000FFFFFFFFFFFFFull;
  if (m >= 0x6A09E667F3BCDull) k += 1, b = m | 0x3FE0000000000000ull;   // f in [sqrt(2)/2, 1)
  else b = m | 0x3FF0000000000000ull;                                     // f in [1, sqrt(2))
  const double f = bits_to_double(b);
  const double s = (f - 1.0) / (f + 1.0);
  const double z = s * s;
  double p = 1.0 / 25.0;
  p = p * z + 1.0 / 23.0;
  p = p * z + 1.0 / 21.0;
  p = p * z + 1.0 / 19.0;
  p = p * z + 1.0 / 17.0;
  p = p * z + 1.0 / 15.0;
  p = p * z + 1.0 / 13.0;
  p = p * z + 1.0 / 11.0;
  p = p * z + 1.0 / 9.0;
  p = p * z + 1.0 / 7.0;
  p = p * z + 1.0 / 5.0;
  p = p * z + 1.0 / 3.0;
  const double lf = 2.0 * s + 2.0 * s * (z * p);
  const double kd = (double)k;
  return kd * 0x1.62e42fee00000p-1 + (lf + kd * 0x1.a39ef35793c76p-33);   // ln 2 = hi (33 bits) + lo
}

// sin and cos of a finite angle |a| < 2^20 pi/2: quadrant q = round(a 2/pi), r = a - q pi/2 by a three-part pi/2
// (the first two products are exact for |q| < 2^20), Taylor series on |r| <= pi/4 (sin to r^21, cos to r^20)
TE_SG_FN void sincos_det(double a, double* s_out, double* c_out) {
  const double t = a * 0x1.45f306dc9c883p-1;            // a * 2/pi
  const double qd = (double)(long long)(t < 0.0 ? t - 0.5 : t + 0.5);
  const long long q = (long long)qd;
  double r = a - qd * 0x1.921fb54400000p+0;            // pi/2, leading 33 bits
  r = r - qd * 0x1.0b4611a600000p-34;                   // next 33 bits
  r = r - qd * 0x1.3198a2e037073p-69;                   // the rest
  const double z = r * r;
  double ps = -1.0 / 51090942171709440000.0;            // -1/21!
  ps = ps * z + 1.0 / 121645100408832000.0;             //  1/19!
  ps = ps * z - 1.0 / 355687428096000.0;                // -1/17!
  ps = ps * z + 1.0 / 1307674368000.0;                  //  1/15!
  ps = ps * z - 1.0 / 6227020800.0;                     // -1/13!
  ps = ps * z + 1.0 / 39916800.0;                       //  1/11!
  ps = ps * z - 1.0 / 362880.0;                         // -1/9!
  ps = ps * z + 1.0 / 5040.0;                           //  1/7!
  ps = ps * z - 1.0 / 120.0;                            // -1/5!
  ps = ps * z + 1.0 / 6.0;                              //  1/3!
  const double sn = r - r * (z * ps);
  double pc = 1.0 / 2432902008176640000.0;              //  1/20!
  pc = pc * z - 1.0 / 6402373705728000.0;               // -1/18!
  pc = pc * z + 1.0 / 20922789888000.0;                 //  1/16!
  pc = pc * z - 1.0 / 87178291200.0;                    // -1/14!
  pc = pc * z + 1.0 / 479001600.0;                      //  1/12!
  pc = pc * z - 1.0 / 3628800.0;                        // -1/10!
  pc = pc * z + 1.0 / 40320.0;                          //  1/8!
  pc = pc * z - 1.0 / 720.0;                            // -1/6!
  pc = pc * z + 1.0 / 24.0;                             //  1/4!
  const double cs = (1.0 - 0.5 * z) + z * (z * pc);
  switch ((int)(q & 3)) {
    case 0: *s_out = sn; *c_out = cs; break;
    case 1: *s_out = cs; *c_out = -sn; break;
    case 2: *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn; break;
  }
}

#if defined(__HIPCC__)
TE_SG_FN double sqrt_rn(double x) { return __builtin_sqrt(x); }   // correctly rounded on host and on gfx950 (v_sqrt_f64 + fix-up)
#else
TE_SG_FN double sqrt_rn(double x) { return __builtin_sqrt(x); }
#endif

TE_SG_FN double normal(uint64_t seed, uint64_t target, uint32_t tick, uint32_t comp) {
  const double u1 = uniform01(key(seed, target, tick, comp));
  const double u2 = uniform01(key(seed, target, tick, comp + 32));
  double sn, cs;
  sincos_det(6.283185307179586 * u2, &sn, &cs);
  return sqrt_rn(-2.0 * log_det(u1)) * cs;
}
TE_SG_FN double uniform(uint64_t seed, uint64_t target, uint32_t tick, uint32_t comp, double lo, double hi) {
  return lo + (hi - lo) * uniform01(key(seed, target, tick, comp));
}

struct Truth {
  double p[3], v[3], a[3], w[3];
};

// model: TargetManager::target_t (target_manager.hpp:38); 2 = uniform_acceleration
TE_SG_FN Truth truth_of(int model, uint64_t seed, uint64_t target) {
  Truth tr;
  for (int c = 0; c < 3; ++c) {
    tr.p[c] = uniform(seed, target, kStatic, c, -10.0, 10.0);
    tr.v[c] = uniform(seed, target, kStatic, 3 + c, -1.0, 1.0);
    tr.a[c] = 0.0;
  }
  if (model == 2) {
    for (int c = 0; c < 3; ++c) tr.a[c] = (c == 2 ? -9.81 : 0.0) + uniform(seed, target, kStatic, 6 + c, -0.1, 0.1);
  }
  tr.w[0] = uniform(seed, target, kStatic, 9, -3.0, 3.0);
  tr.w[1] = uniform(seed, target, kStatic, 10, -0.1, 0.1);
  tr.w[2] = uniform(seed, target, kStatic, 11, -0.1, 0.1);
  return tr;
}

// pose handed to TargetManager::init: the start position with the measurement noise, identity orientation
TE_SG_FN void init_pose(const Truth& tr, uint64_t seed, uint64_t target, double* pose7) {
  for (int c = 0; c < 3; ++c) pose7[c] = tr.p[c] + 0.01 * normal(seed, target, kStatic, 12 + c);
  pose7[3] = 0.0; pose7[4] = 0.0; pose7[5] = 0.0; pose7[6] = 1.0;
}

// measurement of tick `tick` (time (tick+1) dt); returns whether the target has one on this tick
TE_SG_FN bool measurement(const Truth& tr, uint64_t seed, uint64_t target, uint32_t tick, double dt, double availability,
                          double rpy_noise, double* meas7) {
  const double t = (double)(tick + 1u) * dt;
  for (int c = 0; c < 3; ++c)
    meas7[c] = ((tr.p[c] + tr.v[c] * t) + (0.5 * tr.a[c]) * (t * t)) + 0.01 * normal(seed, target, tick, c);
  const double n2 = (tr.w[0] * tr.w[0] + tr.w[1] * tr.w[1]) + tr.w[2] * tr.w[2];
  double q[4] = {0.0, 0.0, 0.0, 1.0};
  if (n2 > 0.0) {
    const double nw = sqrt_rn(n2);
    double sn, cs;
    sincos_det(0.5 * (nw * t), &sn, &cs);
    const double k = sn / nw;
    q[0] = k * tr.w[0]; q[1] = k * tr.w[1]; q[2] = k * tr.w[2]; q[3] = cs;
    const double qn = sqrt_rn(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    for (int c = 0; c < 4; ++c) q[c] = q[c] / qn;
  }
  if (rpy_noise > 0.0) {
    double h[4];
    for (int c = 0; c < 3; ++c) h[c] = (0.5 * rpy_noise) * normal(seed, target, tick, 3 + c);
    h[3] = 1.0;
    const double hn = sqrt_rn(((h[0] * h[0] + h[1] * h[1]) + h[2] * h[2]) + h[3] * h[3]);
    for (int c = 0; c < 4; ++c) h[c] = h[c] / hn;
    const double x1 = q[0], y1 = q[1], z1 = q[2], w1 = q[3], x2 = h[0], y2 = h[1], z2 = h[2], w2 = h[3];
    q[0] = ((w1 * x2 + x1 * w2) + y1 * z2) - z1 * y2;
    q[1] = ((w1 * y2 - x1 * z2) + y1 * w2) + z1 * x2;
    q[2] = ((w1 * z2 + x1 * y2) - y1 * x2) + z1 * w2;
    q[3] = ((w1 * w2 - x1 * x2) - y1 * y2) - z1 * z2;
  }
  for (int c = 0; c < 4; ++c) meas7[3 + c] = q[c];
  if (availability >= 1.0) return true;
  return uniform01(key(seed, target, tick, 6)) < availability;
}

#pragma clang fp contract(fast)

}  // namespace sg

#if defined(__HIPCC__)
// host entry points (stream_gen.hip); both throw on bad arguments / HIP errors
struct StreamSpec {
  int model;
  uint64_t seed;
  long first_target;
  double dt, availability, rpy_noise;
};
void stream_fill(const StreamSpec& sp, long n_targets, long first_tick, long n_ticks, bool f32, void* meas_dev, long tick_stride, long ld,
                 unsigned char* has_meas_dev, long has_stride, hipStream_t st);
void stream_truth(const StreamSpec& sp, long n_targets, double* pose0_dev, double* truth_dev, hipStream_t st);
#endif
}  // namespace te
