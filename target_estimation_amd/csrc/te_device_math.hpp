// te_device_math.hpp -- device-side angle / rotation helpers of the Kalman hot path.
// Each function states the reference routine it reproduces (file:line under the reference
// tree); the arithmetic is re-derived for gfx950 (fma where the reference has mul+add).
#pragma once
#include <hip/hip_runtime.h>

namespace te {

template <typename T> struct Mth;
template <> struct Mth<double> {
  static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
  static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
  static __device__ __forceinline__ void sincos(double x, double* s, double* c) { ::sincos(x, s, c); }
  static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
  static __device__ __forceinline__ double asin(double x) { return ::asin(x); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double fmod(double x, double y) { return ::fmod(x, y); }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
  static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
};
template <> struct Mth<float> {
  static __device__ __forceinline__ float sin(float x) { return ::sinf(x); }
  static __device__ __forceinline__ float cos(float x) { return ::cosf(x); }
  static __device__ __forceinline__ void sincos(float x, float* s, float* c) { ::sincosf(x, s, c); }
  static __device__ __forceinline__ float atan2(float y, float x) { return ::atan2f(y, x); }
  static __device__ __forceinline__ float asin(float x) { return ::asinf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
  static __device__ __forceinline__ float fmod(float x, float y) { return ::fmodf(x, y); }
  static __device__ __forceinline__ float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
  static __device__ __forceinline__ float abs(float x) { return ::fabsf(x); }
};

// The geometry helpers below (and derive_outputs, kf_aux.hpp) are inlined into several kernels -- the step kernels, the
// outputs kernel, the fused getter-table epilogue of the indexed step.  Left to the compiler's default (contract = fast) the
// SAME source rounds differently from one inlining context to the next (an a*b+c fused here, not there: one ulp in a
// quaternion component), so they are compiled without contraction: one rounding sequence everywhere, and the one the
// oracle's strict build and the reference's x86-64 build (no FMA target) use.  The filter products keep their explicit fma.
#pragma clang fp contract(off)

template <typename T> __device__ __forceinline__ constexpr T pi_v() { return (T)3.14159265358979323846; }

// fmod(t, 2pi) for the reference's angle helpers.  IEEE fmod is exact, so wherever the quotient is
// known to be 0 or 1 the result can be formed without the (long, especially in fp64) library
// routine and is bit-identical: |t| < 2pi -> t;  2pi <= t < 4pi -> t - 2pi (exact by Sterbenz).
template <typename T> __device__ __forceinline__ T fmod_2pi(T t) {
  const T two_pi = 2 * pi_v<T>();
  if (t > -two_pi && t < two_pi) return t;
  if (t >= two_pi && t < 2 * two_pi) return t - two_pi;
  return Mth<T>::fmod(t, two_pi);
}

// geometry.hpp:31-36 constrainAngle
template <typename T> __device__ __forceinline__ T constrain_angle(T x) {
  const T pi = pi_v<T>();
  x = fmod_2pi(x + pi);
  if (x < 0) x += 2 * pi;
  return x - pi;
}
// geometry.hpp:43-45 angleConv
template <typename T> __device__ __forceinline__ T angle_conv(T a) { return fmod_2pi(constrain_angle(a)); }
// geometry.hpp:53-58 angleDiff
template <typename T> __device__ __forceinline__ T angle_diff(T a, T b) {
  const T pi = pi_v<T>();
  T d = fmod_2pi(b - a + pi);
  if (d < 0) d += 2 * pi;
  return d - pi;
}
// geometry.hpp:70-76 unwrap, one component
template <typename T> __device__ __forceinline__ T unwrap_angle(T prev, T now) { return prev - angle_diff(now, angle_conv(prev)); }

// Eigen::Quaterniond::normalize(), q = [x y z w]
template <typename T> __device__ __forceinline__ void quat_normalize(T* q) {
  T nrm = Mth<T>::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= nrm; q[1] /= nrm; q[2] /= nrm; q[3] /= nrm;
}

// geometry.hpp:154-176 quatToRpy (gimbal branches at |sin pitch| > 0.9999)
// SERIAL: the three inverse trigonometric functions one after the other (scheduling barriers between them) instead of interleaved:
// same values, a third of the temporaries -- for kernels whose capacity is their register count (the resident fp64 kernels).
template <typename T, bool SERIAL = false> __device__ __forceinline__ void quat_to_rpy(const T* q, T* rpy) {
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T sp = -2 * (x * z - w * y);
  if (sp > (T)0.9999) {
    rpy[0] = 0; rpy[1] = pi_v<T>() / 2; rpy[2] = 2 * Mth<T>::atan2(z, w);
  } else if (sp < (T)-0.9999) {
    rpy[0] = 0; rpy[1] = -pi_v<T>() / 2; rpy[2] = 2 * Mth<T>::atan2(z, w);
  } else {
    rpy[0] = Mth<T>::atan2(2 * (y * z + w * x), (w * w - x * x - y * y + z * z));
    if constexpr (SERIAL) __builtin_amdgcn_sched_barrier(0);
    rpy[1] = Mth<T>::asin(sp);
    if constexpr (SERIAL) __builtin_amdgcn_sched_barrier(0);
    rpy[2] = Mth<T>::atan2(2 * (x * y + w * z), (w * w + x * x - y * y - z * z));
  }
}

// geometry.hpp:178-189 rpyToQuat
template <typename T> __device__ __forceinline__ void rpy_to_quat(const T* rpy, T* q) {
  T sph, cph, sth, cth, sps, cps;
  Mth<T>::sincos(rpy[0] / 2, &sph, &cph);
  Mth<T>::sincos(rpy[1] / 2, &sth, &cth);
  Mth<T>::sincos(rpy[2] / 2, &sps, &cps);
  q[3] = cph * cth * cps + sph * sth * sps;
  q[0] = sph * cth * cps - cph * sth * sps;
  q[1] = cph * sth * cps + sph * cth * sps;
  q[2] = cph * cth * sps - sph * sth * cps;
  quat_normalize(q);
}

// Eigen::Quaterniond::toRotationMatrix() (used at angular_rates.cpp:127, angular_velocities.cpp:163)
template <typename T> __device__ __forceinline__ void quat_to_rot(const T* q, T* R) {
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const T twx = tx * w, twy = ty * w, twz = tz * w;
  const T txx = tx * x, txy = ty * x, txz = tz * x;
  const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen rotation-matrix -> quaternion (isometryToPose7d, geometry.hpp:590-594)
template <typename T> __device__ __forceinline__ void rot_to_quat(const T* R, T* q) {
  T t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = Mth<T>::sqrt(t + 1);
    q[3] = (T)0.5 * t;
    t = (T)0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    // largest-diagonal branch, written without runtime-indexed arrays
    if (R[0] >= R[4] && R[0] >= R[8]) {          // i = 0, j = 1, k = 2
      t = Mth<T>::sqrt(R[0] - R[4] - R[8] + 1);
      q[0] = (T)0.5 * t; t = (T)0.5 / t;
      q[3] = (R[7] - R[5]) * t; q[1] = (R[3] + R[1]) * t; q[2] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {    // i = 1, j = 2, k = 0
      t = Mth<T>::sqrt(R[4] - R[8] - R[0] + 1);
      q[1] = (T)0.5 * t; t = (T)0.5 / t;
      q[3] = (R[2] - R[6]) * t; q[2] = (R[7] + R[5]) * t; q[0] = (R[1] + R[3]) * t;
    } else {                                      // i = 2, j = 0, k = 1
      t = Mth<T>::sqrt(R[8] - R[0] - R[4] + 1);
      q[2] = (T)0.5 * t; t = (T)0.5 / t;
      q[3] = (R[3] - R[1]) * t; q[0] = (R[2] + R[6]) * t; q[1] = (R[5] + R[7]) * t;
    }
  }
}

// geometry.hpp:191-196 rotToRpy
template <typename T> __device__ __forceinline__ void rot_to_rpy(const T* R, T* rpy) {
  rpy[0] = Mth<T>::atan2(R[7], R[8]);
  rpy[1] = Mth<T>::atan2(-R[6], Mth<T>::sqrt(R[7] * R[7] + R[8] * R[8]));
  rpy[2] = Mth<T>::atan2(R[3], R[0]);
}

// geometry.hpp:448-465 omegaToMatrix + :493-504 Qtran, applied to a quaternion: q' = Qtran(dt,w) q
template <typename T> __device__ __forceinline__ void qtran_apply(T dt, const T* w, const T* q, T* out) {
  const T nrm = Mth<T>::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (nrm > 0) {
    const T tmp = nrm * dt / (T)2;
    const T c = Mth<T>::cos(tmp), s = (T)2 / nrm * Mth<T>::sin(tmp);
    const T hx = (T)0.5 * w[0], hy = (T)0.5 * w[1], hz = (T)0.5 * w[2];
    // rows of cos*I + s*S, S = 0.5*[[0,-wz,wy,wx],[wz,0,-wx,wy],[-wy,wx,0,wz],[-wx,-wy,-wz,0]]
    out[0] = (c * q[0] + (s * -hz) * q[1]) + (s * hy) * q[2] + (s * hx) * q[3];
    out[1] = ((s * hz) * q[0] + c * q[1]) + (s * -hx) * q[2] + (s * hy) * q[3];
    out[2] = ((s * -hy) * q[0] + (s * hx) * q[1]) + c * q[2] + (s * hz) * q[3];
    out[3] = ((s * -hx) * q[0] + (s * -hy) * q[1]) + (s * -hz) * q[2] + c * q[3];
  } else {
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  }
}

#pragma clang fp contract(fast)   // back to hipcc's default for what follows

}  // namespace te
