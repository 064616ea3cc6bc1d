// intersect_gate.hpp -- the convergence gate of IntersectionSolver::getIntersectionPoseWithSphere
// (src/intersection_solver.cpp:105-120), one gate per target.
//
// The reference holds, per solver object, two MovingAvgFilter (include/target_estimation/
// utils.hpp:206-265; window `filters_length`, default 250) over the position and angle distance
// between consecutive intersection poses, plus the previous pose; the intersection counts as
// converged when both filtered errors are below their thresholds.  The moving average needs the
// value that leaves the window, so the ring buffers are kept: 2 x W doubles per target in HBM
// (4 KB at W = 250), touched at one element per query.
#pragma once
#include <hip/hip_runtime.h>

#include "te_device_math.hpp"

namespace te {

struct GateArgs {
  const int* idx;            // slot of entry e (null: e)
  long n;
  int window;                // W
  const double* delta;       // [n] result of the intersection query (-1: none)
  const double* pose;        // [n][7] intersection pose
  double pos_th, ang_th;
  double* ring;              // [cap][2][W]
  double* sum;               // [cap][2]
  int* state;                // [cap][2]: idx | complete << 30
  double* prev;              // [cap][7]
  unsigned char* converged;  // [n]
  double* filt;              // [n][2] filtered errors, or null
  double* var;               // [n][2] MovingAvgFilter::getVariance() of the two filters, or null (O(W) per query)
};

// geometry.hpp:79-88
__device__ __forceinline__ double wrap_max_d(double x, double mx) { return ::fmod(mx + ::fmod(x, mx), mx); }
__device__ __forceinline__ double wrap_min_max_d(double x, double mn, double mx) { return mn + wrap_max_d(x - mn, mx - mn); }

// MovingAvgFilter::update, utils.hpp:222-251.  The variance it also computes (:241-246: over EVERY window entry,
// the still-unfilled zeros included, divided by the number of samples seen) is never read by the solver; it is
// evaluated only when the caller asks for it (var != null).
__device__ __forceinline__ double moving_avg_update(double* ring, double* sum, int* state, int W, double value, double* var = nullptr) {
  int idx = *state & 0x3fffffff;
  int complete = (*state >> 30) & 1;
  double s = *sum;
  s -= ring[idx];
  s += value;
  ring[idx] = value;
  if (!complete && idx == W - 1) complete = 1;
  const int num = complete ? W : idx + 1;
  const double res = s / num;
  idx = (idx + 1) % W;
  *sum = s;
  *state = idx | (complete << 30);
  if (var) {
    double vs = 0.0;
    for (int k = 0; k < W; ++k) { const double d = ring[k] - res; vs += d * d; }
    *var = vs / num;
  }
  return res;
}

__global__ void gate_kernel(const GateArgs a) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n) return;
  const long slot = a.idx ? (long)a.idx[e] : e;
  bool conv = false;
  double pf = 0.0, af = 0.0;
  if (a.delta[e] > -1) {                                         // intersection_solver.cpp:102
    const double* p = a.pose + e * 7;
    double* prev = a.prev + slot * 7;
    const double dx = p[0] - prev[0], dy = p[1] - prev[1], dz = p[2] - prev[2];
    const double pos_error = ::sqrt(dx * dx + dy * dy + dz * dz);  // :105
    double q1[4] = {p[3], p[4], p[5], p[6]}, q2[4] = {prev[3], prev[4], prev[5], prev[6]};
    quat_normalize(q1);                                           // :108-109
    quat_normalize(q2);
    // computeQuaternionError, geometry.hpp:630-651: q_e = q1 * q2^-1, normalised
    const double n2 = q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2] + q2[3] * q2[3];
    const double ix = -q2[0] / n2, iy = -q2[1] / n2, iz = -q2[2] / n2, iw = q2[3] / n2;
    double qe[4];
    qe[3] = q1[3] * iw - q1[0] * ix - q1[1] * iy - q1[2] * iz;
    qe[0] = q1[3] * ix + q1[0] * iw + q1[1] * iz - q1[2] * iy;
    qe[1] = q1[3] * iy + q1[1] * iw + q1[2] * ix - q1[0] * iz;
    qe[2] = q1[3] * iz + q1[2] * iw + q1[0] * iy - q1[1] * ix;
    quat_normalize(qe);
    const double pi = 3.14159265358979323846;
    const double ang_error = ::fabs(wrap_min_max_d(2 * ::acos(qe[3]), -pi, pi));   // :110
    pf = moving_avg_update(a.ring + (slot * 2 + 0) * a.window, a.sum + slot * 2 + 0, a.state + slot * 2 + 0, a.window, pos_error,
                           a.var ? a.var + e * 2 : nullptr);
    af = moving_avg_update(a.ring + (slot * 2 + 1) * a.window, a.sum + slot * 2 + 1, a.state + slot * 2 + 1, a.window, ang_error,
                           a.var ? a.var + e * 2 + 1 : nullptr);
    for (int c = 0; c < 7; ++c) prev[c] = p[c];                   // :117
    conv = (pf <= a.pos_th && af <= a.ang_th);                    // :119
  }
  a.converged[e] = conv ? 1 : 0;
  if (a.filt) { a.filt[e * 2] = pf; a.filt[e * 2 + 1] = af; }
}

__global__ void gate_reset_kernel(double* prev, long first, long count) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  double* p = prev + (first + k) * 7;
  p[0] = p[1] = p[2] = p[3] = p[4] = p[5] = 0.0;
  p[6] = 1.0;   // initPose(intersection_pose_prev_), intersection_solver.cpp:39
}

}  // namespace te
