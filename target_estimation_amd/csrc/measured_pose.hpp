// measured_pose.hpp -- TargetInterface::measured_pose_ for the batched store (optional).
//
// The reference keeps the last measurement of every target (updateMeasurement: measured_pose_ = meas,
// src/target_interface.cpp:142-146; getter getMeasuredPose :117-121; rt_logger channel "measurement" :35) and starts
// it at initPose = [0 0 0 0 0 0 1] (:25, utils.hpp:64-67).  The filters never read it back, so the dense record does not
// carry it: a manager that was asked to (target_manager_set_keep_measurement) keeps one row of seven doubles per slot
// beside the records, written by a small kernel behind every step launch -- 56 B per measured target per tick, off by
// default so that the default tick moves the record and nothing else.
#pragma once
#include <hip/hip_runtime.h>

namespace te {

// rows [slot][7] <- the last measurement each entry had among the n_ticks ticks of the launch (entries without one keep
// their row).  meas: SoA, tick t at meas + t * tick_stride, component c at + c * ld; rows_valid = measurement rows the
// caller transported (3: x y z only -- the quaternion is then reported as the identity).
template <typename T>
__global__ void __launch_bounds__(256) keep_measurement_kernel(const T* __restrict__ meas, long ld, long tick_stride,
                                                               const unsigned char* __restrict__ has, long has_stride, int n_ticks,
                                                               const int* __restrict__ idx, long n, int rows_valid,
                                                               double* __restrict__ rows) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const long slot = idx ? (long)idx[e] : e;
  if (slot < 0) return;
  for (int t = n_ticks - 1; t >= 0; --t) {
    if (has && !has[(long)t * has_stride + e]) continue;
    const T* m = meas + (long)t * tick_stride + e;
#pragma unroll
    for (int c = 0; c < 7; ++c) rows[slot * 7 + c] = c < rows_valid ? (double)m[(long)c * ld] : (c == 6 ? 1.0 : 0.0);
    return;
  }
}

__global__ void __launch_bounds__(256) init_measured_rows_kernel(double* __restrict__ rows, long first, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count * 7) return;
  rows[first * 7 + i] = (i % 7 == 6) ? 1.0 : 0.0;   // initPose, utils.hpp:64-67
}

}  // namespace te
