// stream_gen.hip -- kernels and C entry points of the synthetic stream generator (stream_gen.hpp).
// One thread per (tick, target): seven measurement words into the SoA ring [tick][7][ld] in the batch precision, the
// availability byte into [tick][n].  Set-up work, not on the per-tick path: bench.py and the tests fill their rings with it
// before any timed region (and no longer launch torch kernels for that).
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>

#include "hip_check.hpp"
#include "stream_gen.hpp"

namespace te {
namespace {

template <class T>
__global__ void __launch_bounds__(256) stream_fill_kernel(int model, uint64_t seed, long first_target, long n, long first_tick,
                                                          long n_ticks, double dt, double availability, double rpy_noise,
                                                          T* __restrict__ meas, long tick_stride, long ld,
                                                          unsigned char* __restrict__ has, long has_stride) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long s = blockIdx.y;
  if (i >= n || s >= n_ticks) return;
  const uint64_t target = (uint64_t)(first_target + i);
  const sg::Truth tr = sg::truth_of(model, seed, target);
  double m7[7];
  const bool got = sg::measurement(tr, seed, target, (uint32_t)(first_tick + s), dt, availability, rpy_noise, m7);
  T* row = meas + s * tick_stride + i;
#pragma unroll
  for (int c = 0; c < 7; ++c) row[c * ld] = (T)m7[c];
  if (has) has[s * has_stride + i] = got ? 1 : 0;
}

__global__ void __launch_bounds__(256) stream_truth_kernel(int model, uint64_t seed, long first_target, long n,
                                                           double* __restrict__ pose0, double* __restrict__ truth) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t target = (uint64_t)(first_target + i);
  const sg::Truth tr = sg::truth_of(model, seed, target);
  if (pose0) sg::init_pose(tr, seed, target, pose0 + 7 * i);
  if (truth) {
    for (int c = 0; c < 3; ++c) {
      truth[12 * i + c] = tr.p[c];
      truth[12 * i + 3 + c] = tr.v[c];
      truth[12 * i + 6 + c] = tr.a[c];
      truth[12 * i + 9 + c] = tr.w[c];
    }
  }
}

}  // namespace

void stream_fill(const StreamSpec& sp, long n_targets, long first_tick, long n_ticks, bool f32, void* meas_dev, long tick_stride,
                 long ld, unsigned char* has_meas_dev, long has_stride, hipStream_t st) {
  if (!meas_dev) throw std::invalid_argument("NULL measurement buffer");
  if (sp.model < 0 || sp.model > 3) throw std::invalid_argument("unknown model");
  if (n_targets < 0 || n_ticks < 0 || first_tick < 0 || first_tick + n_ticks > 0xFFFFFFFEl) throw std::invalid_argument("bad tick / target range");
  if (ld < n_targets || tick_stride < 7 * ld || (has_meas_dev && has_stride < n_targets))
    throw std::invalid_argument("strides shorter than the rows");
  if (n_targets == 0 || n_ticks == 0) return;
  const long per = 65535;   // gridDim.y limit: ticks in slabs
  for (long t0 = 0; t0 < n_ticks; t0 += per) {
    const long nt = n_ticks - t0 < per ? n_ticks - t0 : per;
    dim3 grid((unsigned)((n_targets + 255) / 256), (unsigned)nt);
    unsigned char* hp = has_meas_dev ? has_meas_dev + t0 * has_stride : nullptr;
    if (!f32)
      stream_fill_kernel<double><<<grid, 256, 0, st>>>(sp.model, sp.seed, sp.first_target, n_targets, first_tick + t0, nt, sp.dt, sp.availability,
                                                        sp.rpy_noise, (double*)meas_dev + t0 * tick_stride, tick_stride, ld, hp, has_stride);
    else
      stream_fill_kernel<float><<<grid, 256, 0, st>>>(sp.model, sp.seed, sp.first_target, n_targets, first_tick + t0, nt, sp.dt, sp.availability,
                                                       sp.rpy_noise, (float*)meas_dev + t0 * tick_stride, tick_stride, ld, hp, has_stride);
    TE_HIP_CHECK(hipGetLastError());
  }
}

void stream_truth(const StreamSpec& sp, long n_targets, double* pose0_dev, double* truth_dev, hipStream_t st) {
  if (sp.model < 0 || sp.model > 3) throw std::invalid_argument("unknown model");
  if (n_targets <= 0 || (!pose0_dev && !truth_dev)) return;
  stream_truth_kernel<<<(unsigned)((n_targets + 255) / 256), 256, 0, st>>>(sp.model, sp.seed, sp.first_target, n_targets, pose0_dev, truth_dev);
  TE_HIP_CHECK(hipGetLastError());
}

}  // namespace te
