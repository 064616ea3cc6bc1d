// batched_replay.cpp -- the device-resident batch API from plain C++ (no Python, no torch): N targets of one
// model file, a block of synthetic measurements in HBM, one step launch per tick replayed from a hipGraph,
// timed with HIP events.  The same loop bench.py times for `value`.
//   hipcc --offload-arch=gfx950 -O2 -I include/target_estimation_amd examples/batched_replay.cpp -o batched_replay \
//         -L target_estimation_amd/lib -ltarget_estimation_amd -Wl,-rpath,$PWD/target_estimation_amd/lib
//   ./batched_replay models/model_uniform_velocity_params.yaml 10000 2000
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "target_batch_c.h"
#include "target_manager_c.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
  const char* file = argc > 1 ? argv[1] : "models/model_uniform_velocity_params.yaml";
  const long n = argc > 2 ? std::atol(argv[2]) : 10000;
  const long steps = argc > 3 ? std::atol(argv[3]) : 2000;
  const int ticks = 64;                      // block of measurements kept in HBM and replayed
  const double dt = 0.004;

  target_manager_c* m = target_manager_new(file);          // fp64, automatic layout
  if (!m) return 3;
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(-10.0, 10.0);
  std::normal_distribution<double> noise(0.0, 0.01);
  std::vector<unsigned> ids((size_t)n);
  std::vector<double> p0((size_t)n * 7, 0.0);
  for (long i = 0; i < n; ++i) {
    ids[(size_t)i] = (unsigned)i;
    for (int c = 0; c < 3; ++c) p0[(size_t)i * 7 + c] = U(g);
    p0[(size_t)i * 7 + 6] = 1.0;
  }
  if (target_manager_init_batch(m, ids.data(), n, dt, 0.0, p0.data(), nullptr, nullptr) != n) return 4;
  target_batch_c* b = target_manager_get_batch(m, 0);

  // measurements: SoA [ticks][7][n] doubles (the batch precision), straight-line motion + noise
  std::vector<double> meas((size_t)ticks * 7 * (size_t)n);
  for (int s = 0; s < ticks; ++s)
    for (long i = 0; i < n; ++i) {
      for (int c = 0; c < 3; ++c) meas[((size_t)s * 7 + c) * n + i] = p0[(size_t)i * 7 + c] + 0.5 * (s + 1) * dt + noise(g);
      for (int c = 3; c < 6; ++c) meas[((size_t)s * 7 + c) * n + i] = 0.0;
      meas[((size_t)s * 7 + 6) * n + i] = 1.0;
    }
  double* d_meas = nullptr;
  HIP_OK(hipMalloc((void**)&d_meas, meas.size() * sizeof(double)));
  HIP_OK(hipMemcpy(d_meas, meas.data(), meas.size() * sizeof(double), hipMemcpyHostToDevice));

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  target_manager_set_stream(m, stream);
  // record the 64-launch graph once, warm up, then time whole blocks
  if (target_batch_step_sequence(b, ticks, dt, d_meas, 7 * n, n, nullptr, 0, 2) != 0) return 5;
  for (int w = 0; w < 4; ++w) target_batch_step_sequence(b, ticks, dt, d_meas, 7 * n, n, nullptr, 0, 1);
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const long blocks = (steps + ticks - 1) / ticks;
  HIP_OK(hipEventRecord(e0, stream));
  for (long k = 0; k < blocks; ++k)
    if (target_batch_step_sequence(b, ticks, dt, d_meas, 7 * n, n, nullptr, 0, 1) != 0) return 6;
  HIP_OK(hipEventRecord(e1, stream));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  const double per_tick_us = ms * 1e3 / (double)(blocks * ticks);
  const double bytes = (double)target_batch_algorithmic_bytes(b) * (double)n;

  double pose[7];
  if (!target_manager_get_est_pose(m, 0, pose)) return 7;
  std::printf("%ld targets, %ld ticks: %.2f us per tick, %.3e predict+update cycles/s, %.0f GB/s algorithmic (%.1f %% of 8 TB/s); "
              "target 0 at (%.3f, %.3f, %.3f), %d measurements\n",
              n, blocks * ticks, per_tick_us, (double)n / (per_tick_us * 1e-6), bytes / (per_tick_us * 1e-6) / 1e9,
              bytes / (per_tick_us * 1e-6) / 8e12 * 100.0, pose[0], pose[1], pose[2], target_manager_get_n_measurements(m, 0));
  HIP_OK(hipFree(d_meas));
  target_manager_delete(m);
  return 0;
}
