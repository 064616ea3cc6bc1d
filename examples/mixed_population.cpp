// mixed_population.cpp -- BASELINE configs[3] / configs[4] from plain C++ (no Python, no torch): two motion models in ONE manager,
// every target of both stepped each tick by ONE launch (target_manager_population_tick), optionally with the own-time sphere
// intersection of every target fused into the same launch, replayed from a recorded hipGraph and timed with HIP events.
// What the call replaces in the reference: the caller's loop over every target every tick (src/target_manager.cpp:190-225) plus, for
// the query, IntersectionSolver::getIntersectionPoseWithSphere per target (src/intersection_solver.cpp:84-124).
//   hipcc --offload-arch=gfx950 -O2 -I include/target_estimation_amd examples/mixed_population.cpp -o mixed_population \
//         -L target_estimation_amd/lib -ltarget_estimation_amd -Wl,-rpath,$PWD/target_estimation_amd/lib
//   ./mixed_population models 62500 512 [query]        (angular_rates + angular_velocities; with `query`: angular_rates + uniform_acceleration)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "target_batch_c.h"
#include "target_manager_c.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

// Q, R, P0 of a shipped model file, through a throw-away manager that loads it (the reference's YAML format)
static bool model_matrices(const std::string& file, int* type, std::vector<double>& Q, std::vector<double>& R, std::vector<double>& P0) {
  target_manager_c* t = target_manager_new(file.c_str());
  if (!t) return false;
  double one[7] = {0, 0, 0, 0, 0, 0, 1};
  target_manager_init(t, 1u, 0.004, one, 0.0);
  const long n = target_manager_get_n(t, 1u), m = target_manager_get_m(t, 1u);
  *type = target_batch_type(target_manager_get_batch(t, 0));
  Q.assign((size_t)(n * n), 0.0); R.assign((size_t)(m * m), 0.0); P0.assign((size_t)(n * n), 0.0);
  const bool ok = n > 0 && target_manager_get_model_matrices(t, 1u, Q.data(), R.data(), P0.data());
  target_manager_delete(t);
  return ok;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "models";
  const long n = argc > 2 ? std::atol(argv[2]) : 62500;       // targets per model
  const long steps = argc > 3 ? std::atol(argv[3]) : 512;
  const bool query = argc > 4 && std::strcmp(argv[4], "query") == 0;
  const int ticks = 64;                                        // block of measurements kept in HBM and replayed
  const double dt = 0.004;
  const char* files[2] = {"model_angular_rates_params.yaml", query ? "model_uniform_acceleration_params.yaml" : "model_angular_velocities_params.yaml"};

  target_manager_c* m = target_manager_new_ex(nullptr, TARGET_DTYPE_F64, 0);   // no default model: every batch is created with its own matrices
  if (!m) return 3;
  hipStream_t stream;
  HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  target_manager_set_stream(m, stream);
  target_batch_sequence_c per_batch[2];
  std::memset(per_batch, 0, sizeof per_batch);
  std::vector<void*> to_free;
  for (int k = 0; k < 2; ++k) {
    int type = 0;
    std::vector<double> Q, R, P0;
    if (!model_matrices(dir + "/" + files[k], &type, Q, R, P0)) return 4;
    // the library's keyed stream generator: initial poses, then a ring of `ticks` ticks of measurements, all on the device
    target_stream_c spec;
    std::memset(&spec, 0, sizeof spec);
    spec.model = type; spec.seed = 20240004ull + 17ull * (unsigned)k; spec.first_target = 0; spec.dt = dt; spec.availability = 1.0;
    double* pose0_dev = nullptr;
    HIP_OK(hipMalloc((void**)&pose0_dev, sizeof(double) * 7 * n));
    if (target_stream_truth_dev(&spec, n, pose0_dev, nullptr, nullptr) != 0) return 5;
    std::vector<double> p0((size_t)n * 7);
    HIP_OK(hipMemcpy(p0.data(), pose0_dev, sizeof(double) * 7 * n, hipMemcpyDeviceToHost));
    HIP_OK(hipFree(pose0_dev));
    std::vector<unsigned> ids((size_t)n);
    for (long i = 0; i < n; ++i) ids[(size_t)i] = (unsigned)(i + k * n);
    if (target_manager_init_batch_typed(m, type, ids.data(), n, dt, 0.0, Q.data(), R.data(), P0.data(), 0, p0.data(), nullptr, nullptr) != n) return 6;
    double* meas = nullptr;                                    // SoA [ticks][7][n] doubles
    HIP_OK(hipMalloc((void**)&meas, sizeof(double) * 7 * n * ticks));
    to_free.push_back(meas);
    if (target_stream_fill_dev(&spec, n, 0, ticks, TARGET_DTYPE_F64, meas, 7 * n, n, nullptr, 0, stream) != 0) return 7;
    per_batch[k].meas_dev = meas; per_batch[k].tick_stride = 7 * n; per_batch[k].ld = n; per_batch[k].ring_ticks = ticks;
    if (query) {
      double *delta = nullptr, *pose = nullptr;
      HIP_OK(hipMalloc((void**)&delta, sizeof(double) * n));
      HIP_OK(hipMalloc((void**)&pose, sizeof(double) * 7 * n));
      to_free.push_back(delta); to_free.push_back(pose);
      per_batch[k].delta_dev = delta; per_batch[k].pose_dev = pose;
    }
  }
  HIP_OK(hipStreamSynchronize(stream));
  const double origin[3] = {0, 0, 0};
  const int one_launch = target_manager_population_tick(m);
  // record the graph of one block once (use_graph = 2), warm up, then time whole blocks
  if (target_manager_step_sequence_all(m, ticks, dt, per_batch, 2, query, origin, 1.0, 2) != 0) return 8;
  for (int w = 0; w < 2; ++w) target_manager_step_sequence_all(m, ticks, dt, per_batch, 2, query, origin, 1.0, 1);
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const long blocks = (steps + ticks - 1) / ticks;
  HIP_OK(hipEventRecord(e0, stream));
  for (long b = 0; b < blocks; ++b)
    if (target_manager_step_sequence_all(m, ticks, dt, per_batch, 2, query, origin, 1.0, 1) != 0) return 9;
  HIP_OK(hipEventRecord(e1, stream));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / (double)(blocks * ticks);
  double bytes = 0;
  for (int k = 0; k < 2; ++k) bytes += (double)target_batch_algorithmic_bytes(target_manager_get_batch(m, k)) * (double)n + (query ? 64.0 * (double)n : 0.0);
  long hits = 0;
  if (query) {
    std::vector<double> d((size_t)n);
    for (int k = 0; k < 2; ++k) {
      HIP_OK(hipMemcpy(d.data(), per_batch[k].delta_dev, sizeof(double) * n, hipMemcpyDeviceToHost));
      for (double v : d) hits += v > -1.0 ? 1 : 0;
    }
  }
  double pose[7];
  if (!target_manager_get_est_pose(m, 0, pose) || !target_manager_get_est_pose(m, (unsigned)n, pose)) return 10;
  std::printf("%ld + %ld targets (%s), %ld ticks, %s per tick: %.2f us per tick, %.3e predict+update cycles/s, %.0f GB/s algorithmic (%.1f %% of 8 TB/s)",
              n, n, query ? "angular_rates + uniform_acceleration, sphere query fused" : "angular_rates + angular_velocities", blocks * ticks,
              one_launch == 1 ? "ONE launch" : "one launch per batch", us, 2.0 * (double)n / (us * 1e-6), bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 8e12 * 100.0);
  if (query) std::printf("; %ld intersections at the last tick", hits);
  std::printf("; %d measurements per target\nmixed population example ok\n", target_manager_get_n_measurements(m, 0));
  for (void* p : to_free) (void)hipFree(p);
  target_manager_delete(m);
  (void)hipStreamDestroy(stream);
  return one_launch == 1 ? 0 : 11;
}
