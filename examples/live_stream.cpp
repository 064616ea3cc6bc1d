// live_stream.cpp -- a live stream without a launch per tick, from plain C++ (no Python, no torch): the resident mode of the
// batch API (target_batch_live_*), the library's keyed stream generator filling the ring behind the running session on a
// second stream, one doorbell per tick, and at the end the round-3 getters of the plugin surface on a few targets.
//   hipcc --offload-arch=gfx950 -O2 -I include/target_estimation_amd examples/live_stream.cpp -o live_stream \
//         -L target_estimation_amd/lib -ltarget_estimation_amd -Wl,-rpath,$PWD/target_estimation_amd/lib
//   ./live_stream models/model_uniform_velocity_params.yaml 10000 4000
// Prints the ticks served, microseconds per tick (doorbells posted back to back) and checks the result against the same
// ticks stepped one launch at a time on a second manager (bit for bit).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "target_batch_c.h"
#include "target_manager_c.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
  const char* file = argc > 1 ? argv[1] : "models/model_uniform_velocity_params.yaml";
  const long n = argc > 2 ? std::atol(argv[2]) : 10000;
  const long steps = argc > 3 ? std::atol(argv[3]) : 4000;
  const long ring = 64;                       // ticks the ring holds; refilled half by half while the session runs
  const double dt = 0.004;

  target_manager_c* m = target_manager_new(file);      // fp64, automatic layout (axis-separable for the shipped files)
  target_manager_c* ref = target_manager_new(file);
  if (!m || !ref) return 3;
  target_manager_set_keep_measurement(ref, 1);         // TargetInterface::getMeasuredPose on the reference copy
  // the stream: what the reference's test generator produces, per target (csrc/stream_gen.hpp)
  target_stream_c spec;
  std::memset(&spec, 0, sizeof spec);
  spec.seed = 20240002ull; spec.first_target = 0; spec.dt = dt; spec.availability = 1.0; spec.rpy_noise = 0.0;
  double* pose0_dev = nullptr;
  HIP_OK(hipMalloc((void**)&pose0_dev, sizeof(double) * 7 * n));
  std::vector<unsigned> ids((size_t)n);
  for (long i = 0; i < n; ++i) ids[(size_t)i] = (unsigned)i;
  std::vector<double> p0((size_t)n * 7);
  // the model of the file decides the stream's motion model: create one target to ask, then the rest
  {
    double one[7] = {0, 0, 0, 0, 0, 0, 1};
    target_manager_init(m, 0xFFFFFFFFu, dt, one, 0.0);
    spec.model = target_batch_type(target_manager_get_batch(m, 0));
    target_manager_erase(m, 0xFFFFFFFFu);
  }
  if (target_stream_truth_dev(&spec, n, pose0_dev, nullptr, nullptr) != 0) return 4;
  HIP_OK(hipMemcpy(p0.data(), pose0_dev, sizeof(double) * 7 * n, hipMemcpyDeviceToHost));
  if (target_manager_init_batch(m, ids.data(), n, dt, 0.0, p0.data(), nullptr, nullptr) != n) return 5;
  if (target_manager_init_batch(ref, ids.data(), n, dt, 0.0, p0.data(), nullptr, nullptr) != n) return 5;
  target_batch_c* b = target_manager_get_batch(m, 0);
  target_batch_c* rb = target_manager_get_batch(ref, 0);
  if (target_batch_live_capacity(b) < n) { std::fprintf(stderr, "batch too large for the resident mode (%ld targets fit)\n", target_batch_live_capacity(b)); return 6; }

  double* ring_dev = nullptr;                 // SoA [ring][7][n] doubles
  HIP_OK(hipMalloc((void**)&ring_dev, sizeof(double) * 7 * n * ring));
  hipStream_t fill;
  HIP_OK(hipStreamCreateWithFlags(&fill, hipStreamNonBlocking));
  auto fill_ticks = [&](long first_tick, long count) {      // ticks first_tick.. into ring entries first_tick % ring ..
    return target_stream_fill_dev(&spec, n, first_tick, count, TARGET_DTYPE_F64, ring_dev + (first_tick % ring) * 7 * n, 7 * n, n, nullptr, 0, fill);
  };
  if (fill_ticks(0, ring) != 0) return 7;
  HIP_OK(hipStreamSynchronize(fill));

  if (target_batch_live_start(b, dt, ring_dev, 7 * n, n, nullptr, 0, ring, 0, steps, 5.0) != 0) return 8;
  const auto t0 = std::chrono::steady_clock::now();
  long posted = 0;
  while (posted < steps) {
    const long k = std::min<long>(ring / 2, steps - posted);
    if (target_batch_live_post_each(b, k) != 0) return 9;          // k doorbells of one tick each
    posted += k;
    // the half of the ring just posted must be consumed before it is overwritten: wait for it, then refill it with the
    // ticks that come a full ring later (on the second stream, while the session goes on with the other half)
    if (target_batch_live_wait(b, posted, 5.0) != 0) { std::fprintf(stderr, "session stalled at %ld of %ld\n", target_batch_live_done(b), posted); return 10; }
    const long next_first = posted + ring / 2;
    if (next_first < steps) {
      if (fill_ticks(next_first, std::min<long>(ring / 2, steps - next_first)) != 0) return 11;
      HIP_OK(hipStreamSynchronize(fill));
    }
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const long served = target_batch_live_stop(b);
  std::printf("%ld targets, %ld ticks served by one resident launch, %.2f us per tick (incl. the ring refills)\n", n, served, secs / steps * 1e6);
  if (served != steps) return 12;

  // the same ticks, one launch each, on the second manager
  double* tick_dev = nullptr;
  HIP_OK(hipMalloc((void**)&tick_dev, sizeof(double) * 7 * n));
  for (long s = 0; s < steps; ++s) {
    if (target_stream_fill_dev(&spec, n, s, 1, TARGET_DTYPE_F64, tick_dev, 7 * n, n, nullptr, 0, nullptr) != 0) return 13;
    if (target_batch_step(rb, dt, tick_dev, n, nullptr) != 0) return 14;
  }
  int fails = 0;
  const unsigned probe[4] = {0u, 1u, (unsigned)(n / 2), (unsigned)(n - 1)};
  for (unsigned id : probe) {
    double xa[18], Pa[18 * 18], xb[18], Pb[18 * 18];
    const long ns = target_manager_get_state_batch(m, &id, 1, xa, Pa);
    if (target_manager_get_state_batch(ref, &id, 1, xb, Pb) != ns || ns <= 0) { ++fails; continue; }
    if (std::memcmp(xa, xb, sizeof(double) * ns) != 0 || std::memcmp(Pa, Pb, sizeof(double) * ns * ns) != 0) { std::printf("target %u differs\n", id); ++fails; }
    double T[16], per = 0, Q[18 * 18], R[36], P0[18 * 18], mp[7], pose[7];
    if (!target_manager_get_estimated_transform(m, id, T) || !target_manager_get_est_pose(m, id, pose)) ++fails;
    if (T[3] != pose[0] || T[7] != pose[1] || T[11] != pose[2] || T[15] != 1.0) ++fails;
    if (!target_manager_get_period_estimate(m, id, &per)) ++fails;
    if (target_manager_get_n(m, id) != ns || target_manager_get_m(m, id) <= 0) ++fails;
    if (!target_manager_get_model_matrices(m, id, Q, R, P0) || !(Q[0] > 0) || !(R[0] > 0) || !(P0[0] > 0)) ++fails;
    if (target_manager_get_measured_pose(m, id, mp)) ++fails;                     // not kept on this manager
    if (!target_manager_get_measured_pose(ref, id, mp) || !std::isfinite(mp[0]) || mp[6] == 0.0) ++fails;   // kept on the other one
  }
  std::printf("%s\n", fails ? "MISMATCH" : "live stream example ok");
  target_manager_delete(m);
  target_manager_delete(ref);
  (void)hipFree(ring_dev); (void)hipFree(tick_dev); (void)hipFree(pose0_dev);
  (void)hipStreamDestroy(fill);
  return fails ? 1 : 0;
}
