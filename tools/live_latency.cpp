// live_latency.cpp -- the resident mode's round trip from plain C++ (no Python in the loop): one doorbell, spin until the
// relay's done word shows the tick, next doorbell.  Prints mean / median / min microseconds per paced tick and the
// back-to-back figure for the same batch.
//   hipcc --offload-arch=gfx950 -O2 -I include/target_estimation_amd tools/live_latency.cpp -o live_latency -L target_estimation_amd/lib -ltarget_estimation_amd -Wl,-rpath,$PWD/target_estimation_amd/lib
//   ./live_latency models/model_uniform_acceleration_params.yaml 100000 f32
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "target_batch_c.h"
#include "target_manager_c.h"

int main(int argc, char** argv) {
  const char* file = argc > 1 ? argv[1] : "models/model_uniform_velocity_params.yaml";
  const long n = argc > 2 ? std::atol(argv[2]) : 10000;
  const bool f32 = argc > 3 && std::strcmp(argv[3], "f32") == 0;
  const long ring = 64, paced = 3000, burst = 20000;
  const double dt = 0.004;
  target_manager_c* m = target_manager_new_ex(file, f32 ? TARGET_DTYPE_F32 : TARGET_DTYPE_F64, 0);
  if (!m) return 3;
  std::vector<unsigned> ids((size_t)n);
  std::vector<double> p0((size_t)n * 7, 0.0);
  for (long i = 0; i < n; ++i) { ids[(size_t)i] = (unsigned)i; p0[(size_t)i * 7] = 1e-3 * (double)i; p0[(size_t)i * 7 + 6] = 1.0; }
  if (target_manager_init_batch(m, ids.data(), n, dt, 0.0, p0.data(), nullptr, nullptr) != n) return 4;
  target_batch_c* b = target_manager_get_batch(m, 0);
  target_stream_c spec;
  std::memset(&spec, 0, sizeof spec);
  spec.model = target_batch_type(b); spec.seed = 5; spec.dt = dt; spec.availability = 1.0;
  const size_t es = f32 ? 4 : 8;
  void* ring_dev = nullptr;
  if (hipMalloc(&ring_dev, es * 7 * n * ring) != hipSuccess) return 5;
  if (target_stream_fill_dev(&spec, n, 0, ring, f32 ? TARGET_DTYPE_F32 : TARGET_DTYPE_F64, ring_dev, 7 * n, n, nullptr, 0, nullptr) != 0) return 6;
  (void)hipDeviceSynchronize();
  if (target_batch_live_start(b, dt, ring_dev, 7 * n, n, nullptr, 0, ring, 0, 1L << 30, 5.0) != 0) return 7;
  long posted = 0;
  auto tick = [&]() { target_batch_live_post(b, 1); ++posted; return target_batch_live_wait(b, posted, 5.0) == 0; };
  for (int i = 0; i < 200; ++i) if (!tick()) return 8;
  std::vector<double> us((size_t)paced);
  for (long i = 0; i < paced; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!tick()) return 9;
    us[(size_t)i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  std::sort(us.begin(), us.end());
  double mean = 0;
  for (double v : us) mean += v;
  const auto t0 = std::chrono::steady_clock::now();
  target_batch_live_post_each(b, burst);
  posted += burst;
  if (target_batch_live_wait(b, posted, 20.0) != 0) return 10;
  const double b2b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / burst;
  const auto t1 = std::chrono::steady_clock::now();
  target_batch_live_post(b, burst);           // the same number of ticks behind ONE doorbell: the device alone
  posted += burst;
  if (target_batch_live_wait(b, posted, 20.0) != 0) return 11;
  const double one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / burst;
  const long served = target_batch_live_stop(b);
  std::printf("%ld targets %s: paced mean %.2f us, median %.2f, min %.2f, p99 %.2f; back to back %.2f us per tick; one doorbell for %ld ticks %.2f us per tick; %ld ticks served\n", n,
              f32 ? "f32" : "f64", mean / paced, us[(size_t)paced / 2], us[0], us[(size_t)(paced * 0.99)], b2b, burst, one, served);
  target_manager_delete(m);
  (void)hipFree(ring_dev);
  return served == posted ? 0 : 1;
}
