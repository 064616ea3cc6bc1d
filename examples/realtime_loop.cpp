// realtime_loop.cpp -- the reference's node loop (src/target_node.cpp:36-44, src/target_manager_ros.cpp:41-87: every tick take the new
// measurements, step every target, publish every estimated pose) WITHOUT a copy or a kernel launch per tick, from plain C++:
//   * the batch is resident on the GPU (target_batch_live_*): one launch serves tick after tick;
//   * the measurement ring lives in fine-grained DEVICE memory that the host writes through the PCIe BAR (large-BAR systems): the host
//     stores a tick's measurements where the resident kernel reads them -- no hipMemcpy, no staging;
//   * the estimated poses of every target land in HOST-mapped memory after every tick (target_batch_live_set_pose_output with the
//     device view of a hipHostMalloc block): the host reads them from its own memory;
//   * one doorbell store (also behind the BAR) per tick, one completion word back.
// Per tick the host does: a few stores, one spin on a word in its own memory, reads.  Prints the tick-to-poses time and checks the last
// poses against a second manager stepped by single launches on the same measurements (bit for bit).
//   hipcc --offload-arch=gfx950 -O2 -I include/target_estimation_amd examples/realtime_loop.cpp -o realtime_loop \
//         -L target_estimation_amd/lib -ltarget_estimation_amd -Wl,-rpath,$PWD/target_estimation_amd/lib
//   ./realtime_loop models/model_angular_velocities_params.yaml 40 2000
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "target_batch_c.h"
#include "target_manager_c.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
  const char* file = argc > 1 ? argv[1] : "models/model_angular_velocities_params.yaml";
  const long n = argc > 2 ? std::atol(argv[2]) : 40;
  const long steps = argc > 3 ? std::atol(argv[3]) : 2000;
  const long ring = 8;                        // entries of the measurement ring (the host is at most one tick ahead here)
  const double dt = 0.004;                    // the reference's 250 Hz

  target_manager_c* m = target_manager_new(file);
  target_manager_c* ref = target_manager_new(file);
  if (!m || !ref) return 3;
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> U(-5.0, 5.0);
  std::normal_distribution<double> noise(0.0, 0.01);
  std::vector<unsigned> ids((size_t)n);
  std::vector<double> p0((size_t)n * 7, 0.0), vel((size_t)n * 3), yaw_rate((size_t)n);
  for (long i = 0; i < n; ++i) {
    ids[(size_t)i] = (unsigned)(100 + i);
    for (int c = 0; c < 3; ++c) { p0[(size_t)i * 7 + c] = U(g); vel[(size_t)i * 3 + c] = 0.2 * U(g); }
    p0[(size_t)i * 7 + 6] = 1.0;
    yaw_rate[(size_t)i] = 0.3 * U(g);
  }
  if (target_manager_init_batch(m, ids.data(), n, dt, 0.0, p0.data(), nullptr, nullptr) != n) return 4;
  if (target_manager_init_batch(ref, ids.data(), n, dt, 0.0, p0.data(), nullptr, nullptr) != n) return 4;
  target_batch_c* b = target_manager_get_batch(m, 0);
  target_batch_c* rb = target_manager_get_batch(ref, 0);
  if (target_batch_live_capacity(b) < n) { std::fprintf(stderr, "batch too large for the resident mode\n"); return 5; }

  // the ring: SoA [ring][7][n] doubles, in device memory the host can write (or plain device memory + a copy per tick)
  int dev = 0, large_bar = 0;
  HIP_OK(hipGetDevice(&dev));
  HIP_OK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev));
  const size_t tick_words = (size_t)7 * (size_t)n;
  double* ring_dev = nullptr;
  bool through_bar = false;
  if (large_bar && hipExtMallocWithFlags((void**)&ring_dev, sizeof(double) * tick_words * ring, hipDeviceMallocFinegrained) == hipSuccess) through_bar = true;
  else { (void)hipGetLastError(); HIP_OK(hipMalloc((void**)&ring_dev, sizeof(double) * tick_words * ring)); }
  // the poses: SoA [7][n] doubles in host memory the GPU writes
  double* pose_host = nullptr; double* pose_dev_view = nullptr;
  HIP_OK(hipHostMalloc((void**)&pose_host, sizeof(double) * tick_words, hipHostMallocMapped | hipHostMallocCoherent));
  HIP_OK(hipHostGetDevicePointer((void**)&pose_dev_view, pose_host, 0));
  if (target_batch_live_set_pose_output(b, pose_dev_view, n) != 0) return 6;

  std::vector<double> meas(tick_words), staged(tick_words);
  auto measure = [&](long s) {                // what the sensors deliver at tick s: straight-line motion + a yaw rate, noisy positions
    const double t = (double)(s + 1) * dt;
    for (long i = 0; i < n; ++i) {
      for (int c = 0; c < 3; ++c) meas[(size_t)c * n + i] = p0[(size_t)i * 7 + c] + vel[(size_t)i * 3 + c] * t + noise(g);
      const double half = 0.5 * yaw_rate[(size_t)i] * t;
      meas[(size_t)3 * n + i] = 0.0; meas[(size_t)4 * n + i] = 0.0; meas[(size_t)5 * n + i] = std::sin(half); meas[(size_t)6 * n + i] = std::cos(half);
    }
  };
  double* tick_dev = nullptr;                 // the reference manager's copy of a tick
  HIP_OK(hipMalloc((void**)&tick_dev, sizeof(double) * tick_words));

  if (target_batch_live_start(b, dt, ring_dev, (long)tick_words, n, nullptr, 0, ring, 0, steps, 5.0) != 0) return 7;
  std::vector<double> us((size_t)steps);
  double checksum = 0;
  for (long s = 0; s < steps; ++s) {
    measure(s);
    HIP_OK(hipMemcpy(tick_dev, meas.data(), sizeof(double) * tick_words, hipMemcpyHostToDevice));     // (reference copy, outside the timed part)
    if (target_batch_step(rb, dt, tick_dev, n, nullptr) != 0) return 8;
    const auto t0 = std::chrono::steady_clock::now();
    double* entry = ring_dev + (size_t)(s % ring) * tick_words;
    if (through_bar) {
      std::memcpy(entry, meas.data(), sizeof(double) * tick_words);      // stores through the BAR (write-combined) ...
      __builtin_ia32_sfence();                                           // ... out before the doorbell
    } else {
      HIP_OK(hipMemcpy(entry, meas.data(), sizeof(double) * tick_words, hipMemcpyHostToDevice));
    }
    if (target_batch_live_post(b, 1) != 0) return 9;
    if (target_batch_live_wait(b, s + 1, 5.0) != 0) { std::fprintf(stderr, "tick %ld not served\n", s); return 10; }
    for (long i = 0; i < n; ++i) checksum += pose_host[i];               // the poses are in host memory now: "publish" them
    us[(size_t)s] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  std::vector<double> last(pose_host, pose_host + tick_words);
  if (target_batch_live_stop(b) != steps) return 11;
  std::sort(us.begin() + steps / 10, us.end());                           // (the first tenth is warm-up)
  const size_t k0 = (size_t)(steps / 10), kn = (size_t)steps - k0;
  std::printf("%ld targets, %ld ticks, measurements %s: measurements-to-poses %.2f us per tick (median; p10 %.2f, p99 %.2f)\n", n, steps,
              through_bar ? "stored through the PCIe BAR" : "copied with hipMemcpy (no large BAR)", us[k0 + kn / 2], us[k0 + kn / 10], us[k0 + (size_t)(kn * 0.99)]);
  int fails = 0;
  for (long i = 0; i < n; ++i) {
    double pose[7];
    if (!target_manager_get_est_pose(ref, ids[(size_t)i], pose)) { ++fails; continue; }
    for (int c = 0; c < 7; ++c) if (pose[c] != last[(size_t)c * n + i]) { ++fails; break; }
  }
  std::printf("%s (checksum %.6f)\n", fails ? "MISMATCH with single launches" : "realtime loop example ok", checksum);
  target_manager_delete(m); target_manager_delete(ref);
  (void)hipFree(ring_dev); (void)hipFree(tick_dev); (void)hipHostFree(pose_host);
  return fails ? 1 : 0;
}
