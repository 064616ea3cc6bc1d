// bar_doorbell.hip -- can the host ring a doorbell that lives in DEVICE memory (written through the PCIe BAR) instead of one in
// host memory that the GPU has to read over PCIe?  One resident wavefront polls the word and answers in a host-mapped word; the host
// measures the round trip for both placements.  (Experiment for the resident mode's relay: csrc/kf_step.hpp live_relay.)
//   hipcc --offload-arch=gfx950 -O2 tools/bar_doorbell.hip -o tools/_build/bar_doorbell && tools/_build/bar_doorbell
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

__global__ void echo_kernel(const long long* bell, long long* answer, long long rounds) {
  if (threadIdx.x != 0) return;
  long long seen = 0;
  for (long long spins = 0; spins < 2000000000LL; ++spins) {
    const long long v = __hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v != seen) {
      seen = v;
      __hip_atomic_store(answer, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v >= rounds) return;
    }
  }
}

static double run(long long* bell_host_view, const long long* bell_dev_view, long long* ans_h, long long* ans_d, long long rounds, const char* what) {
  __atomic_store_n(bell_host_view, 0LL, __ATOMIC_SEQ_CST);
  __atomic_store_n(ans_h, 0LL, __ATOMIC_SEQ_CST);
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, s, bell_dev_view, ans_d, rounds);
  std::vector<double> us;
  for (long long k = 1; k <= rounds; ++k) {
    const auto t0 = std::chrono::steady_clock::now();
    __atomic_store_n(bell_host_view, k, __ATOMIC_RELEASE);
    __builtin_ia32_sfence();
    while (__atomic_load_n(ans_h, __ATOMIC_ACQUIRE) != k) __builtin_ia32_pause();
    us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  CHECK(hipStreamSynchronize(s));
  CHECK(hipStreamDestroy(s));
  std::sort(us.begin(), us.end());
  printf("%-44s round trip: median %.2f us, p10 %.2f, p90 %.2f (%lld rounds)\n", what, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], rounds);
  return us[us.size() / 2];
}

int main() {
  int dev = 0, large_bar = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev));
  printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
  long long *h = nullptr, *d = nullptr;
  CHECK(hipHostMalloc((void**)&h, 256, hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostGetDevicePointer((void**)&d, h, 0));
  const long long rounds = 2000;
  run(h, d, h + 16, d + 16, rounds, "doorbell in host memory (GPU reads over PCIe)");
  if (large_bar) {
    long long* fb = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&fb, 256, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { printf("fine-grained device memory: %s\n", hipGetErrorString(e)); return 0; }
    hipPointerAttribute_t at;
    CHECK(hipPointerGetAttributes(&at, fb));
    printf("fine-grained device allocation: device pointer %p, host pointer %p\n", at.devicePointer, at.hostPointer);
    // with a large BAR the runtime maps device memory into the process: the device pointer is valid on the host too
    run(fb, fb, h + 16, d + 16, rounds, "doorbell in device memory (host writes through the BAR)");
    CHECK(hipFree(fb));
  }
  CHECK(hipHostFree(h));
  return 0;
}
