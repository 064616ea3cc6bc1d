// launch_latency.hip -- what one host round trip to the GPU costs on this box: the floor under the one-target ABI
// (target_manager_update_meas + target_manager_get_est_pose = one flush: Batch::flush, batch_store.cpp).
//   hipcc --offload-arch=gfx950 -O2 tools/launch_latency.hip -o /tmp/launch_latency && /tmp/launch_latency
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void k_flag(volatile int* flag, volatile double* out, int seq) {
  if (threadIdx.x == 0) {
    out[0] = (double)seq;            // the "outputs" row, host-mapped
    __threadfence_system();
    *flag = seq;                     // then the sequence number the host spins on
  }
}

template <class F> double per_call_us(F f, int n) {
  for (int i = 0; i < 200; ++i) f(i);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f(1000 + i);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* h_flag; double* h_out; int* d_flag; double* d_out;
  CK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped));
  CK(hipHostMalloc((void**)&h_out, 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&d_flag, h_flag, 0));
  CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
  *h_flag = -1;
  const int N = 20000;
  std::printf("one empty kernel + hipStreamSynchronize          %7.2f us\n", per_call_us([&](int) { k_empty<<<1, 64, 0, s>>>(nullptr); CK(hipStreamSynchronize(s)); }, N));
  std::printf("two empty kernels + hipStreamSynchronize         %7.2f us\n", per_call_us([&](int) { k_empty<<<1, 64, 0, s>>>(nullptr); k_empty<<<1, 64, 0, s>>>(nullptr); CK(hipStreamSynchronize(s)); }, N));
  std::printf("one kernel, host spins on a mapped flag          %7.2f us\n", per_call_us([&](int i) {
    k_flag<<<1, 64, 0, s>>>(d_flag, d_out, i);
    while (*(volatile int*)h_flag != i) { }
  }, N));
  std::printf("two kernels, host spins on a mapped flag         %7.2f us\n", per_call_us([&](int i) {
    k_empty<<<1, 64, 0, s>>>(nullptr);
    k_flag<<<1, 64, 0, s>>>(d_flag, d_out, i);
    while (*(volatile int*)h_flag != i) { }
  }, N));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  std::printf("one kernel + event record + hipEventSynchronize  %7.2f us\n", per_call_us([&](int) { k_empty<<<1, 64, 0, s>>>(nullptr); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); }, N));
  std::printf("launch only (no wait), amortised                 %7.2f us\n", per_call_us([&](int) { k_empty<<<1, 64, 0, s>>>(nullptr); }, N));
  CK(hipStreamSynchronize(s));
  return 0;
}
