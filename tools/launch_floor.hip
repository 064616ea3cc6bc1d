// launch_floor.hip -- how short can one dependent tick be?  Back-to-back launches on one stream of
// (a) an empty kernel, (b) a read-modify-write of a 7.3 MB buffer (the configs[1] working set).
// build: hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void rmw_kernel(double2* buf, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { double2 v = buf[i]; v.x = v.x * 1.0000001 + 1e-9; v.y = v.y * 0.9999999; buf[i] = v; }
}

template <class F> float time_loop(F f, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 50; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const long bytes = 10000L * 42 * 8;  // 10k UV fp64 records
  const long n = bytes / 16;
  double2* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  for (int blocks : {1, 120, 480, 2048}) {
    float us = time_loop([&] { hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, 0, (int*)nullptr); }, 2000);
    printf("empty kernel, %4d blocks x 256: %.2f us per launch\n", blocks, us);
  }
  for (int tpb : {64, 256, 1024}) {
    int blocks = (int)((n + tpb - 1) / tpb);
    float us = time_loop([&] { hipLaunchKernelGGL(rmw_kernel, dim3(blocks), dim3(tpb), 0, 0, buf, n); }, 2000);
    printf("rmw 3.36 MB in place, %5d blocks x %4d: %.2f us per launch (%.0f GB/s r+w)\n", blocks, tpb, us, 2.0 * bytes / us * 1e-3);
  }
  // graph of 32 dependent rmw launches
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(rmw_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, buf, n);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < 100; ++i) hipGraphLaunch(ge, s);
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("graph of 32 rmw launches: %.2f us per launch\n", ms * 1e3f / 3200);
  return 0;
}
