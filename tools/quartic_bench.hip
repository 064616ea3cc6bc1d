// quartic_bench.hip -- microbenchmark of the device quartic solver (te_quartic.hpp) on synthetic
// sphere-intersection coefficients.  Build: hipcc --offload-arch=gfx950 -O3 -I target_estimation_amd/csrc
//   tools/quartic_bench.hip -o /tmp/quartic_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "te_quartic.hpp"
#include "quartic_bracketing.hpp"   // the earlier solver, for A/B

__global__ void solve_kernel(const double* c, long n, double* out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double k[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) k[i] = c[i * n + e];
  out[e] = te::first_crossing_quartic(k);
}
__global__ void solve_bracketing_kernel(const double* c, long n, double* out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double k[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) k[i] = c[i * n + e];
  const double d = te_bracketing::lowest_real_root_quartic(k);
  out[e] = d < 0.0 ? -1.0 : d;
}
__global__ void copy_kernel(const double* c, long n, double* out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double s = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) s += c[i * n + e];
  out[e] = s;
}

int main(int argc, char** argv) {
  const double VN = argc > 1 ? atof(argv[1]) : 0.5, AN = argc > 2 ? atof(argv[2]) : 50.0;
  for (long n : {62500L, 250000L, 1000000L}) {
    std::mt19937_64 g(1);
    std::normal_distribution<double> N(0, 1);
    std::uniform_real_distribution<double> U(0, 1);
    std::vector<double> h(5 * n);
    for (long i = 0; i < n; ++i) {
      double p[3], v[3], a[3];
      for (int k = 0; k < 3; ++k) { p[k] = -10 + 20 * U(g); v[k] = -1 + 2 * U(g) + VN * N(g); a[k] = AN * N(g); }
      h[4 * n + i] = 0.25 * (a[0]*a[0]+a[1]*a[1]+a[2]*a[2]);
      h[3 * n + i] = v[0]*a[0]+v[1]*a[1]+v[2]*a[2];
      h[2 * n + i] = v[0]*v[0]+v[1]*v[1]+v[2]*v[2] + p[0]*a[0]+p[1]*a[1]+p[2]*a[2];
      h[1 * n + i] = 2 * (p[0]*v[0]+p[1]*v[1]+p[2]*v[2]);
      h[0 * n + i] = p[0]*p[0]+p[1]*p[1]+p[2]*p[2] - 1.0;
    }
    double *dc, *dout;
    hipMalloc(&dc, sizeof(double) * 5 * n); hipMalloc(&dout, sizeof(double) * n);
    hipMemcpy(dc, h.data(), sizeof(double) * 5 * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> prev;
    for (int which = 0; which < 3; ++which) {
      const int reps = 50;
      for (int r = 0; r < reps + 5; ++r) {
        if (r == 5) hipEventRecord(e0, 0);
        if (which == 0) hipLaunchKernelGGL(copy_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dc, n, dout);
        else if (which == 1) hipLaunchKernelGGL(solve_bracketing_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dc, n, dout);
        else hipLaunchKernelGGL(solve_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dc, n, dout);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<double> o(n); hipMemcpy(o.data(), dout, sizeof(double) * n, hipMemcpyDeviceToHost);
      long hits = 0; for (double x : o) hits += x >= 0;
      printf("n=%ld %s: %.2f us per launch (%ld >= 0)\n", n, which == 0 ? "copy      " : which == 1 ? "bracketing" : "solve     ", ms * 1e3 / reps, hits);
      if (which == 2) {
        long cls = 0; double worst = 0;
        for (long i = 0; i < n; ++i) {
          if ((o[i] == -1) != (prev[i] == -1)) ++cls;
          else if (o[i] != -1) { double r = fabs(o[i] - prev[i]) / fabs(prev[i]); if (r > worst) worst = r; }
        }
        printf("   vs bracketing: %ld class differences, worst rel %.3g\n", cls, worst);
      }
      prev = o;
      if (which == 2 && n == 1000000L && argc > 3) {   // dump for A/B comparison of solver versions
        FILE* f = fopen(argv[3], "wb");
        if (f) { fwrite(o.data(), sizeof(double), (size_t)n, f); fclose(f); }
      }
    }
    hipFree(dc); hipFree(dout);
  }
  return 0;
}
