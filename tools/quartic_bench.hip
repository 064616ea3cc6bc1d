// quartic_bench.hip -- the sphere query's quartic solver (csrc/te_quartic.hpp) alone, on the quartics the fused query of configs[4]'s
// per-GPU share really meets (tools/dump_query_coeffs.py), one wavefront per workgroup as in the small-population step launch:
// shader-clock cycles per wavefront between the header's TE_QTS marks, the kernel's duration, and -- with a second copy of the header
// (tools/_build/te_quartic_base.hpp, e.g. `git show <commit>:target_estimation_amd/csrc/te_quartic.hpp`; -DWITH_BASE) -- the two
// versions compared result by result.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWITH_BASE] -o tools/_build/quartic_bench tools/quartic_bench.hip
//   tools/_build/quartic_bench tools/_build/query_coeffs_f32.npy
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define NMARK 16
__device__ long long* g_marks;   // [waves][NMARK]: the clock when the wavefront (any of its lanes, last writer) passed mark i
#ifdef NO_MARKS   // the solver as the product compiles it: only the whole-solve cycles and the kernel's duration
#define TE_QTS(i)
#else
#define TE_QTS(i) do { __builtin_amdgcn_sched_barrier(0); g_marks[(long)blockIdx.x * NMARK + (i)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
#include "../target_estimation_amd/csrc/te_quartic.hpp"
#ifdef WITH_BASE
#undef TE_QTS
#define TE_QTS(i)
#define te te_base
#include "_build/te_quartic_base.hpp"
#undef te
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V>
__global__ void solve_kernel(const double* c, long n, double* d, long long* cyc, long long* marks) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  if (V == 0 && threadIdx.x == 0 && blockIdx.x == 0) {}
  double cc[5] = {1, 0, 0, 0, 0};
  if (i < n)
    for (int k = 0; k < 5; ++k) cc[k] = c[i * 5 + k];
  __builtin_amdgcn_sched_barrier(0);
  const long long t0 = (long long)__builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  double r;
  if constexpr (V == 0) r = te::first_crossing_quartic(cc);
#ifdef WITH_BASE
  else r = te_base::first_crossing_quartic(cc);
#endif
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = (long long)__builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  if (i < n) d[i] = r;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; marks[(long)blockIdx.x * NMARK + NMARK - 1] = t0; }
}

int main(int argc, char** argv) {
  FILE* f = fopen(argc > 1 ? argv[1] : "tools/_build/query_coeffs_f32.npy", "rb");
  if (!f) { perror("coefficients"); return 1; }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f) - 128;
  fseek(f, 128, SEEK_SET);
  const long n = bytes / 40;
  std::vector<double> c((size_t)n * 5);
  if (fread(c.data(), 8, c.size(), f) != c.size()) return 1;
  const long waves = (n + 63) / 64;
  double *dc, *dd;
  long long *dcyc, *dmarks;
  CHECK(hipMalloc(&dc, c.size() * 8));
  CHECK(hipMalloc(&dd, n * 8));
  CHECK(hipMalloc(&dcyc, waves * 8));
  CHECK(hipMalloc(&dmarks, waves * NMARK * 8));
  CHECK(hipMemcpy(dc, c.data(), c.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &dmarks, sizeof(dmarks)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<double> res[2];
  for (int v = 0; v < 2; ++v) {
#ifndef WITH_BASE
    if (v == 1) break;
#endif
    CHECK(hipMemset(dmarks, 0, waves * NMARK * 8));
    for (int rep = 0; rep < 3; ++rep) {
      if (v == 0) hipLaunchKernelGGL(solve_kernel<0>, dim3((unsigned)waves), dim3(64), 0, 0, dc, n, dd, dcyc, dmarks);
      else hipLaunchKernelGGL(solve_kernel<1>, dim3((unsigned)waves), dim3(64), 0, 0, dc, n, dd, dcyc, dmarks);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) {
      if (v == 0) hipLaunchKernelGGL(solve_kernel<0>, dim3((unsigned)waves), dim3(64), 0, 0, dc, n, dd, dcyc, dmarks);
      else hipLaunchKernelGGL(solve_kernel<1>, dim3((unsigned)waves), dim3(64), 0, 0, dc, n, dd, dcyc, dmarks);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc(waves), marks((size_t)waves * NMARK);
    res[v].resize(n);
    CHECK(hipMemcpy(cyc.data(), dcyc, waves * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(marks.data(), dmarks, waves * NMARK * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(res[v].data(), dd, n * 8, hipMemcpyDeviceToHost));
    double mean = 0; long long mx = 0;
    for (long w = 0; w < waves; ++w) { mean += (double)cyc[w]; if (cyc[w] > mx) mx = cyc[w]; }
    long hits = 0;
    for (long i = 0; i < n; ++i) hits += res[v][i] > -1;
    printf("%s: %ld quartics, %ld crossings; kernel %.2f us; cycles per wavefront mean %.0f max %lld\n", v == 0 ? "te_quartic.hpp" : "base copy", n, hits,
           ms / reps * 1e3, mean / waves, mx);
    if (v == 0) {   // phases: mark i relative to the previous mark that the wavefront passed
      printf("  marks (mean cycles since the solver's start, over the wavefronts that passed the mark; share of wavefronts):\n");
      for (int i = 0; i < NMARK - 1; ++i) {
        double s = 0; long cnt = 0;
        for (long w = 0; w < waves; ++w) {
          const long long m = marks[(size_t)w * NMARK + i], t0 = marks[(size_t)w * NMARK + NMARK - 1];
          if (m != 0) { s += (double)(m - t0); ++cnt; }
        }
        if (cnt) printf("    mark %2d: %7.0f   (%5.1f %%)\n", i, s / cnt, 100.0 * cnt / waves);
      }
    }
  }
#ifdef WITH_BASE
  long diff_class = 0; double max_rel = 0;
  for (long i = 0; i < n; ++i) {
    const double a = res[0][i], b = res[1][i];
    if ((a > -1) != (b > -1)) { ++diff_class; continue; }
    if (a > -1) { const double r = fabs(a - b) / fmax(fabs(b), 1e-300); if (r > max_rel) max_rel = r; }
  }
  printf("against the base copy: %ld quartics decided differently (crossing / none), largest relative difference of a crossing time %.3g\n", diff_class, max_rel);
#endif
  return 0;
}
