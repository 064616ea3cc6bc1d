// record_copy_nt.hip -- cache-policy experiment for the step kernels' access pattern (in-place read-modify-write of
// AoSoA lane records): plain vs nontemporal loads / stores, state inside and beyond the 256 MB Infinity Cache.
// build: hipcc --offload-arch=gfx950 -O3 tools/record_copy_nt.hip -o tools/_build/record_copy_nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NCH, int MODE>
__global__ void __launch_bounds__(256) copy_records(v4f* rec, long n_tiles) {
  const int lane = threadIdx.x & 63;
  const long tile = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  v4f* tb = rec + tile * (long)NCH * 64;
  v4f r[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if constexpr (MODE & 1) r[c] = __builtin_nontemporal_load(&tb[c * 64 + lane]);
    else r[c] = tb[c * 64 + lane];
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    r[c].x += 1.0f;
    if constexpr (MODE & 2) __builtin_nontemporal_store(r[c], &tb[c * 64 + lane]);
    else tb[c * 64 + lane] = r[c];
  }
}

template <int NCH, int MODE>
void run(long n_targets) {
  const long n_tiles = (n_targets + 63) / 64;
  const size_t bytes = (size_t)n_tiles * NCH * 64 * 16;
  v4f* d; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  for (int r = 0; r < reps + 3; ++r) {
    if (r == 3) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((copy_records<NCH, MODE>), dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, 0, d, n_tiles);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  static const char* names[4] = {"plain ld / plain st", "nt ld / plain st", "plain ld / nt st", "nt ld / nt st"};
  printf("targets %9ld  NCH %2d (%4d B/lane, state %6.0f MB)  %-20s %8.1f us  %6.0f GB/s\n", n_targets, NCH, NCH * 16, bytes / 1e6, names[MODE],
         ms * 1e3 / reps, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
  hipFree(d);
}

template <int NCH>
void all(long n) { run<NCH, 0>(n); run<NCH, 1>(n); run<NCH, 2>(n); run<NCH, 3>(n); }

int main() {
  for (long n : {1000000L, 4000000L, 10000000L}) {
    all<7>(n);     // UV fp64 packed groups: 120 B
    all<15>(n);    // AV fp32 packed groups: 240 B
    all<30>(n);    // AR fp64 packed groups: 484 B
  }
  return 0;
}
