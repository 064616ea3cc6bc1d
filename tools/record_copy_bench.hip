// record_copy_bench.hip -- ceiling of the step kernels' access pattern: every thread reads its lane record
// (NCH 16-byte chunks, AoSoA: chunk c of the 64 lanes of a tile contiguous), then writes it back.
// Occupancy is limited with dynamic LDS (bytes per block) to mimic register-heavy kernels.
// build: hipcc --offload-arch=gfx950 -O3 tools/record_copy_bench.hip -o tools/_build/record_copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NCH>
__global__ void __launch_bounds__(256) copy_records(float4* rec, long n_tiles) {
  extern __shared__ char pad[];
  const int lane = threadIdx.x & 63;
  const long tile = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  float4* tb = rec + tile * (long)NCH * 64;
  float4 r[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) r[c] = tb[c * 64 + lane];
  if (pad[0] == 123) r[0].x += 1.f;   // keep the LDS allocation alive
#pragma unroll
  for (int c = 0; c < NCH; ++c) { r[c].x += 1.0f; tb[c * 64 + lane] = r[c]; }
}

template <int NCH>
void run(long n_targets, int lds_bytes) {
  const long n_tiles = (n_targets + 63) / 64;
  const size_t bytes = (size_t)n_tiles * NCH * 64 * 16;
  float4* d; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
  hipFuncSetAttribute((const void*)copy_records<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  for (int r = 0; r < reps + 3; ++r) {
    if (r == 3) hipEventRecord(e0, 0);
    hipLaunchKernelGGL(copy_records<NCH>, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), lds_bytes, 0, d, n_tiles);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int waves_per_cu = lds_bytes > 0 ? (160 * 1024 / lds_bytes) * 4 : 32;
  printf("NCH %2d (%4d B/lane)  lds %6d B/block (<= %2d waves/CU)  %7.1f us  %6.0f GB/s\n", NCH, NCH * 16, lds_bytes, waves_per_cu,
         ms * 1e3 / reps, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
  hipFree(d);
}

int main() {
  const long n = 1000000;
  for (int lds : {0, 20 * 1024, 40 * 1024, 53 * 1024, 80 * 1024, 160 * 1024}) {
    run<15>(n, lds);    // AV fp32 packed groups: 240 B
    run<24>(n, lds);    // AR fp64 full-block-ish
    run<30>(n, lds);    // AR fp64 packed groups: 484 B
  }
  return 0;
}
