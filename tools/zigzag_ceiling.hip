// zigzag_ceiling.hip -- does alternating the traversal direction between ticks keep the tail of the state in the
// 256 MB Infinity Cache?  The step kernels' tile pattern (AoSoA lane records, in place or ping-pong), tick s walking
// the tiles forwards and tick s+1 backwards ("zig-zag"), against forwards every tick.
// build: hipcc --offload-arch=gfx950 -O3 tools/zigzag_ceiling.hip -o tools/_build/zigzag_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NCH, bool NT>
__global__ void __launch_bounds__(256) tile_kernel(const v4f* in, v4f* out, long n_tiles, int reverse) {
  const int lane = threadIdx.x & 63;
  long tile = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  if (reverse) tile = n_tiles - 1 - tile;
  const v4f* ti = in + tile * (long)NCH * 64;
  v4f* to = out + tile * (long)NCH * 64;
  v4f r[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) r[c] = ti[c * 64 + lane];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    r[c].x += 1.0f;
    if constexpr (NT) __builtin_nontemporal_store(r[c], &to[c * 64 + lane]);
    else to[c * 64 + lane] = r[c];
  }
}

template <int NCH>
void run(long n_targets) {
  const long n_tiles = (n_targets + 63) / 64;
  const size_t bytes = (size_t)n_tiles * NCH * 64 * 16;
  v4f *a, *b;
  (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
  (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
  const char* names[6] = {"in place, forwards   ", "in place, zig-zag    ", "ping-pong, forwards  ", "ping-pong, zig-zag   ",
                          "ping-pong nt, fwd    ", "ping-pong nt, zig-zag"};
  const unsigned tb = (unsigned)((n_tiles + 3) / 4);
  for (int mode = 0; mode < 6; ++mode) {
    const bool pp = mode >= 2, nt = mode >= 4, zz = mode & 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    for (int r = 0; r < reps + 4; ++r) {
      if (r == 4) (void)hipEventRecord(e0, 0);
      const int rev = zz ? (r & 1) : 0;
      v4f* out = pp ? b : a;
      if (nt) hipLaunchKernelGGL((tile_kernel<NCH, true>), dim3(tb), dim3(256), 0, 0, a, out, n_tiles, rev);
      else hipLaunchKernelGGL((tile_kernel<NCH, false>), dim3(tb), dim3(256), 0, 0, a, out, n_tiles, rev);
      if (pp) std::swap(a, b);
    }
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("targets %9ld  %4d B/lane  state %6.0f MB  %s %8.1f us  %6.0f GB/s\n", n_targets, NCH * 16, bytes / 1e6, names[mode], ms * 1e3,
           2.0 * bytes / (ms * 1e-3) / 1e9);
  }
  (void)hipFree(a); (void)hipFree(b);
}

int main() {
  run<15>(1000000);   // 240 MB
  run<22>(1000000);   // 352 MB
  run<30>(1000000);   // 480 MB
  run<15>(4000000);   // 0.96 GB
  run<30>(4000000);   // 1.9 GB
  return 0;
}
