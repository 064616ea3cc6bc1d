// rmw_ceiling.hip -- what bounds the step kernels' traffic pattern on this box?  Same bytes, four shapes:
//   tile  : the library's AoSoA lane records (NCH 16-byte chunks per lane, a wave owns a contiguous tile), in place
//   tile2 : the same, but read from buffer A and written to buffer B (ping-pong state)
//   flat  : a plain grid-stride float4 stream, in place (x += 1)
//   flat2 : a plain grid-stride float4 copy A -> B (the guide's 6.29 TB/s "float4 copy")
// each with plain or nontemporal stores.  build: hipcc --offload-arch=gfx950 -O3 tools/rmw_ceiling.hip -o tools/_build/rmw_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NCH, bool NT>
__global__ void __launch_bounds__(256) tile_kernel(const v4f* in, v4f* out, long n_tiles) {
  const int lane = threadIdx.x & 63;
  const long tile = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
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

template <bool NT>
__global__ void __launch_bounds__(256) flat_kernel(const v4f* in, v4f* out, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    v4f v = in[i];
    v.x += 1.0f;
    if constexpr (NT) __builtin_nontemporal_store(v, &out[i]);
    else out[i] = v;
  }
}

static float timed(void (*launch)(void*), void* ctx, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) launch(ctx);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch(ctx);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

struct Ctx { v4f* a; v4f* b; long n_tiles; long n_vec; int mode; };

template <int NCH>
void run(long n_targets) {
  const long n_tiles = (n_targets + 63) / 64;
  const size_t bytes = (size_t)n_tiles * NCH * 64 * 16;
  Ctx c;
  (void)hipMalloc(&c.a, bytes); (void)hipMalloc(&c.b, bytes);
  (void)hipMemset(c.a, 0, bytes); (void)hipMemset(c.b, 0, bytes);
  c.n_tiles = n_tiles; c.n_vec = (long)(bytes / 16);
  const char* names[8] = {"tile  in place      ", "tile  in place, nt  ", "tile2 A->B (swap)   ", "tile2 A->B, nt      ",
                          "flat  in place      ", "flat  in place, nt  ", "flat2 A->B (swap)   ", "flat2 A->B, nt      "};
  for (int mode = 0; mode < 8; ++mode) {
    c.mode = mode;
    auto launch = [](void* p) {
      Ctx* c = (Ctx*)p;
      const unsigned tb = (unsigned)((c->n_tiles + 3) / 4);
      switch (c->mode) {
        case 0: hipLaunchKernelGGL((tile_kernel<NCH, false>), dim3(tb), dim3(256), 0, 0, c->a, c->a, c->n_tiles); break;
        case 1: hipLaunchKernelGGL((tile_kernel<NCH, true>), dim3(tb), dim3(256), 0, 0, c->a, c->a, c->n_tiles); break;
        case 2: hipLaunchKernelGGL((tile_kernel<NCH, false>), dim3(tb), dim3(256), 0, 0, c->a, c->b, c->n_tiles); std::swap(c->a, c->b); break;
        case 3: hipLaunchKernelGGL((tile_kernel<NCH, true>), dim3(tb), dim3(256), 0, 0, c->a, c->b, c->n_tiles); std::swap(c->a, c->b); break;
        case 4: hipLaunchKernelGGL((flat_kernel<false>), dim3(256 * 16), dim3(256), 0, 0, c->a, c->a, c->n_vec); break;
        case 5: hipLaunchKernelGGL((flat_kernel<true>), dim3(256 * 16), dim3(256), 0, 0, c->a, c->a, c->n_vec); break;
        case 6: hipLaunchKernelGGL((flat_kernel<false>), dim3(256 * 16), dim3(256), 0, 0, c->a, c->b, c->n_vec); std::swap(c->a, c->b); break;
        case 7: hipLaunchKernelGGL((flat_kernel<true>), dim3(256 * 16), dim3(256), 0, 0, c->a, c->b, c->n_vec); std::swap(c->a, c->b); break;
      }
    };
    const float ms = timed(launch, &c, 20);
    printf("targets %9ld  %4d B/lane  state %6.0f MB  %s %8.1f us  %6.0f GB/s\n", n_targets, NCH * 16, bytes / 1e6, names[mode], ms * 1e3,
           2.0 * bytes / (ms * 1e-3) / 1e9);
  }
  (void)hipFree(c.a); (void)hipFree(c.b);
}

int main() {
  run<15>(1000000);   // 240 MB: inside the Infinity Cache
  run<30>(1000000);   // 480 MB
  run<7>(10000000);   // 1.1 GB
  run<15>(4000000);   // 0.96 GB
  run<30>(4000000);   // 1.9 GB
  return 0;
}
