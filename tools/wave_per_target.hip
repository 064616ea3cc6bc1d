// wave_per_target.hip -- BASELINE.json north_star's literal design point, measured: ONE WAVEFRONT PER TARGET with the covariance
// tile held in LDS, for the case it was written for: the angular-rates model (n = 18, m = 6), fp64, dense full P.
//
// The library ships something else for this case (6 lanes per target, P in registers, LDS only as the exchange medium:
// csrc/kf_step.hpp, lanes code 6) and DESIGN.md section 2 says why; this tool makes that a measured row instead of an argument
// (VERDICT round 3, item 7).  Same arithmetic as the reference's step for this model, exploiting the band of A and the selector C
// exactly as the shipped dense kernel does:
//     x- = A x,  P- = (A P) A^T + Q                      src/kalman.cpp:84-88, src/types/angular_rates.cpp:108-115
//     S = P-[0:6,0:6] + R,  K = P-[:,0:6] S^-1,  x+ = x- + K (y - x-[0:6]),  P+ = (I - K C) P-        src/kalman.cpp:90-95
//     y = [xyz, unwrap(rpy(normalised quaternion))]       src/types/angular_rates.cpp:81-88
// Layout (this tool's own): per target 345 contiguous doubles [x(18) | unwrap memory(3) | P(324, row-major)]; measurements AoS
// [N][7]: with one target per wavefront every access is a contiguous run of that target's words.
// A wavefront owns 7 KB of LDS (P, the predict's temporary, K, S); lanes take elements e = lane + 64 k of whatever is being formed
// (324 elements of P: 5.06 per lane, 84 % lane use) and every multiply-add operand is read from LDS.
//
// It checks itself against a plain host implementation of the same dense step on the first targets, then times ticks.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wave_per_target.hip -o tools/_build/wave_per_target
//   run:   tools/_build/wave_per_target [targets=1000000] [ticks=20]        (prints one row for profiles/r04_layout_sweep.txt)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

constexpr int N = 18, M = 6, RW = N + 3 + N * N;   // record words
constexpr int WPB = 4;                              // wavefronts per workgroup

// ---- the reference's angle helpers (geometry.hpp:31-76, :154-176), host and device
__host__ __device__ inline double constrain_angle(double x) { x = fmod(x + M_PI, 2 * M_PI); if (x < 0) x += 2 * M_PI; return x - M_PI; }
__host__ __device__ inline double angle_conv(double a) { return fmod(constrain_angle(a), 2 * M_PI); }
__host__ __device__ inline double angle_diff(double a, double b) { double d = fmod(b - a + M_PI, 2 * M_PI); if (d < 0) d += 2 * M_PI; return d - M_PI; }
__host__ __device__ inline double unwrap1(double prev, double now) { return prev - angle_diff(now, angle_conv(prev)); }
__host__ __device__ inline void meas_to_y(const double* m7, double* uw, double* y) {
  double q[4] = {m7[3], m7[4], m7[5], m7[6]};
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= nrm;
  const double x = q[0], yy = q[1], z = q[2], w = q[3];
  double rpy[3];
  const double sp = -2 * (x * z - w * yy);
  if (sp > 0.9999) { rpy[0] = 0; rpy[1] = M_PI / 2; rpy[2] = 2 * atan2(z, w); }
  else if (sp < -0.9999) { rpy[0] = 0; rpy[1] = -M_PI / 2; rpy[2] = 2 * atan2(z, w); }
  else { rpy[0] = atan2(2 * (yy * z + w * x), w * w - x * x - yy * yy + z * z); rpy[1] = asin(sp); rpy[2] = atan2(2 * (x * yy + w * z), w * w + x * x - yy * yy - z * z); }
  for (int i = 0; i < 3; ++i) { y[i] = m7[i]; y[3 + i] = unwrap1(uw[i], rpy[i]); uw[i] = y[3 + i]; }
}

__device__ __forceinline__ void wave_sync() {   // same-wavefront LDS hand-off: DS operations of a wave execute in order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct WaveLds { double P[N * N], T[N * N], K[N * M], S[M * M], x[N], nu[M]; };

__global__ void __launch_bounds__(64 * WPB) step_wave_per_target(double* rec, const double* meas, const double* qr /* Q(324) R(36) */, long n, double dt) {
  __shared__ WaveLds lds[WPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long t = (long)blockIdx.x * WPB + wave;
  if (t >= n) return;
  WaveLds& L = lds[wave];
  double* r = rec + t * RW;
  const double* Q = qr;
  const double* R = qr + N * N;
  const double hdt = 0.5 * dt * dt;
  // record -> LDS (coalesced: 512 contiguous bytes per load instruction)
  for (int e = lane; e < N * N; e += 64) L.P[e] = r[N + 3 + e];
  if (lane < N) L.x[lane] = r[lane];
  double uw[3] = {r[N], r[N + 1], r[N + 2]};
  double m7[7], y[M];
  for (int i = 0; i < 7; ++i) m7[i] = meas[t * 7 + i];
  meas_to_y(m7, uw, y);                               // every lane the same values (one target per wavefront)
  wave_sync();
  // x- = A x (rows r, r+6, r+12 form a [p v a] chain)
  double xm = 0;
  if (lane < N) {
    xm = L.x[lane];
    if (lane + 6 < N) xm = fma(dt, L.x[lane + 6], xm);
    if (lane + 12 < N) xm = fma(hdt, L.x[lane + 12], xm);
  }
  // T = A P (rows)
  for (int e = lane; e < N * N; e += 64) {
    const int rr = e / N, c = e % N;
    double v = L.P[e];
    if (rr + 6 < N) v = fma(dt, L.P[(rr + 6) * N + c], v);
    if (rr + 12 < N) v = fma(hdt, L.P[(rr + 12) * N + c], v);
    L.T[e] = v;
  }
  wave_sync();
  if (lane < N) L.x[lane] = xm;
  // P- = T A^T + Q (columns)
  for (int e = lane; e < N * N; e += 64) {
    const int rr = e / N, c = e % N;
    double v = L.T[e];
    if (c + 6 < N) v = fma(dt, L.T[rr * N + c + 6], v);
    if (c + 12 < N) v = fma(hdt, L.T[rr * N + c + 12], v);
    L.P[e] = v + Q[e];
  }
  wave_sync();
  // S = P-[0:6,0:6] + R, inverted in place by Gauss-Jordan (lanes 0..35 own one entry each)
  if (lane < M * M) L.S[lane] = L.P[(lane / M) * N + lane % M] + R[lane];
  if (lane < M) L.nu[lane] = y[lane] - L.x[lane];
  wave_sync();
  for (int p = 0; p < M; ++p) {
    const int rr = lane / M, c = lane % M;
    double v = 0, piv = 0, f = 0, prow = 0;
    if (lane < M * M) { piv = L.S[p * M + p]; v = L.S[lane]; f = L.S[rr * M + p]; prow = L.S[p * M + c]; }
    wave_sync();
    if (lane < M * M) {
      const double inv = 1.0 / piv;
      const double pr = (c == p ? 1.0 : prow) * inv;          // the scaled pivot row (its pivot entry becomes 1 * inv)
      double nv;
      if (rr == p) nv = pr;
      else nv = fma(-f, pr, c == p ? 0.0 : v);
      L.S[lane] = nv;
    }
    wave_sync();
  }
  // K = P-[:,0:6] S^-1
  for (int e = lane; e < N * M; e += 64) {
    const int rr = e / M, l = e % M;
    double acc = L.P[rr * N] * L.S[l];
    for (int c = 1; c < M; ++c) acc = fma(L.P[rr * N + c], L.S[c * M + l], acc);
    L.K[e] = acc;
  }
  wave_sync();
  // x+ = x- + K nu
  if (lane < N) {
    double acc = L.K[lane * M] * L.nu[0];
    for (int c = 1; c < M; ++c) acc = fma(L.K[lane * M + c], L.nu[c], acc);
    r[lane] = L.x[lane] + acc;
  }
  if (lane < 3) r[N + lane] = uw[lane];
  // P+ = (I - K C) P- : row r of (I - K C) is e_r - K[r][0:6] on the first six columns
  for (int e = lane; e < N * N; e += 64) {
    const int rr = e / N, c = e % N;
    double acc = ((rr == 0 ? 1.0 : 0.0) - L.K[rr * M]) * L.P[c];
    for (int j = 1; j < M; ++j) acc = fma((rr == j ? 1.0 : 0.0) - L.K[rr * M + j], L.P[j * N + c], acc);
    if (rr >= M) acc += L.P[e];
    r[N + 3 + e] = acc;
  }
}

// ---- plain host implementation of the same dense step (reference order of operations, no structure exploited)
static void host_step(double* r, const double* m7, const double* Q, const double* R, double dt) {
  double A[N][N] = {}, x[N], P[N][N], T[N][N], Pm[N][N];
  for (int i = 0; i < N; ++i) { A[i][i] = 1; if (i + 6 < N) A[i][i + 6] = dt; if (i + 12 < N) A[i][i + 12] = 0.5 * dt * dt; }
  double uw[3] = {r[N], r[N + 1], r[N + 2]}, y[M];
  meas_to_y(m7, uw, y);
  for (int i = 0; i < N; ++i) { x[i] = 0; for (int j = 0; j < N; ++j) x[i] += A[i][j] * r[j]; }
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) P[i][j] = r[N + 3 + i * N + j];
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < N; ++k) s += A[i][k] * P[k][j]; T[i][j] = s; }
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < N; ++k) s += T[i][k] * A[j][k]; Pm[i][j] = s + Q[i * N + j]; }
  double S[M][2 * M];
  for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) { S[i][j] = Pm[i][j] + R[i * M + j]; S[i][M + j] = i == j; }
  for (int p = 0; p < M; ++p) {
    int best = p;
    for (int i = p + 1; i < M; ++i) if (fabs(S[i][p]) > fabs(S[best][p])) best = i;
    for (int j = 0; j < 2 * M; ++j) std::swap(S[p][j], S[best][j]);
    const double inv = 1.0 / S[p][p];
    for (int j = 0; j < 2 * M; ++j) S[p][j] *= inv;
    for (int i = 0; i < M; ++i) if (i != p) { const double f = S[i][p]; for (int j = 0; j < 2 * M; ++j) S[i][j] -= f * S[p][j]; }
  }
  double K[N][M];
  for (int i = 0; i < N; ++i) for (int l = 0; l < M; ++l) { double s = 0; for (int c = 0; c < M; ++c) s += Pm[i][c] * S[c][M + l]; K[i][l] = s; }
  for (int i = 0; i < N; ++i) { double s = 0; for (int c = 0; c < M; ++c) s += K[i][c] * (y[c] - x[c]); r[i] = x[i] + s; }
  for (int i = 0; i < 3; ++i) r[N + i] = uw[i];
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
    double s = 0;
    for (int k = 0; k < M; ++k) s += ((i == k ? 1.0 : 0.0) - K[i][k]) * Pm[k][j];
    if (i >= M) s += Pm[i][j];
    r[N + 3 + i * N + j] = s;
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1000000;
  const int ticks = argc > 2 ? atoi(argv[2]) : 20;
  const double dt = 0.004;
  // the shipped model's structure: Q = Gamma diag(sigma^2) Gamma^T per axis, R and P0 diagonal (matlab/generateModel.m:9-41)
  std::vector<double> qr(N * N + M * M, 0.0);
  const double g[3] = {dt * dt * dt / 6, dt * dt / 2, dt};
  for (int ax = 0; ax < 6; ++ax) {
    const double s2 = ax < 3 ? 1e-6 : 1e-4;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) qr[(ax + 6 * a) * N + ax + 6 * b] = s2 * g[a] * g[b];
    qr[N * N + ax * M + ax] = ax < 3 ? 1e-4 : 1e-2;
  }
  std::vector<double> rec((size_t)n * RW, 0.0), meas((size_t)n * 7 * 2);
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1, 1);
  std::normal_distribution<double> G(0, 0.01);
  for (long t = 0; t < n; ++t) {
    double* r = &rec[(size_t)t * RW];
    for (int i = 0; i < 3; ++i) r[i] = 10 * U(rng);
    for (int i = 0; i < N; ++i) r[N + 3 + i * N + i] = i < 3 ? 0.1 : 0.01;
    for (int k = 0; k < 2; ++k) {
      double* m = &meas[((size_t)k * n + t) * 7];
      for (int i = 0; i < 3; ++i) m[i] = r[i] + G(rng);
      const double a = 0.3 * U(rng) + 0.05 * k;
      m[3] = sin(a / 2); m[4] = 0.02 * U(rng); m[5] = 0.02 * U(rng); m[6] = cos(a / 2);
    }
  }
  double *d_rec, *d_meas, *d_qr;
  CHECK(hipMalloc(&d_rec, rec.size() * 8)); CHECK(hipMalloc(&d_meas, meas.size() * 8)); CHECK(hipMalloc(&d_qr, qr.size() * 8));
  CHECK(hipMemcpy(d_rec, rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_meas, meas.data(), meas.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_qr, qr.data(), qr.size() * 8, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)((n + WPB - 1) / WPB);
  // two ticks, checked against the host on the first 64 targets
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(step_wave_per_target, dim3(blocks), dim3(64 * WPB), 0, 0, d_rec, d_meas + (size_t)k * n * 7, d_qr, n, dt);
  CHECK(hipDeviceSynchronize());
  const long nchk = n < 64 ? n : 64;
  std::vector<double> got((size_t)nchk * RW);
  CHECK(hipMemcpy(got.data(), d_rec, got.size() * 8, hipMemcpyDeviceToHost));
  double worst_x = 0, worst_P = 0;
  for (long t = 0; t < nchk; ++t) {
    double* r = &rec[(size_t)t * RW];
    for (int k = 0; k < 2; ++k) host_step(r, &meas[((size_t)k * n + t) * 7], qr.data(), qr.data() + N * N, dt);
    double pmax = 0;
    for (int e = 0; e < N * N; ++e) pmax = fmax(pmax, fabs(r[N + 3 + e]));
    for (int i = 0; i < N + 3; ++i) worst_x = fmax(worst_x, fabs(got[(size_t)t * RW + i] - r[i]) / fmax(1.0, fabs(r[i])));
    for (int e = 0; e < N * N; ++e) worst_P = fmax(worst_P, fabs(got[(size_t)t * RW + N + 3 + e] - r[N + 3 + e]) / pmax);
  }
  const bool ok = worst_x < 1e-12 && worst_P < 1e-10;
  fprintf(stderr, "check against the host on %ld targets after 2 ticks: max |dx| %.2e (rel), max |dP| / max|P| %.2e  %s\n", nchk, worst_x, worst_P, ok ? "ok" : "FAILED");
  if (!ok) return 1;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(step_wave_per_target, dim3(blocks), dim3(64 * WPB), 0, 0, d_rec, d_meas + (size_t)(k & 1) * n * 7, d_qr, n, dt);
  CHECK(hipEventRecord(e0, 0));
  for (int k = 0; k < ticks; ++k) hipLaunchKernelGGL(step_wave_per_target, dim3(blocks), dim3(64 * WPB), 0, 0, d_rec, d_meas + (size_t)(k & 1) * n * 7, d_qr, n, dt);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / ticks;
  const double bytes = (2.0 * RW + 7) * 8;          // record read + written, measurement read: 5576 B, SURVEY 8d's full-P figure for this model
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(step_wave_per_target)));
  printf("angular_rates          f64  W64 %9ld %10.2f %12.4g %9.0f %6.3f   regs %3d scratch %3zu lds %5zu B / workgroup of %d waves  <- north_star's literal form: one wavefront per target, P in LDS (tools/wave_per_target.hip)\n",
         n, us, n / (us * 1e-6), n * bytes / (us * 1e-6) / 1e9, n * bytes / (us * 1e-6) / 1e9 / 8000.0, fa.numRegs, (size_t)fa.localSizeBytes, (size_t)fa.sharedSizeBytes, WPB);
  return 0;
}
