// ekf_sym.hpp -- one predict(+update) tick of the angular-velocities EKF on a SYMMETRIC-PACKED covariance, thread per target,
// everything in registers (the (EKF, LAYOUT_PACKED, G = 1) case of kf_step_kernel).
//
// Reference arithmetic: ExtendedKalmanFilter::predict / estimate (src/kalman.cpp:129-140) with the model of
// src/types/angular_velocities.cpp:116-140 -- x = [p(0:3) rpy(3:6) v(6:9) w(9:12)], C = [I6 0]:
//     P^- = A P A^T + Q,   A = I + N,   N[0:3,6:9] = dt I,  N[3:6,3:6] = Jr - I,  N[3:6,9:12] = Jw      (Jacobians at x_{k-1})
//     S = P^-[0:6,0:6] + R,  K = P^-[:,0:6] S^-1,  x^+ = x^- + K (y - x^-[0:6]),  P^+ = (I - K C) P^-
// The generic dense kernel keeps the full 12 x 12 image plus its row exchanges in registers (408 of them in fp32: one
// wavefront per SIMD).  With P symmetric -- which a packed batch is by construction -- only the upper triangle (78 words) is
// held and updated IN PLACE:
//     predict : M = N P has six non-zero rows; rows 0..2 are dt * P[6+j][:] (never materialised), rows 3..5 take 36
//               temporaries.  P^-[r][c] = P[r][c] + M[r][c] + M[c][r] + (M N^T)[r][c] + Q[r][c], each term where it exists
//               (r < 6, c < 6, both).  Order: top-left block, then top-right, then the bottom-right (+Q only), so that
//               every read still sees the prior covariance.
//     update  : W = P^-[:,0:6] is a VIEW of the triangle (W[c][l] = P^-[l][c]); the gain U = W S^-1 is formed and consumed in
//               pieces (rows 6..8, 9..11: 18 registers each; rows 0..5: 36); P^+[r][c] = P^-[r][c] - sum_l U[r][l] W[c][l] for r <= c, bottom-right first,
//               then the top-right and the top-left block column by column (six temporaries; ascending in the top-left,
//               so that every read still sees the prior block).
// The products are the reference's, summed k / l ascending with fma; what differs from the full-P kernel is the association
// A P A^T = P + NP + PN^T + NPN^T (rounding level, like every packed layout; stated tolerance unchanged).
// Register budget (fp32): 93 words of record + S^-1 (36) + a piece of the gain (18 / 36) + innovation (6) = 160 allocated, three
// wavefronts per SIMD -- given that (a) the record words are detached from their 16-byte load / store tuples
// (kf_step.hpp opaque_copy) and (b) the SLP vectorizer is off for this translation unit (kf_model_av_sym.hip): it hoists
// blocks across the sched_barriers below to build v_pk_fma_f32 pairs and costs 120 registers.
#pragma once
#include <hip/hip_runtime.h>

#include "te_device_math.hpp"
#include "te_layout.hpp"

namespace te {

// mem: the HBM image of the record, [ triangle (78) | x (12) | unwrap memory (3) ] (Cfg<ModelAV, T, 1, LAYOUT_PACKED>);
// Qm: Q 12 x 12 row-major, Rm: R 6 x 6 row-major (LDS, or the target's class row); yxyz: measured position;
// mrpy: measured Euler angles (already converted from the quaternion), both only read when `has`.
template <class C, typename T>
__device__ __forceinline__ void ekf_sym_tick(T* mem, const T* Qm, const T* Rm, const T dt, const bool has, const T* yxyz, const T* mrpy) {
  using F = Mth<T>;
  constexpr int N = 12, K = 6;
  static_assert(C::N == N && C::K == K && C::PK && C::G == 1, "angular-velocities EKF, symmetric-packed, thread per target");
#define TRI_(r, c) ((r) <= (c) ? C::tri((r), (c)) : C::tri((c), (r)))
#define PS_(r, c) mem[TRI_(r, c)]
#define XS_(r) mem[C::X_OFF + (r)]
#define US_(s) mem[C::UW_OFF + (s)]

  // ---- Jacobians at the previous posterior (geometry.hpp:359-374, :394-426), exactly the dense kernel's expressions
  T s_r, c_r, s_p, c_p;
  F::sincos(XS_(3), &s_r, &c_r);
  F::sincos(XS_(4), &s_p, &c_p);
  const T wy = XS_(10), wz = XS_(11);
  T Jm[3][3], Jw[3][3], Ei[3][3];   // Jm = Jr - I
  Jm[0][0] = (dt * (wy * c_r * s_p - wz * s_p * s_r)) / c_p;
  Jm[0][1] = (dt * (wz * c_r + wy * s_r)) / (c_p * c_p);
  Jm[0][2] = 0;
  Jm[1][0] = -dt * (wz * c_r + wy * s_r);
  Jm[1][1] = 0;
  Jm[1][2] = 0;
  Jm[2][0] = (dt * (wy * c_r - wz * s_r)) / c_p;
  Jm[2][1] = (dt * s_p * (wz * c_r + wy * s_r)) / (c_p * c_p);
  Jm[2][2] = 0;
  Jw[0][0] = dt; Jw[0][1] = (dt * s_p * s_r) / c_p; Jw[0][2] = (dt * c_r * s_p) / c_p;
  Jw[1][0] = 0;  Jw[1][1] = dt * c_r;               Jw[1][2] = -dt * s_r;
  Jw[2][0] = 0;  Jw[2][1] = (dt * s_r) / c_p;       Jw[2][2] = (dt * c_r) / c_p;
  Ei[0][0] = 1; Ei[0][1] = (s_p * s_r) / c_p; Ei[0][2] = (c_r * s_p) / c_p;
  Ei[1][0] = 0; Ei[1][1] = c_r;               Ei[1][2] = -s_r;
  Ei[2][0] = 0; Ei[2][1] = s_r / c_p;         Ei[2][2] = c_r / c_p;

  // ---- x^- = f(x) (angular_velocities.cpp:126-140)
  {
    T nr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      T acc = (dt * Ei[c][0]) * XS_(9);
      acc = F::fma(dt * Ei[c][1], XS_(10), acc);
      acc = F::fma(dt * Ei[c][2], XS_(11), acc);
      nr[c] = XS_(3 + c) + acc;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      XS_(c) = F::fma(dt, XS_(6 + c), XS_(c));
      XS_(3 + c) = nr[c];
    }
  }

  // ---- P^- = A P A^T + Q on the triangle
  T Ma[3][N];   // rows 3..5 of M = N P
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int c = 0; c < N; ++c) {
      T v = Jm[j][0] * PS_(3, c);
      v = F::fma(Jm[j][1], PS_(4, c), v);
      v = F::fma(Jm[j][2], PS_(5, c), v);
      v = F::fma(Jw[j][0], PS_(9, c), v);
      v = F::fma(Jw[j][1], PS_(10, c), v);
      v = F::fma(Jw[j][2], PS_(11, c), v);
      Ma[j][c] = v;
    }
  // M[r][c], r < 6, from the PRIOR covariance (rows 0..2: dt * P[6+r][c])
#define M_(r, c) ((r) < 3 ? dt * PS_(6 + (r), (c)) : Ma[(r) - 3][(c)])
  // top-left block (r <= c < 6): reads top-right and bottom-right prior entries, so it goes first
#pragma unroll
  for (int r = 0; r < K; ++r)
#pragma unroll
    for (int c = r; c < K; ++c) {
      T mnt;   // (M N^T)[r][c] = sum_k M[r][k] N[c][k]
      if (c < 3) {
        mnt = dt * M_(r, 6 + c);
      } else {
        mnt = M_(r, 3) * Jm[c - 3][0];
        mnt = F::fma(M_(r, 4), Jm[c - 3][1], mnt);
        mnt = F::fma(M_(r, 5), Jm[c - 3][2], mnt);
        mnt = F::fma(M_(r, 9), Jw[c - 3][0], mnt);
        mnt = F::fma(M_(r, 10), Jw[c - 3][1], mnt);
        mnt = F::fma(M_(r, 11), Jw[c - 3][2], mnt);
      }
      PS_(r, c) = ((PS_(r, c) + M_(r, c)) + M_(c, r)) + mnt;
    }
  // top-right block (r < 6 <= c): reads the bottom-right prior entries and Ma
#pragma unroll
  for (int r = 0; r < K; ++r)
#pragma unroll
    for (int c = K; c < N; ++c) PS_(r, c) = PS_(r, c) + M_(r, c);
#undef M_
  // + Q, last and in a pass of its own (after the barrier only the triangle is live: the 78 loads of Q cannot pile up on
  // top of the temporaries of the products above)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < N; ++r) {
#pragma unroll
    for (int c = r; c < N; ++c) PS_(r, c) = PS_(r, c) + Qm[r * N + c];
    __builtin_amdgcn_sched_barrier(0);   // row by row: at most one row of Q in flight
  }

  if (!has) return;

  // ---- estimate: S^-1 by unpivoted Gauss-Jordan (S is SPD), as the dense kernel
  T S[K][K];
#pragma unroll
  for (int r = 0; r < K; ++r)
#pragma unroll
    for (int c = 0; c < K; ++c) S[r][c] = PS_(r, c) + Rm[r * K + c];
#pragma unroll
  for (int p = 0; p < K; ++p) {
    const T inv = (T)1 / S[p][p];
    S[p][p] = 1;
#pragma unroll
    for (int c = 0; c < K; ++c) S[p][c] *= inv;
#pragma unroll
    for (int r = 0; r < K; ++r) {
      if (r == p) continue;
      const T f = S[r][p];
      S[r][p] = 0;
#pragma unroll
      for (int c = 0; c < K; ++c) S[r][c] = F::fma(-f, S[p][c], S[r][c]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // innovation: xyz, and the unwrapped Euler angles (angular_velocities.cpp:89-96)
  T nu[K];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    nu[c] = yxyz[c] - XS_(c);
    const T y = unwrap_angle(US_(c), mrpy[c]);
    US_(c) = y;
    nu[3 + c] = y - XS_(3 + c);
  }
  // The gain U = P^-[:,0:6] S^-1 (12 x 6) is formed and consumed in pieces: rows 6..8 and 9..11 first (18 registers each; they
  // update the bottom-right block, which reads the top-right PRIOR entries), then rows 0..5 (top-right, then top-left).
  // W[c][l] = P^-[l][c] is a view of the triangle.
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    constexpr int H = K / 2;
    T U[H][K];   // rows 6 + 3 h .. 8 + 3 h of the gain
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
      for (int c = 0; c < K; ++c)
#pragma unroll
        for (int r = 0; r < H; ++r) U[r][l] = (c == 0) ? PS_(K + H * h + r, 0) * S[0][l] : F::fma(PS_(K + H * h + r, c), S[c][l], U[r][l]);
#pragma unroll
    for (int r = 0; r < H; ++r) {
      T acc = U[r][0] * nu[0];
#pragma unroll
      for (int l = 1; l < K; ++l) acc = F::fma(U[r][l], nu[l], acc);
      XS_(K + H * h + r) += acc;
    }
    // (a) bottom-right (6 <= r <= c)
#pragma unroll
    for (int r = 0; r < H; ++r)
#pragma unroll
      for (int c = K + H * h + r; c < N; ++c) {
        T acc = U[r][0] * PS_(0, c);
#pragma unroll
        for (int l = 1; l < K; ++l) acc = F::fma(U[r][l], PS_(l, c), acc);
        PS_(K + H * h + r, c) = PS_(K + H * h + r, c) - acc;
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    T U[K][K];   // rows 0..5 of the gain (the top-left block is still the prior one)
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
      for (int c = 0; c < K; ++c)
#pragma unroll
        for (int r = 0; r < K; ++r) U[r][l] = (c == 0) ? PS_(r, 0) * S[0][l] : F::fma(PS_(r, c), S[c][l], U[r][l]);
#pragma unroll
    for (int r = 0; r < K; ++r) {
      T acc = U[r][0] * nu[0];
#pragma unroll
      for (int l = 1; l < K; ++l) acc = F::fma(U[r][l], nu[l], acc);
      XS_(r) += acc;
    }
    // (b) top-right, one column at a time; (c) top-left, columns ascending: column c needs the prior P^-[l][c] for every
    // l < 6 -- its own column for l <= c and, by symmetry, row c of the LATER columns for l > c -- so six temporaries per
    // column do, no copy of the block
#pragma unroll
    for (int cc = 0; cc < N; ++cc) {
      const int c = (cc < K) ? cc + K : cc - K;   // columns 6..11 first, then 0..5 ascending
      T old[K];
#pragma unroll
      for (int l = 0; l < K; ++l) old[l] = PS_(l, c);
#pragma unroll
      for (int r = 0; r < K; ++r) {
        if (r > c) continue;   // upper triangle only (top-left block)
        T acc = U[r][0] * old[0];
#pragma unroll
        for (int l = 1; l < K; ++l) acc = F::fma(U[r][l], old[l], acc);
        PS_(r, c) = old[r] - acc;
      }
    }
  }
#undef TRI_
#undef PS_
#undef XS_
#undef US_
}

}  // namespace te
