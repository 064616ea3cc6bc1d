// kf_aux.hpp -- record access, target construction and output derivation kernels.
// Not bandwidth-critical (one thread per target or per row); the step kernel is kf_step.hpp.
#pragma once
#include <hip/hip_runtime.h>

#include "te_device_math.hpp"
#include "te_layout.hpp"
#include "te_quartic.hpp"

namespace te {

// pointer to element (r, c) of P (c < N) or to x[r] (c == N) of `slot`; nullptr for a
// structurally zero element of the separable layout
template <class C, typename T>
__device__ __forceinline__ T* state_ptr(char* rec, long slot, int r, int c) {
  const long tile = slot / C::TPW;
  const int lane = (int)(slot % C::TPW) * C::G + ((c >= C::N) ? (r % C::G) : C::p_lane(r, c));
  const int q = (r % C::K) / C::G + (r / C::K) * C::KPL;
  const int w = (c >= C::N) ? C::X_OFF + q : C::p_word(r, c);
  if (w < 0) return nullptr;
  return reinterpret_cast<T*>(rec + tile * C::TILE_BYTES + record_word_offset<C, T>(lane, w));
}
template <class C, typename T>
__device__ __forceinline__ T state_get(char* rec, long slot, int r, int c) {
  const T* p = state_ptr<C, T>(rec, slot, r, c);
  return p ? *p : (T)0;
}
template <class C, typename T>
__device__ __forceinline__ void state_set(char* rec, long slot, int r, int c, T v) {
  T* p = state_ptr<C, T>(rec, slot, r, c);
  if (p) *p = v;
}
// pointer to unwrap-memory component cc (0..2) of `slot`
template <class C, typename T>
__device__ __forceinline__ T* unwrap_ptr(char* rec, long slot, int cc) {
  const int r = 3 + cc;
  const long tile = slot / C::TPW;
  const int lane = (int)(slot % C::TPW) * C::G + (r % C::G);
  const int w = C::UW_OFF + cc / C::G;
  return reinterpret_cast<T*>(rec + tile * C::TILE_BYTES + record_word_offset<C, T>(lane, w));
}

// geometry.hpp:619-628 pose7dToPose6d
template <typename T> __device__ __forceinline__ void pose7_to_pose6(const T* p7, T* p6) {
  T q[4] = {p7[3], p7[4], p7[5], p7[6]};
  p6[0] = p7[0]; p6[1] = p7[1]; p6[2] = p7[2];
  quat_normalize(q);
  quat_to_rpy(q, p6 + 3);
}

struct InitArgs {
  char* rec;
  const int* idx;          // slot of entry e
  long n;
  const double* p0;        // [n][7]
  const double* v0;        // [n][6] or null
  const double* a0;        // [n][6] or null
  const double* P0;        // [N*N] shared, [n][N*N] when per_target_P0, or a table indexed by P0_index
  int per_target_P0;
  const int* P0_index = nullptr;   // [n]: entry e starts from P0 + P0_index[e] * N*N (per-class initial covariances)
  int* cls = nullptr;              // per-slot parameter class array of the batch
  const int* cls_of = nullptr;     // [n] class of entry e, or null: every entry gets cls_value
  int cls_value = 0;
  double t_off;            // t0 - batch clock
  int nm_off;              // - batch measurement counter
  double* t_base;
  int* nm_base;
};

// Model constructors: uniform_velocity.cpp:50-56, uniform_acceleration.cpp:50-57,
// angular_rates.cpp:57-65, angular_velocities.cpp:51-56,73 + KalmanFilterInterface::init
// src/kalman.cpp:16-21 (x = x0, P = P0).  The unwrap memory starts at zero (the reference
// leaves meas_rpy_internal_ uninitialised: angular_rates.hpp:110, angular_velocities.hpp:127).
template <class M, typename T, int G, int LAYOUT>
__global__ void init_kernel(const InitArgs a) {
  using C = Cfg<M, T, G, LAYOUT>;
  constexpr int N = C::N;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n) return;
  const long slot = a.idx[e];
  T p7[7], x0[N];
#pragma unroll
  for (int c = 0; c < 7; ++c) p7[c] = (T)a.p0[e * 7 + c];
#pragma unroll
  for (int r = 0; r < N; ++r) x0[r] = 0;
  T v6[6], a6[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) { v6[c] = a.v0 ? (T)a.v0[e * 6 + c] : (T)0; a6[c] = a.a0 ? (T)a.a0[e * 6 + c] : (T)0; }
  if constexpr (M::TYPE == UNIFORM_VELOCITY) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { x0[c] = p7[c]; x0[3 + c] = v6[c]; }
  } else if constexpr (M::TYPE == UNIFORM_ACCELERATION) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { x0[c] = p7[c]; x0[3 + c] = v6[c]; x0[6 + c] = a6[c]; }
  } else {
    T p6[6];
    pose7_to_pose6(p7, p6);
#pragma unroll
    for (int c = 0; c < 6; ++c) { x0[c] = p6[c]; x0[6 + c] = v6[c]; }
    if constexpr (M::TYPE == ANGULAR_RATES) {
#pragma unroll
      for (int c = 0; c < 6; ++c) x0[12 + c] = a6[c];
    }
  }
  const double* P0 = a.P0 + (a.P0_index ? (long)a.P0_index[e] * N * N : (a.per_target_P0 ? e * N * N : 0));
  if (a.cls) a.cls[slot] = a.cls_of ? a.cls_of[e] : a.cls_value;
  for (int r = 0; r < N; ++r) {
    state_set<C, T>(a.rec, slot, r, N, x0[r]);
    for (int c = ((C::PK || C::SEPPK) ? r : 0); c < N; ++c) state_set<C, T>(a.rec, slot, r, c, (T)P0[r * N + c]);
  }
  if constexpr (M::ANGULAR) {
    for (int cc = 0; cc < 3; ++cc) *unwrap_ptr<C, T>(a.rec, slot, cc) = 0;
  }
  a.t_base[slot] = a.t_off;
  a.nm_base[slot] = a.nm_off;
}

// x [n][N] and P [n][N*N] (row-major, doubles) of the listed slots; one thread per (entry,row)
template <class M, typename T, int G, int LAYOUT>
__global__ void get_state_kernel(char* rec, const int* idx, long n, double* x_out, double* P_out) {
  using C = Cfg<M, T, G, LAYOUT>;
  constexpr int N = C::N;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n * N) return;
  const long e = tid / N;
  const int r = (int)(tid % N);
  const long slot = idx ? (long)idx[e] : e;
  if (x_out) x_out[e * N + r] = (double)state_get<C, T>(rec, slot, r, N);
  if (P_out)
    for (int c = 0; c < N; ++c) P_out[(e * N + r) * N + c] = (double)state_get<C, T>(rec, slot, r, c);
}

template <class M, typename T, int G, int LAYOUT>
__global__ void set_state_kernel(char* rec, const int* idx, long n, const double* x_in, const double* P_in,
                                 const double* uw_in) {
  using C = Cfg<M, T, G, LAYOUT>;
  constexpr int N = C::N;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n * N) return;
  const long e = tid / N;
  const int r = (int)(tid % N);
  const long slot = idx ? (long)idx[e] : e;
  if (x_in) state_set<C, T>(rec, slot, r, N, (T)x_in[e * N + r]);
  if (P_in)
    for (int c = ((C::PK || C::SEPPK) ? r : 0); c < N; ++c) state_set<C, T>(rec, slot, r, c, (T)P_in[(e * N + r) * N + c]);
  if constexpr (M::ANGULAR) {
    if (uw_in && r < 3) *unwrap_ptr<C, T>(rec, slot, r) = (T)uw_in[e * 3 + r];
  }
}

// copy the whole record of slot `src` over slot `dst` (erase = swap-with-last compaction)
template <class M, typename T, int G, int LAYOUT>
__global__ void move_record_kernel(char* rec, long src, long dst, double* t_base, int* nm_base, int* cls) {
  using C = Cfg<M, T, G, LAYOUT>;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= C::G * C::RW) return;
  const int i = tid / C::RW, w = tid % C::RW;
  const T v = *reinterpret_cast<T*>(rec + (src / C::TPW) * C::TILE_BYTES +
                                    record_word_offset<C, T>((int)(src % C::TPW) * C::G + i, w));
  *reinterpret_cast<T*>(rec + (dst / C::TPW) * C::TILE_BYTES +
                        record_word_offset<C, T>((int)(dst % C::TPW) * C::G + i, w)) = v;
  if (tid == 0) { t_base[dst] = t_base[src]; nm_base[dst] = nm_base[src]; if (cls) cls[dst] = cls[src]; }
}

// m independent moves at once (batched erase: survivors from the tail fill the holes); blockIdx.y = move
template <class M, typename T, int G, int LAYOUT>
__global__ void move_records_kernel(char* rec, const int* src, const int* dst, long m, double* t_base, int* nm_base, int* cls) {
  using C = Cfg<M, T, G, LAYOUT>;
  const long mv = blockIdx.y;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (mv >= m || tid >= C::G * C::RW) return;
  const long s = src[mv], d = dst[mv];
  const int i = tid / C::RW, w = tid % C::RW;
  const T v = *reinterpret_cast<T*>(rec + (s / C::TPW) * C::TILE_BYTES + record_word_offset<C, T>((int)(s % C::TPW) * C::G + i, w));
  *reinterpret_cast<T*>(rec + (d / C::TPW) * C::TILE_BYTES + record_word_offset<C, T>((int)(d % C::TPW) * C::G + i, w)) = v;
  if (tid == 0) { t_base[d] = t_base[s]; nm_base[d] = nm_base[s]; if (cls) cls[d] = cls[s]; }
}

// measurements: AoS doubles [n][7] (the reference's Vector7d rows) -> SoA T [7][ld]
template <typename T>
__global__ void pack_meas_kernel(const double* aos, long n, T* soa, long ld) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
#pragma unroll
  for (int c = 0; c < 7; ++c) soa[c * ld + e] = (T)aos[e * 7 + c];
}

struct OutArgs {
  char* rec;
  const int* idx;   // null: dense slots 0..n-1
  long n;
  double* pose;     // [n][7] or null
  double* twist;    // [n][6] or null
  double* acc;      // [n][6] or null
  int at_time;      // 0: current outputs; 1: extrapolated to t1 (getEstimated*(t1))
  double t1;        // absolute query time (at_time); NaN = each target's own time
  double t_acc;     // batch clock; target time = t_base[slot] + t_acc
  const double* t_base;
  int by_slot = 0;  // 1: output row = slot (scatter into a per-slot table) instead of the entry index
  // One-workgroup launches only (n <= kOutputsBlock): after every row has been written, store done_seq to *done_flag
  // (host-mapped memory).  A host thread that spins on the flag sees the rows without a stream synchronisation: the
  // one-target ABI's round trip is a launch and a PCIe write instead of a launch and the runtime's completion path.
  int* done_flag = nullptr;
  int done_seq = 0;
};
constexpr int kOutputsBlock = 128;

// Derived outputs of one target, from x only.
//  current : updateTargetState + getEstimatedPose()/Twist()/Acceleration()
//            (uniform_velocity.cpp:98-115, uniform_acceleration.cpp:101-118, angular_rates.cpp:117-138,
//             angular_velocities.cpp:153-169, target_interface.cpp:100-115, geometry.hpp:590-608)
//  at t1   : getEstimatedPose(t1)/Twist(t1)/Acceleration(t1)
//            (uniform_velocity.cpp:117-133, uniform_acceleration.cpp:120-136, angular_rates.cpp:140-157,
//             angular_velocities.cpp:171-184, target_interface.cpp:123-140)
template <class M, typename T>
__device__ __forceinline__ void derive_outputs(const T* x, bool at_time, T d, T* pose7, T* twist6, T* acc6) {
#pragma clang fp contract(off)   // the same roundings wherever this is inlined (see te_device_math.hpp)
  T R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  T pint[6] = {x[0], x[1], x[2], 0, 0, 0};  // pose_internal_
#pragma unroll
  for (int c = 0; c < 6; ++c) { twist6[c] = 0; acc6[c] = 0; }
  if constexpr (M::TYPE == UNIFORM_VELOCITY) {
#pragma unroll
    for (int c = 0; c < 3; ++c) twist6[c] = x[3 + c];
  } else if constexpr (M::TYPE == UNIFORM_ACCELERATION) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { twist6[c] = x[3 + c]; acc6[c] = x[6 + c]; }
  } else {
    T q[4];
    rpy_to_quat(x + 3, q);
    quat_to_rot(q, R);
    rot_to_rpy(R, pint + 3);
    if constexpr (M::TYPE == ANGULAR_RATES) {
      // twist.angular = EarBase(rotToRpy(R)) * rates, geometry.hpp:333-351
      T s_r, c_r, s_p, c_p;
      Mth<T>::sincos(pint[3], &s_r, &c_r);
      Mth<T>::sincos(pint[4], &s_p, &c_p);
#pragma unroll
      for (int c = 0; c < 3; ++c) twist6[c] = x[6 + c];
      twist6[3] = (1 * x[9] + 0 * x[10]) + (-s_p) * x[11];
      twist6[4] = (0 * x[9] + c_r * x[10]) + (c_p * s_r) * x[11];
      twist6[5] = (0 * x[9] + (-s_r) * x[10]) + (c_p * c_r) * x[11];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc6[c] = x[12 + c];
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) twist6[c] = x[6 + c];
    }
  }
  if (!at_time) {
    pose7[0] = x[0]; pose7[1] = x[1]; pose7[2] = x[2];
    rot_to_quat(R, pose7 + 3);
    return;
  }
  // extrapolation to t1 = t + d
  if constexpr (M::TYPE == UNIFORM_VELOCITY) {
#pragma unroll
    for (int c = 0; c < 3; ++c) pose7[c] = x[c] + twist6[c] * d;
    pose7[3] = 0; pose7[4] = 0; pose7[5] = 0; pose7[6] = 1;
  } else if constexpr (M::TYPE == UNIFORM_ACCELERATION) {
#pragma unroll
    for (int c = 0; c < 3; ++c) pose7[c] = x[c] + twist6[c] * d + (T)0.5 * acc6[c] * d * d;
    pose7[3] = 0; pose7[4] = 0; pose7[5] = 0; pose7[6] = 1;
#pragma unroll
    for (int c = 0; c < 6; ++c) twist6[c] = twist6[c] + acc6[c] * d;
  } else if constexpr (M::TYPE == ANGULAR_RATES) {
    T v6[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v6[c] = pint[c] + twist6[c] * d + (T)0.5 * acc6[c] * d * d;
    pose7[0] = v6[0]; pose7[1] = v6[1]; pose7[2] = v6[2];
    rpy_to_quat(v6 + 3, pose7 + 3);
    quat_normalize(pose7 + 3);
#pragma unroll
    for (int c = 0; c < 6; ++c) twist6[c] = twist6[c] + acc6[c] * d;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) pose7[c] = x[c] + twist6[c] * d;
    T q0[4];
    rpy_to_quat(pint + 3, q0);
    qtran_apply(d, twist6 + 3, q0, pose7 + 3);
    quat_normalize(pose7 + 3);
  }
}

template <class M, typename T, int G, int LAYOUT>
__global__ void outputs_kernel(const OutArgs a) {
  using C = Cfg<M, T, G, LAYOUT>;
  constexpr int N = C::N;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long slot = -1;
  if (e < a.n) slot = a.idx ? (long)a.idx[e] : e;   // negative: an entry that is not in this batch (device-resolved ids, id_resolve.hpp)
  if (slot >= 0) {
    T x[N];
#pragma unroll
    for (int r = 0; r < N; ++r) x[r] = state_get<C, T>(a.rec, slot, r, N);
    T d = 0;
    if (a.at_time) d = (a.t1 != a.t1) ? (T)0 : (T)(a.t1 - (a.t_base[slot] + a.t_acc));
    T pose7[7], twist6[6], acc6[6];
    derive_outputs<M, T>(x, a.at_time != 0, d, pose7, twist6, acc6);
    const long row = a.by_slot ? slot : e;
    if (a.pose) for (int c = 0; c < 7; ++c) a.pose[row * 7 + c] = (double)pose7[c];
    if (a.twist) for (int c = 0; c < 6; ++c) a.twist[row * 6 + c] = (double)twist6[c];
    if (a.acc) for (int c = 0; c < 6; ++c) a.acc[row * 6 + c] = (double)acc6[c];
  }
  if (a.done_flag != nullptr) {   // uniform; the launch is one workgroup (OutArgs::done_flag)
    __threadfence_system();       // this thread's rows are visible to the host ...
    __syncthreads();              // ... for every thread of the workgroup ...
    if (threadIdx.x == 0) __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the flag is
  }
}

struct IntersectArgs {
  char* rec;
  const int* idx;     // null: dense slots 0..n-1
  long n;
  double t1;          // absolute query time; NaN = each target's own time (t1 = t_)
  double origin[3];
  double radius;
  double t_acc;
  const double* t_base;
  double* delta;      // [n]: time to the first intersection after t1, or -1
  double* pose;       // [n][7] pose at t1 + delta (initPose if none), or null
};

// IntersectionSolver::getIntersectionTimeWithSphere / getIntersectionPoseWithSphere without the
// moving-average convergence gate (src/intersection_solver.cpp:42-104) for one target with state x:
// quartic in delta from the (p, v, a) extrapolated by dq = t1 - t_, smallest real root, pose at
// t1 + delta.  own = the query is at the target's own time (dq = 0, pose offset = delta itself).
// The values: delta (the reference's intersection time, -1 if none) and, if wanted, the pose at that time (identity if none).
template <class M, typename T>
__device__ __forceinline__ void sphere_query_values(const T* x, bool own, double t1, double t, const double* origin, double radius,
                                                    double& delta, double (&pose)[7], const bool want_pose
#ifdef TE_QUERY_PHASE_CLOCK
                                                    , long long* ts = nullptr
#endif
                                                    ) {
#pragma clang fp contract(off)   // fused query, intersect kernel: the same roundings (te_device_math.hpp)
#ifdef TE_QUERY_PHASE_CLOCK
#define TE_TS(i) do { if (ts) { __builtin_amdgcn_sched_barrier(0); ts[i] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define TE_TS(i) do {} while (0)
#endif
  T pose7[7], twist6[6], acc6[6];
  derive_outputs<M, T>(x, true, own ? (T)0 : (T)(t1 - t), pose7, twist6, acc6);
  const double px = (double)pose7[0] - origin[0], py = (double)pose7[1] - origin[1], pz = (double)pose7[2] - origin[2];
  const double vx = (double)twist6[0], vy = (double)twist6[1], vz = (double)twist6[2];
  const double ax = (double)acc6[0], ay = (double)acc6[1], az = (double)acc6[2];
  double c[5];
  c[4] = 0.25 * (ax * ax + ay * ay + az * az);
  c[3] = vx * ax + vy * ay + vz * az;
  c[2] = vx * vx + vy * vy + vz * vz + px * ax + py * ay + pz * az;
  c[1] = 2 * (px * vx + py * vy + pz * vz);
  c[0] = px * px + py * py + pz * pz - radius * radius;
  TE_TS(1);
  const double d = first_crossing_quartic(c);   // leftmost real root if >= 0, else -1
  TE_TS(2);
  delta = d;
  // (the array is taken by reference and the choice is a flag: a pointer that may be null made the caller's array escape into
  // scratch memory -- 64 bytes per lane, and a resident kernel with scratch is also limited by the process's scratch wave slots)
  if (want_pose) {
    pose[0] = pose[1] = pose[2] = pose[3] = pose[4] = pose[5] = 0; pose[6] = 1;
    if (d > -1) {
      derive_outputs<M, T>(x, true, own ? (T)d : (T)((d + t1) - t), pose7, twist6, acc6);
#pragma unroll
      for (int k = 0; k < 7; ++k) pose[k] = (double)pose7[k];
    }
  }
  TE_TS(3);
}

template <class M, typename T>
__device__ __forceinline__ void sphere_query(const T* x, bool own, double t1, double t, const double* origin, double radius,
                                             double* delta_out, double* pose_out /* [7] or null */) {
  double d, out[7];
  sphere_query_values<M, T>(x, own, t1, t, origin, radius, d, out, pose_out != nullptr);
  *delta_out = d;
  if (pose_out)
    for (int k = 0; k < 7; ++k) pose_out[k] = out[k];
}

template <class M, typename T, int G, int LAYOUT>
__global__ void intersect_kernel(const IntersectArgs a) {
  using C = Cfg<M, T, G, LAYOUT>;
  constexpr int N = C::N;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n) return;
  const long slot = a.idx ? (long)a.idx[e] : e;
  T x[N];
#pragma unroll
  for (int r = 0; r < N; ++r) x[r] = state_get<C, T>(a.rec, slot, r, N);
  // own-time query (t1 = NaN): the offsets are 0 and delta themselves, so a recorded launch (hipGraph)
  // does not depend on the batch clock; the reference's (delta + t1) - t_ differs by at most ulp(t_)
  const bool own = a.t1 != a.t1;
  const double t = own ? 0.0 : a.t_base[slot] + a.t_acc;
  sphere_query<M, T>(x, own, a.t1, t, a.origin, a.radius, &a.delta[e], a.pose ? &a.pose[e * 7] : nullptr);
}

}  // namespace te
