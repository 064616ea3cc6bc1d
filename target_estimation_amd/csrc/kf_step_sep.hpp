// kf_step_sep.hpp -- predict+update for axis-separable batches (LAYOUT_SEPARABLE).
//
// When Q, R and P0 have no entries between different axis groups (te_layout.hpp, group_of) the
// dense filter of src/kalman.cpp:84-95 / :129-140 never creates any: every product that would
// touch such an entry multiplies an exact zero.  The step below is therefore the SAME arithmetic
// as kf_step.hpp with the structurally zero terms left out:
//   linear models : one [p v (a)] filter with a scalar measurement per axis
//                   (uniform_velocity 3 x 2 states, uniform_acceleration 3 x 3, angular_rates 6 x 3);
//   EKF           : three [p v] filters for x, y, z and one 6-state [rpy, omega] EKF with the
//                   3-vector Euler-angle measurement (Jacobians of angular_velocities.cpp:116-124).
// Per target the record shrinks from n^2 + n to sum(group^2) + n words (angular_rates: 345 -> 75),
// which is what makes this worth a kernel: the step is HBM-bound.  Thread per target, everything
// in registers, no LDS; Q and R are read through the scalar cache (uniform addresses).
// The host only selects this layout after checking the matrices (target_manager.cpp,
// is_axis_separable); general matrices use the dense kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "kf_aux.hpp"
#include "kf_step.hpp"

namespace te {

// one [p v (a)] chain with a scalar position measurement: rows r0, r0+K, r0+2K of the model
template <int NB, typename T>
__device__ __forceinline__ void sep_linear_axis(T* x, T (&P)[NB][NB], const T (&Q)[NB][NB], T r_meas, T dt, bool has, T y) {
#pragma clang fp contract(off)  // only the explicit fma calls fuse: same roundings as the dense kernel
  using F = Mth<T>;
  const T hdt = (T)0.5 * dt * dt;
  // x^- = A x ; AP = A P (rows)
#pragma unroll
  for (int b = 0; b + 1 < NB; ++b) {
    x[b] = F::fma(dt, x[b + 1], x[b]);
    if (NB == 3 && b == 0) x[b] = F::fma(hdt, x[b + 2], x[b]);
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      T v = F::fma(dt, P[b + 1][c], P[b][c]);
      if (NB == 3 && b == 0) v = F::fma(hdt, P[b + 2][c], v);
      P[b][c] = v;
    }
  }
  // (AP) A^T (columns), + Q
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      T v = P[b][c];
      if (c + 1 < NB) {
        v = F::fma(dt, P[b][c + 1], v);
        if (NB == 3 && c == 0) v = F::fma(hdt, P[b][c + 2], v);
      }
      P[b][c] = v + Q[b][c];
    }
  if (!has) return;
  // S = P00 + R ; K = P[:,0] / S ; x += K (y - x0) ; P = (I - K C) P
  const T inv = (T)1 / (P[0][0] + r_meas);
  T Kg[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) Kg[b] = P[b][0] * inv;
  const T nu = y - x[0];
#pragma unroll
  for (int b = 0; b < NB; ++b) x[b] += Kg[b] * nu;
  T top[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) top[c] = P[0][c];
  const T d0 = (T)1 - Kg[0];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    P[0][c] = d0 * top[c];
#pragma unroll
    for (int b = 1; b < NB; ++b) {
      const T t = ((T)0 - Kg[b]) * top[c];
      P[b][c] = t + P[b][c];
    }
  }
}

// QUERY: the own-time sphere-intersection query of the target (kf_aux.hpp, sphere_query) runs on the
// posterior state while it is still in registers -- BASELINE.json configs[4], "per-step interception
// point fused on-GPU": one launch per tick instead of step + query.
// PERQR: Q and R come from the target's own parameter class (a table in HBM, per-lane loads) instead of the one
// shared pair read through the scalar cache.
// Register budget the allocator must respect (wavefronts per SIMD it has to leave room for; 1 = unconstrained).  The
// per-class angular_rates kernel in fp64 on packed group blocks sits two registers over the three-wave limit (170 of 168) when
// left alone; held to it, it parks 12-24 B per lane in scratch (grouped classes 162 -> 156 us per 10^6-target tick).
// (Resident kernels are never held to a register limit that makes them spill: a kernel with scratch is also limited by the
// scratch wave slots the runtime has sized for the process, which the occupancy query does not see -- the angular_rates fp64
// kernel, 6 registers over the two-wave limit, held to it: 114 000 targets started in a fresh process and did not in one that
// had run other kernels before.)
template <class M, typename T, int LAYOUT, bool PERQR, int LIVE = 0>
constexpr int sep_min_waves() {
  // (the resident uniform-acceleration fp64 and angular-rates fp32 kernels with the per-tick query: 170 / 171 registers when
  // scheduled freely, 3 wavefronts per SIMD need 168)
  return ((PERQR && M::TYPE == ANGULAR_RATES && sizeof(T) == 8 && LAYOUT == LAYOUT_SEPARABLE_PACKED) ||
          (LIVE == 2 && M::TYPE == UNIFORM_ACCELERATION && sizeof(T) == 8) || (LIVE == 2 && M::TYPE == ANGULAR_RATES && sizeof(T) == 4)) ? 3 : 1;
}

// Resident kernels keep the whole record in registers between ticks, and the capacity of the mode is the register file.  For
// the fp64 angular models that is not enough (angular_rates: 57 words = 114 registers of state under an fp64 atan2 / asin chain
// -> 262, one wavefront per SIMD, 49 152 targets): part of the record is PARKED in the wavefront's LDS between its uses --
// every lane its own column, word-interleaved (conflict-free ds_read/write_b64), read when its chain's turn comes and written
// back behind it.  live_park_p_chains / live_park_x_chains<M, T>() = how many [p v (a)] chains park their covariance words (P) /
// their state words (x); chosen so that the resident kernels fit three wavefronts per SIMD with at most 13 KB of LDS per
// wavefront (12 per CU).  -DTE_PARK_P / -DTE_PARK_X override them for experiments (one kernel at a time through
// tools/kres.py / tools/isa_liveness.py; the library is built without them).
template <class M, typename T> constexpr int live_park_p_chains() {
#ifdef TE_PARK_P
  return TE_PARK_P;
#else
  return sizeof(T) == 8 && M::TYPE == ANGULAR_RATES ? 4 : sizeof(T) == 8 && M::TYPE == ANGULAR_VELOCITIES ? 3 : 0;
#endif
}
template <class M, typename T> constexpr int live_park_x_chains() {
#ifdef TE_PARK_X
  return TE_PARK_X;
#else
  return sizeof(T) == 8 && M::TYPE == ANGULAR_VELOCITIES ? 3 : 0;
#endif
}
template <class C, class M, typename T> struct LivePark {
  static constexpr int NLIN = M::EKF ? 3 : C::K, LB = M::EKF ? 2 : C::NB, STRIDE = M::EKF ? 6 : C::K;
  struct Table { int v[C::RW]; int count; };
  static constexpr Table make() {
    Table t{};
    for (int w = 0; w < C::RW; ++w) t.v[w] = -1;
    int k = 0;
    for (int i = 0; i < NLIN && i < live_park_p_chains<M, T>(); ++i)
      for (int b = 0; b < LB; ++b)
        for (int c = b; c < LB; ++c) t.v[C::PWORD.v[i + STRIDE * b][i + STRIDE * c]] = k++;
    for (int i = 0; i < NLIN && i < live_park_x_chains<M, T>(); ++i)
      for (int b = 0; b < LB; ++b) t.v[C::X_OFF + i + STRIDE * b] = k++;
    t.count = k;
    return t;
  }
  static constexpr Table SLOT = make();
};

// LIVE: a resident launch (StepArgs::live_*): the tick loop of FUSED with a wait for the host's doorbell in front of every tick
// and a progress word behind it (1), optionally with the per-tick sphere query and pose output (2: more registers, so fewer
// resident targets).  Same arithmetic per tick, same results as single ticks.
// AB: an A -> B tick (StepArgs::rec_out), its own instantiation (see kf_step_kernel).
// The step of one wavefront's targets: `wg` = index of the wavefront among those of the launch (of the BATCH, in a population
// launch: kf_step_population_kernel below), lane = its lane.
template <class M, typename T, int LAYOUT, bool INDEXED, bool FUSED = false, bool QUERY = false, bool PERQR = false, int LIVE = 0, bool AB = false>
__device__ __forceinline__ void sep_step_wave(const StepArgs<T>& a, long wg, const int lane) {
  static_assert(!AB || (!INDEXED && !FUSED && !QUERY && !LIVE), "A -> B ticks are dense single-tick launches without the fused query");
  static_assert(!(QUERY && (INDEXED || FUSED)), "the fused query is for dense single-tick launches");
  static_assert(!(PERQR && (FUSED || QUERY)), "per-class Q/R: single-tick launches without the fused query");
  static_assert(!LIVE || (FUSED && !INDEXED && !QUERY && !PERQR), "live launches are dense multi-tick launches");
  using C = Cfg<M, T, 1, LAYOUT>;
  static_assert(C::SEP, "separable layouts only");
  constexpr int N = C::N, K = C::K, NB = C::NB, TPW = C::TPW;
  using F = Mth<T>;

  if constexpr (LIVE) {   // the workgroup behind the last worker is the relay between the host's words and the device's
    const long workers = (a.n + TPW - 1) / TPW;
    if (wg == workers) {
      live_relay(a.live_posted, a.live_mirror, a.live_progress, a.live_done, workers, a.live_spin_limit, a.live_idle_ticks, lane, a.live_flags);
      return;
    }
  }
#ifdef TE_QUERY_PHASE_CLOCK
  const long long ts_begin = (long long)__builtin_readcyclecounter();
#endif
  if (wg * TPW >= a.n) return;
  const long wave_id = wg;                               // LIVE: index of this wavefront's progress word
  if (a.reverse) wg = (a.n + TPW - 1) / TPW - 1 - wg;   // zig-zag traversal (StepArgs::reverse)
  const long entry = wg * TPW + lane;
  bool valid = entry < a.n;
  long tile;
  int lt;
  long slot_of = entry;
  if constexpr (INDEXED) {
    long slot = valid ? (long)a.idx[entry] : -1;   // a negative slot = "skip this entry"
    valid = slot >= 0;
    if (!valid) slot = 0;
    slot_of = slot;
    tile = slot / TPW;
    lt = (int)(slot % TPW);
  } else {
    tile = wg;
    lt = lane;
  }
  char* tb = a.rec + tile * C::TILE_BYTES;
  T mem[C::RW];
  // LIVE: the parked part of the record (LivePark) lives in the wavefront's LDS for the whole session
  using Park = LivePark<C, M, T>;
  constexpr int NPARK = (LIVE != 0 && C::SEPPK) ? Park::SLOT.count : 0;
  __shared__ T park_lds[NPARK > 0 ? NPARK * 64 : 1];
  __shared__ double qpose_lds[LIVE == 2 ? 64 * 7 : 1];   // the per-tick query's pose rows on their way out (LIVE == 2, below)
  if constexpr (NPARK > 0) {
    // the record goes to its two homes chunk by chunk, a few loads in flight at a time: loaded whole (load_record) it would
    // itself be the register peak of the kernel.  Once per session: its speed does not matter.
    static_assert(C::REM2 == 0, "parked records: 16-byte chunks and a one-word tail");
    using V = typename Vec16<T>::type;
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
      V v{};
      if (valid) v = *reinterpret_cast<const V*>(tb + (long)c * C::LPT * 16 + (long)lt * 16);
      T w2[C::VW];
      __builtin_memcpy(w2, &v, 16);
#pragma unroll
      for (int k = 0; k < C::VW; ++k) {
        const int w = c * C::VW + k;
        if (Park::SLOT.v[w] >= 0) park_lds[Park::SLOT.v[w] * 64 + lane] = w2[k];
        else mem[w] = w2[k];
      }
      if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (C::REM1) {
      const T v = valid ? *reinterpret_cast<const T*>(tb + C::TAIL1_OFF + (long)lt * (long)sizeof(T)) : T(0);
      if (Park::SLOT.v[C::RW - 1] >= 0) park_lds[Park::SLOT.v[C::RW - 1] * 64 + lane] = v;
      else mem[C::RW - 1] = v;
    }
  } else if (valid) {
    load_record<C, T>(tb, lt, mem);
  } else {
#pragma unroll
    for (int w = 0; w < C::RW; ++w) mem[w] = 0;
  }
  auto RD = [&](int w) -> T {
    if (NPARK > 0 && Park::SLOT.v[w] >= 0) return park_lds[Park::SLOT.v[w] * 64 + lane];
    return mem[w];
  };
  auto WR = [&](int w, T v) {
    if (NPARK > 0 && Park::SLOT.v[w] >= 0) park_lds[Park::SLOT.v[w] * 64 + lane] = v;
    else mem[w] = v;
  };
  double dtd = a.dt;
  if constexpr (INDEXED) {
    if (a.dt_per && valid) dtd = a.dt_per[entry];
  }
  const T dt = (T)dtd;
  // The one (Q, R) row of a one-class batch, through the CONSTANT address space: uniform and never written while a step kernel
  // runs, so every read is a scalar load wherever it sits.  As a plain global pointer it stops being one behind the resident
  // loop's atomics (the compiler can no longer prove that nothing clobbers it): the fp64 angular kernels, whose 60 words do not
  // fit the scalar registers ahead of the loop, fetched them with vector loads every tick -- 120 vector registers.
  typedef const T __attribute__((address_space(4))) ConstT;
  ConstT* Qm = (ConstT*)a.qr;
  // PERQR: when every target of the wavefront belongs to ONE class -- the usual case, classes arrive in runs -- the row is
  // read through the scalar cache like the single (Q, R) of a one-class batch (Qu, a uniform address); only a wavefront
  // that mixes classes pays for per-lane gathers of its rows.
  const T* Qu = a.qr;
  bool cls_uniform = false;
  int cls_own = 0;
  if constexpr (PERQR) {
    if (valid) cls_own = a.cls[slot_of];
    const int cls_first = __builtin_amdgcn_readfirstlane(cls_own);
    cls_uniform = __all(!valid || cls_own == cls_first) != 0;   // (a ragged wave whose lane 0 is idle may take the per-lane path: same results)
    Qu = a.qr + (long)cls_first * C::QR_WORDS;
  }
  // the per-lane row, formed where it is needed (one register for the class instead of two for the address)
  auto lane_row = [&]() -> const T* {
    const T* q = a.qr + (long)cls_own * C::QR_WORDS;
    if constexpr ((C::QR_WORDS * sizeof(T)) % 16 == 0) q = static_cast<const T*>(__builtin_assume_aligned(q, 16));   // rows are whole 16-byte chunks: wide loads
    return q;
  };
  if constexpr (!PERQR) (void)lane_row;
  // the [p v (a)] chains.  Linear models: K of them, rows {i, i+K, i+2K}.  EKF: x, y, z with rows
  // {i, i+6} (position, velocity), plus the 6-state attitude group (rows 3..5, 9..11).
  constexpr int NLIN = M::EKF ? 3 : K;
  constexpr int LB = M::EKF ? 2 : NB;         // states per chain
  constexpr int STRIDE = M::EKF ? 6 : K;      // row distance between the states of a chain
  constexpr int GRA[6] = {3, 4, 5, 9, 10, 11};
  // Q and R of every chain are uniform (scalar loads): when they fit the scalar registers they are requested
  // once, here, behind the record loads, instead of chain by chain with a wait each (a serial chain of
  // scalar-cache round trips that a small, latency-bound batch feels directly).
  constexpr int QR_NEED = NLIN * LB * LB + NLIN + (M::EKF ? 36 + 9 : 0);
  constexpr bool HOIST_QR = !PERQR && QR_NEED * (int)sizeof(T) <= 256;
  T Qlin[HOIST_QR ? NLIN : 1][LB][LB], Rlin[HOIST_QR ? NLIN : 1], Qatt[HOIST_QR && M::EKF ? 6 : 1][6], Ratt[HOIST_QR && M::EKF ? 3 : 1][3];
  if constexpr (HOIST_QR) {
#pragma unroll
    for (int i = 0; i < NLIN; ++i) {
#pragma unroll
      for (int b = 0; b < LB; ++b)
#pragma unroll
        for (int c = 0; c < LB; ++c) Qlin[i][b][c] = Qm[C::QWORD.v[i + STRIDE * b][i + STRIDE * c]];
      Rlin[i] = Qm[C::RWORD.v[i][i]];
    }
    if constexpr (M::EKF) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) Qatt[r][c] = Qm[C::QWORD.v[GRA[r]][GRA[c]]];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Ratt[r][c] = Qm[C::RWORD.v[3 + r][3 + c]];
    }
  }
  int n_has = 0;
  const int n_ticks = FUSED ? a.n_ticks : 1;
  long long live_seen = 0;
  for (int tick = 0; tick < n_ticks; ++tick) {
  long slot_tick = tick;
  if constexpr (LIVE) {
    // (the worker's own limit is a backstop far behind the relay's: a relay round is a PCIe read + a scan, a worker poll an L2 hit)
    if (!live_wait_tick(a.live_mirror + (wave_id / kLiveGroup) * kLiveMirrorStride, a.live_posted, (long long)tick + 1, a.live_spin_limit < 0x07000000u ? 32u * a.live_spin_limit + 10000000u : 0xffffffffu, live_seen, lane, a.live_flags)) break;
    slot_tick = (a.live_first + tick) % a.live_ring;
  }
  const T* meas_t = a.meas ? a.meas + slot_tick * a.tick_stride : nullptr;
  const unsigned char* has_t = a.has_meas ? a.has_meas + slot_tick * a.has_stride : nullptr;
  // Every measurement word of the tick is requested up front, right behind the record loads and regardless
  // of the mask: with a cache-resident state they are the only operands that come from HBM itself, and one
  // round trip for all of them replaces one per axis.
  constexpr int MW = M::ANGULAR ? 7 : 3;
  T ymeas[MW];
#pragma unroll
  for (int c = 0; c < MW; ++c) ymeas[c] = 0;
  unsigned char hmask = 1;
  if (valid && meas_t != nullptr) {
#pragma unroll
    for (int c = 0; c < MW; ++c) {
      if constexpr (LIVE) ymeas[c] = load_meas_live(&meas_t[(long)c * a.meas_ld + entry]);   // (no per-load run-time choice here: it would get a branch and a wait per word)
      else ymeas[c] = load_meas(&meas_t[(long)c * a.meas_ld + entry], a.nt_meas);
    }
    if (has_t != nullptr) {
      if constexpr (LIVE) hmask = (unsigned char)__hip_atomic_load(&has_t[entry], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else hmask = has_t[entry];
    }
  }
  const bool has = valid && meas_t != nullptr && hmask != 0;
  n_has += has ? 1 : 0;
  if constexpr (HOIST_QR) {
    // pin the hoisted Q / R values into scalar registers here, i.e. wait for their loads now, while the record
    // and measurement loads are still in flight (otherwise the compiler sinks them back next to their uses)
    if (tick == 0) {
#pragma unroll
      for (int i = 0; i < NLIN; ++i) {
#pragma unroll
        for (int b = 0; b < LB; ++b)
#pragma unroll
          for (int c = 0; c < LB; ++c) asm volatile("" : "+s"(Qlin[i][b][c]));
        asm volatile("" : "+s"(Rlin[i]));
      }
      if constexpr (M::EKF) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c < 6; ++c) asm volatile("" : "+s"(Qatt[r][c]));
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) asm volatile("" : "+s"(Ratt[r][c]));
      }
    }
  }

  T mrpy[3] = {0, 0, 0};
  if constexpr (M::ANGULAR) {
    if (has) {
      T q[4];
      q[0] = ymeas[3];
      q[1] = ymeas[4];
      q[2] = ymeas[5];
      q[3] = ymeas[6];
      quat_normalize(q);
      quat_to_rpy<T, (LIVE != 0 && sizeof(T) == 8)>(q, mrpy);
    }
  }
#define XW_(r) mem[C::X_OFF + (r)]
#define UWW_(s) mem[C::UW_OFF + (s)]
#define XR_(r) RD(C::X_OFF + (r))
#define XS_(r, v) WR(C::X_OFF + (r), (v))

  // ---- the [p v (a)] chains
#pragma unroll
  for (int i = 0; i < NLIN; ++i) {
    T xs[LB], Pb[LB][LB], Qb[LB][LB];
    T r_meas = T(0);   // (every path assigns it; the per-class branches hide that from the compiler)
#pragma unroll
    for (int b = 0; b < LB; ++b) {
      xs[b] = XR_(i + STRIDE * b);
#pragma unroll
      for (int c = 0; c < LB; ++c) Pb[b][c] = RD(C::PWORD.v[i + STRIDE * b][i + STRIDE * c]);
    }
    if constexpr (HOIST_QR) {
#pragma unroll
      for (int b = 0; b < LB; ++b)
#pragma unroll
        for (int c = 0; c < LB; ++c) Qb[b][c] = Qlin[i][b][c];
      r_meas = Rlin[i];
    } else {
      if constexpr (PERQR) {
        const T* Qsrc = lane_row();
        if (cls_uniform) {   // wave-uniform branch: scalar loads from the one row
#pragma unroll
          for (int b = 0; b < LB; ++b)
#pragma unroll
            for (int c = 0; c < LB; ++c) Qb[b][c] = Qu[C::QWORD.v[i + STRIDE * b][i + STRIDE * c]];
          r_meas = Qu[C::RWORD.v[i][i]];
          Qsrc = nullptr;
        }
        if (Qsrc != nullptr) {
#pragma unroll
          for (int b = 0; b < LB; ++b)
#pragma unroll
            for (int c = 0; c < LB; ++c) Qb[b][c] = Qsrc[C::QWORD.v[i + STRIDE * b][i + STRIDE * c]];
          r_meas = Qsrc[C::RWORD.v[i][i]];
        }
      } else {
#pragma unroll
        for (int b = 0; b < LB; ++b)
#pragma unroll
          for (int c = 0; c < LB; ++c) Qb[b][c] = Qm[C::QWORD.v[i + STRIDE * b][i + STRIDE * c]];
        r_meas = Qm[C::RWORD.v[i][i]];
      }
    }
    T y = 0;
    if (has) {
      if (!M::ANGULAR || i < 3) {
        y = ymeas[i];
      } else {
        y = unwrap_angle(UWW_(i - 3), mrpy[i - 3]);   // angular_rates.cpp:85-88
        UWW_(i - 3) = y;
      }
    }
    sep_linear_axis<LB, T>(xs, Pb, Qb, r_meas, dt, has, y);
#pragma unroll
    for (int b = 0; b < LB; ++b) {
      XS_(i + STRIDE * b, xs[b]);
#pragma unroll
      for (int c = (C::SEPPK ? b : 0); c < LB; ++c) WR(C::PWORD.v[i + STRIDE * b][i + STRIDE * c], Pb[b][c]);
    }
    if constexpr (NPARK > 0) __builtin_amdgcn_sched_barrier(0);   // a parked chain's words are read when its turn comes, not ahead of it
  }

  // ---- EKF attitude group: local rows 0..2 = rpy (global 3..5), 3..5 = omega (global 9..11)
  if constexpr (M::EKF) {
    constexpr int GR[6] = {3, 4, 5, 9, 10, 11};
    T xr[6], Pr[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      xr[r] = XW_(GR[r]);
#pragma unroll
      for (int c = 0; c < 6; ++c) Pr[r][c] = mem[C::PWORD.v[GR[r]][GR[c]]];
    }
    T s_r, c_r, s_p, c_p;
    F::sincos(xr[0], &s_r, &c_r);
    F::sincos(xr[1], &s_p, &c_p);
    const T wy = xr[4], wz = xr[5];
    // geometry.hpp:394-426 (Jacobians at the previous posterior), :359-374 (EarBaseInv)
    T Jr[3][3], Jw[3][3], Ei[3][3];
    Jr[0][0] = (dt * (wy * c_r * s_p - wz * s_p * s_r)) / c_p + 1;
    Jr[0][1] = (dt * (wz * c_r + wy * s_r)) / (c_p * c_p);
    Jr[0][2] = 0;
    Jr[1][0] = -dt * (wz * c_r + wy * s_r);
    Jr[1][1] = 1;
    Jr[1][2] = 0;
    Jr[2][0] = (dt * (wy * c_r - wz * s_r)) / c_p;
    Jr[2][1] = (dt * s_p * (wz * c_r + wy * s_r)) / (c_p * c_p);
    Jr[2][2] = 1;
    Jw[0][0] = dt; Jw[0][1] = (dt * s_p * s_r) / c_p; Jw[0][2] = (dt * c_r * s_p) / c_p;
    Jw[1][0] = 0;  Jw[1][1] = dt * c_r;               Jw[1][2] = -dt * s_r;
    Jw[2][0] = 0;  Jw[2][1] = (dt * s_r) / c_p;       Jw[2][2] = (dt * c_r) / c_p;
    Ei[0][0] = 1; Ei[0][1] = (s_p * s_r) / c_p; Ei[0][2] = (c_r * s_p) / c_p;
    Ei[1][0] = 0; Ei[1][1] = c_r;               Ei[1][2] = -s_r;
    Ei[2][0] = 0; Ei[2][1] = s_r / c_p;         Ei[2][2] = c_r / c_p;
    // x^- = f(x): rpy += dt * EarBaseInv(rpy) * omega  (angular_velocities.cpp:137)
    {
      T nr[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        T acc = (dt * Ei[c][0]) * xr[3];
        acc = F::fma(dt * Ei[c][1], xr[4], acc);
        acc = F::fma(dt * Ei[c][2], xr[5], acc);
        nr[c] = xr[c] + acc;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) xr[c] = nr[c];
    }
    // A P: rows 0..2 = Jr P[0:3,:] + Jw P[3:6,:]; rows 3..5 unchanged
    {
      T np[3][6];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          T v = Jr[r][0] * Pr[0][c];
          v = F::fma(Jr[r][1], Pr[1][c], v);
          v = F::fma(Jr[r][2], Pr[2][c], v);
          v = F::fma(Jw[r][0], Pr[3][c], v);
          v = F::fma(Jw[r][1], Pr[4][c], v);
          v = F::fma(Jw[r][2], Pr[5][c], v);
          np[r][c] = v;
        }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) Pr[r][c] = np[r][c];
    }
    // (A P) A^T: columns 0..2, then + Q
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      T nw[3];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        T v = Pr[r][0] * Jr[cc][0];
        v = F::fma(Pr[r][1], Jr[cc][1], v);
        v = F::fma(Pr[r][2], Jr[cc][2], v);
        v = F::fma(Pr[r][3], Jw[cc][0], v);
        v = F::fma(Pr[r][4], Jw[cc][1], v);
        v = F::fma(Pr[r][5], Jw[cc][2], v);
        nw[cc] = v;
      }
      T qrow[6];
      if constexpr (HOIST_QR) {
#pragma unroll
        for (int c = 0; c < 6; ++c) qrow[c] = Qatt[r][c];
      } else {
        if constexpr (PERQR) {
          if (cls_uniform) {
#pragma unroll
            for (int c = 0; c < 6; ++c) qrow[c] = Qu[C::QWORD.v[GR[r]][GR[c]]];
          } else {
            const T* Qsrc = lane_row();
#pragma unroll
            for (int c = 0; c < 6; ++c) qrow[c] = Qsrc[C::QWORD.v[GR[r]][GR[c]]];
          }
        } else {
#pragma unroll
          for (int c = 0; c < 6; ++c) qrow[c] = Qm[C::QWORD.v[GR[r]][GR[c]]];
        }
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) Pr[r][c] = (c < 3 ? nw[c < 3 ? c : 0] : Pr[r][c]) + qrow[c];
    }
    if (has) {
      T S[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          T rrc;
          if constexpr (HOIST_QR) rrc = Ratt[r][c];
          else if constexpr (PERQR) rrc = cls_uniform ? Qu[C::RWORD.v[3 + r][3 + c]] : lane_row()[C::RWORD.v[3 + r][3 + c]];
          else rrc = Qm[C::RWORD.v[3 + r][3 + c]];
          S[r][c] = Pr[r][c] + rrc;
        }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const T inv = (T)1 / S[p][p];
        S[p][p] = 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) S[p][c] *= inv;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (r == p) continue;
          const T f = S[r][p];
          S[r][p] = 0;
#pragma unroll
          for (int c = 0; c < 3; ++c) S[r][c] = F::fma(-f, S[p][c], S[r][c]);
        }
      }
      T nu[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const T y = unwrap_angle(UWW_(c), mrpy[c]);   // angular_velocities.cpp:93-96
        UWW_(c) = y;
        nu[c] = y - xr[c];
      }
      T Kg[6][3];
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int r = 0; r < 6; ++r) Kg[r][l] = (c == 0) ? Pr[r][0] * S[0][l] : F::fma(Pr[r][c], S[c][l], Kg[r][l]);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        T acc = Kg[r][0] * nu[0];
        acc = F::fma(Kg[r][1], nu[1], acc);
        acc = F::fma(Kg[r][2], nu[2], acc);
        xr[r] += acc;
      }
      T D[6][3];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) D[r][j] = ((r == j) ? (T)1 : (T)0) - Kg[r][j];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const T t0 = Pr[0][c], t1 = Pr[1][c], t2 = Pr[2][c];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          T acc = D[r][0] * t0;
          acc = F::fma(D[r][1], t1, acc);
          acc = F::fma(D[r][2], t2, acc);
          Pr[r][c] = (r < 3) ? acc : acc + Pr[r][c];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      XW_(GR[r]) = xr[r];
#pragma unroll
      for (int c = (C::SEPPK ? r : 0); c < 6; ++c) mem[C::PWORD.v[GR[r]][GR[c]]] = Pr[r][c];
    }
  }

  if constexpr (LIVE == 2) {
    // the own-time sphere query of every target after every tick (BASELINE configs[4]), on the posterior still in registers,
    // and / or the tick's poses for a consumer outside the kernel.  Run-time choices inside the LIVE == 2 variant only: the
    // query's fp64 quartic and the pose derivation cost 30 - 60 registers, i.e. resident capacity, which a plain session
    // (LIVE == 1) keeps.
    if (a.q_delta != nullptr && valid) {
      T xq[N];
#pragma unroll
      for (int r = 0; r < N; ++r) xq[r] = XR_(r);
      // Written THROUGH the caches like the poses below: a consumer on another stream or a copy engine that reads them once
      // `done` has reached the tick must see this tick's results, not what an XCD's L2 still holds.  The pose rows are
      // [target][7]: a lane's own row is 56 bytes at a stride of 56, and written through as such every store is a partial line
      // straight to memory (configs[4]'s share: 4.9 -> 10.5 us per tick).  The wavefront's 64 rows are one contiguous run of
      // 3584 bytes, so they go through its LDS and leave as seven full 512-byte stores.
      double qd, qp[7];
      sphere_query_values<M, T>(xq, true, 0.0, 0.0, a.q_origin, a.q_radius, qd, qp, a.q_pose != nullptr);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(&a.q_delta[entry]), (unsigned long long)__double_as_longlong(qd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.q_pose != nullptr) {
#pragma unroll
        for (int k = 0; k < 7; ++k) qpose_lds[lane * 7 + k] = qp[k];
      }
    }
    if (a.q_delta != nullptr && a.q_pose != nullptr) {   // (uniform) every lane takes part in the transposed store
      wave_lds_fence();
      const long first = wg * TPW;                                        // first target of this wavefront
      const long words = (a.n - first < TPW ? a.n - first : (long)TPW) * 7;   // its rows, as one run of words
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int f = k * 64 + lane;
        if (f < words)
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(&a.q_pose[first * 7 + f]), (unsigned long long)__double_as_longlong(qpose_lds[f]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      wave_lds_fence();
    }
    if (a.live_pose != nullptr) {   // the tick's estimated poses for a consumer outside the kernel (StepArgs::live_pose)
      if (valid) {
        T xq[N], pose7[7], twist6[6], acc6[6];
#pragma unroll
        for (int r = 0; r < N; ++r) xq[r] = XR_(r);
        derive_outputs<M, T>(xq, false, (T)0, pose7, twist6, acc6);
#pragma unroll
        for (int c = 0; c < 7; ++c)   // system-scope stores: written through to memory, 512 contiguous bytes per wavefront and row
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(&a.live_pose[(long)c * a.live_pose_ld + entry]),
                             (unsigned long long)__double_as_longlong((double)pose7[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tick's outputs have left before the progress word says so
  }
  if constexpr (LIVE) {
    // tick `tick` is done (state in registers): a word in device memory for the relay
    if (lane == 0) {
      if (a.live_flags & kLiveWorkerRel) __hip_atomic_store(&a.live_progress[wave_id], tick + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(&a.live_progress[wave_id], tick + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (see kf_step.hpp, "Ordering of the hand-offs")
    }
  }
  }  // tick loop
  if (valid) {
    if constexpr (NPARK > 0) {   // the reverse of the session's first lines: chunk by chunk from the record's two homes
      using V = typename Vec16<T>::type;
#pragma unroll
      for (int c = 0; c < C::NC; ++c) {
        T w2[C::VW];
#pragma unroll
        for (int k = 0; k < C::VW; ++k) {
          const int w = c * C::VW + k;
          w2[k] = Park::SLOT.v[w] >= 0 ? park_lds[Park::SLOT.v[w] * 64 + lane] : mem[w];
        }
        V v;
        __builtin_memcpy(&v, w2, 16);
        *reinterpret_cast<V*>(tb + (long)c * C::LPT * 16 + (long)lt * 16) = v;
        if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (C::REM1)
        *reinterpret_cast<T*>(tb + C::TAIL1_OFF + (long)lt * (long)sizeof(T)) =
            Park::SLOT.v[C::RW - 1] >= 0 ? park_lds[Park::SLOT.v[C::RW - 1] * 64 + lane] : mem[C::RW - 1];
    }
    else if constexpr (AB) store_record<C, T, false, true>(a.rec_out + tile * C::TILE_BYTES, lt, mem);   // A -> B tick (StepArgs::rec_out)
    else store_record<C, T>(tb, lt, mem);
    if constexpr (QUERY) {
      T xq[N];
#pragma unroll
      for (int r = 0; r < N; ++r) xq[r] = XW_(r);
      double qd, qp[7];
#ifdef TE_QUERY_PHASE_CLOCK
      long long ts[8];
      __builtin_amdgcn_sched_barrier(0); ts[0] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
      sphere_query_values<M, T>(xq, true, 0.0, 0.0, a.q_origin, a.q_radius, qd, qp, a.q_pose != nullptr, ts);
      // lanes 0..4 of every wavefront report: cycles from the wavefront's first instruction to the head of the query, then the query's phases
      qd = lane == 0 ? (double)(ts[0] - ts_begin) : lane == 1 ? (double)(ts[1] - ts[0]) : lane == 2 ? (double)(ts[2] - ts[1]) : lane == 3 ? (double)(ts[3] - ts[2]) : qd;
#else
      sphere_query_values<M, T>(xq, true, 0.0, 0.0, a.q_origin, a.q_radius, qd, qp, a.q_pose != nullptr);
#endif
      a.q_delta[entry] = qd;
      if (a.q_pose != nullptr) {
#pragma unroll
        for (int k = 0; k < 7; ++k) a.q_pose[entry * 7 + k] = qp[k];
      }
    }
    if constexpr (INDEXED) {
      const long slot = slot_of;
      a.t_base[slot] += dtd * n_ticks;
      a.nm_base[slot] += n_has;
    } else {
      if (a.has_meas != nullptr) a.nm_base[entry] += n_has;
    }
  }
  if constexpr (INDEXED) {
    if (a.o_pose != nullptr) {   // uniform: the getter table and the completion flag in the same launch (StepArgs::o_pose)
      if (valid) {
        T xq[N];
#pragma unroll
        for (int r = 0; r < N; ++r) xq[r] = XW_(r);
        write_outputs_row<M, T>(xq, slot_of, a.o_pose, a.o_twist, a.o_acc);
      }
      signal_done(a.done_flag, a.done_seq, lane, a.done_count, a.n, TPW);
    }
  }
#undef XW_
#undef UWW_
#undef XR_
#undef XS_
}

template <class M, typename T, int LAYOUT, bool INDEXED, bool FUSED = false, bool QUERY = false, bool PERQR = false, int LIVE = 0, bool AB = false>
__global__ void __launch_bounds__(256, (sep_min_waves<M, T, LAYOUT, PERQR, LIVE>())) kf_step_sep_kernel(const StepArgs<T> a) {
  sep_step_wave<M, T, LAYOUT, INDEXED, FUSED, QUERY, PERQR, LIVE, AB>(a, (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), (int)(threadIdx.x & 63));
}

// ---- one launch for the whole population of a manager ------------------------------------------------------------------------
// A manager with several motion models has one batch per model; their ticks are independent, and as separate launches they only
// overlap if the runtime happens to put their streams on hardware queues that the device runs side by side.  That is not a
// property to build on: measured (profiles/r04_queue_pipes.txt), the two branches of a recorded tick ran 2-3x slower as soon as
// the process owned a fifth hardware queue, whoever created it.  Here the tick of ALL batches is one grid: the first end[0]
// workgroups step the angular-rates batch, the next ones the angular-velocities batch, and so on in the order of the
// reference's enum (target_manager.hpp:38) -- the heaviest model first (measured at configs[4]'s share: light parts first 13.6 -> 15.2 us
// per tick, the two parts' workgroups alternating 14.9; gpurun_out/r4popgrid).  Each part is exactly kf_step_sep_kernel's arithmetic
// (sep_step_wave) on its own StepArgs; an absent model has end[k] == end[k - 1].
template <typename T>
struct PopulationArgs {
  StepArgs<T> part[4];     // indexed by ModelType
  unsigned end[4];         // first workgroup BEHIND part k
  int reverse_blocks;      // zig-zag over the whole population: walk the workgroups (parts and their tiles) last to first
};

template <typename T, bool QUERY, bool AB>
__global__ void __launch_bounds__(256) kf_step_population_kernel(const PopulationArgs<T> p) {
  const int lane = (int)(threadIdx.x & 63);
  const unsigned wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
  unsigned b = blockIdx.x;
  if (p.reverse_blocks) b = gridDim.x - 1 - b;
  constexpr int L = LAYOUT_SEPARABLE_PACKED;
  if (b < p.end[0]) sep_step_wave<ModelAR, T, L, false, false, QUERY, false, 0, AB>(p.part[0], (long)b * wpb + wave, lane);
  else if (b < p.end[1]) sep_step_wave<ModelAV, T, L, false, false, QUERY, false, 0, AB>(p.part[1], (long)(b - p.end[0]) * wpb + wave, lane);
  else if (b < p.end[2]) sep_step_wave<ModelUA, T, L, false, false, QUERY, false, 0, AB>(p.part[2], (long)(b - p.end[1]) * wpb + wave, lane);
  else sep_step_wave<ModelUV, T, L, false, false, QUERY, false, 0, AB>(p.part[3], (long)(b - p.end[2]) * wpb + wave, lane);
}

}  // namespace te
