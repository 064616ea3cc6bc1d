// te_layout.hpp -- model traits and the HBM record layout of the batched Kalman store.
//
// One "batch" holds all targets of one motion model / parameter set / precision.  A target's
// filter state (x, P, and for the angular models the 3-word unwrap memory) is split over G
// cooperating lanes ("row-residue ownership": lane i of a target's group owns rows
// r = i + G*q of x and P).  With G | K (K = size of one [p|v|a] block = measurement size m)
// rows r, r+K, r+2K live in the same lane, so the banded transition of the linear models
// (reference: src/types/uniform_velocity.cpp:90-96, uniform_acceleration.cpp:91-99,
// angular_rates.cpp:108-115) is applied entirely in registers.
//
// HBM layout (AoSoA, "lane records"): a tile = the TPW = 64/G targets one wavefront processes.
// Each of the LPT = TPW*G lanes owns one record of RW words
//     [ P rows (RPL x N) | x (RPL) | unwrap memory (UW, angular models) ]
// stored as 16-byte chunks, chunk c of all LPT lanes contiguous:
//     tile_base + c*LPT*16 + lane*16        (+ an 8/4-byte tail row when RW*sizeof(T) % 16 != 0)
// so every wavefront load/store instruction moves LPT*16 contiguous bytes (global_load_dwordx4).
#pragma once
#include <cstddef>
#include <cstdint>

namespace te {

// enum order of the reference: include/target_estimation/target_manager.hpp:38
enum ModelType : int { ANGULAR_RATES = 0, ANGULAR_VELOCITIES = 1, UNIFORM_ACCELERATION = 2, UNIFORM_VELOCITY = 3 };
enum DType : int { F64 = 0, F32 = 1 };

// Axis groups: the sets of state rows that the model's transition and measurement couple.  The
// linear models move every axis (x, y, z, roll, pitch, yaw) independently (A = I + dt E_K + ...
// only links rows r, r+K, r+2K); the EKF links the three Euler angles and the three body rates
// through its Jacobian (angular_velocities.cpp:116-124) and leaves x, y, z independent.  If Q, R
// and P0 have no entries between different groups -- true for every shipped model file, whose Q
// is Gamma diag(sigma^2) Gamma^T and whose R, P0 are diagonal (matlab/generateModel.m:9-41) --
// P keeps exact zeros there for ever and the filter factorises into one small filter per group.
constexpr int group_of(int type, int r) {
  return type == UNIFORM_VELOCITY ? r % 3
       : type == UNIFORM_ACCELERATION ? r % 3
       : type == ANGULAR_RATES ? r % 6
       : /* ANGULAR_VELOCITIES */ ((r % 6) < 3 ? r % 6 : 3);
}

struct ModelUV { static constexpr int TYPE = UNIFORM_VELOCITY, N = 6, K = 3, NB = 2; static constexpr bool ANGULAR = false, EKF = false; };
struct ModelUA { static constexpr int TYPE = UNIFORM_ACCELERATION, N = 9, K = 3, NB = 3; static constexpr bool ANGULAR = false, EKF = false; };
struct ModelAR { static constexpr int TYPE = ANGULAR_RATES, N = 18, K = 6, NB = 3; static constexpr bool ANGULAR = true, EKF = false; };
struct ModelAV { static constexpr int TYPE = ANGULAR_VELOCITIES, N = 12, K = 6, NB = 2; static constexpr bool ANGULAR = true, EKF = true; };

enum Layout : int { LAYOUT_FULL = 0, LAYOUT_PACKED = 1, LAYOUT_SEPARABLE = 2, LAYOUT_SEPARABLE_PACKED = 3 };

constexpr int model_n(int type) { return type == UNIFORM_VELOCITY ? 6 : type == UNIFORM_ACCELERATION ? 9 : type == ANGULAR_RATES ? 18 : 12; }
constexpr int model_m(int type) { return (type == UNIFORM_VELOCITY || type == UNIFORM_ACCELERATION) ? 3 : 6; }

// Position of Q(r, c) / R(r, c) inside one parameter-class row of a batch's (Q, R) table.  Dense layouts: the full
// matrices, row-major, [Q | R].  Separable layouts: only the entries inside an axis group (the others are zero by the
// layout's precondition), GROUP BY GROUP -- the group's Q block row-major, then its R block -- so that what one filter
// chain needs is one contiguous run (linear models: 4 + 1 / 9 + 1 words per chain; the EKF's attitude group 36 + 9):
// 15 / 30 / 60 / 60 words (UV / UA / AV / AR) instead of 45 / 90 / 180 / 360 scattered ones.
constexpr int qr_n_groups(int type) { return type == ANGULAR_RATES ? 6 : type == ANGULAR_VELOCITIES ? 4 : 3; }
// index of entry (r, c) of Q (is_r = false) or R (is_r = true) in the separable ordering, -1 if it lies between groups
constexpr int qr_sep_word(int type, bool is_r, int r, int c) {
  const int n = model_n(type), m = model_m(type);
  if (group_of(type, r) != group_of(type, c)) return -1;
  int k = 0;
  for (int g = 0; g < qr_n_groups(type); ++g) {
    for (int rr = 0; rr < n; ++rr)
      for (int cc = 0; cc < n; ++cc) {
        if (group_of(type, rr) != g || group_of(type, cc) != g) continue;
        if (!is_r && rr == r && cc == c) return k;
        ++k;
      }
    for (int rr = 0; rr < m; ++rr)
      for (int cc = 0; cc < m; ++cc) {
        if (group_of(type, rr) != g || group_of(type, cc) != g) continue;
        if (is_r && rr == r && cc == c) return k;
        ++k;
      }
  }
  return -1;
}
constexpr int qr_words(int type, bool sep) {
  const int n = model_n(type), m = model_m(type);
  if (!sep) return n * n + m * m;
  int k = 0;
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) k += group_of(type, r) == group_of(type, c) ? 1 : 0;
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < m; ++c) k += group_of(type, r) == group_of(type, c) ? 1 : 0;
  return k;
}
constexpr int qr_q_word(int type, bool sep, int r, int c) { return sep ? qr_sep_word(type, false, r, c) : r * model_n(type) + c; }
constexpr int qr_r_word(int type, bool sep, int r, int c) {   // r, c < m: the measurement rows are state rows 0..m-1
  return sep ? qr_sep_word(type, true, r, c) : model_n(type) * model_n(type) + r * model_m(type) + c;
}

// PK_ = symmetric-packed storage: only the upper triangle of P (r <= c, row-major, N(N+1)/2 words) is
// kept in HBM.  G = 1 (thread per target): the mirror into the registers is a register rename.
// G > 1: lane i of a target stores the i-th slice of ceil(N(N+1)/2 / G) words of the triangle; after
// the load the wavefront puts the triangle into its LDS scratch and every lane gathers its full rows
// from there (and the reverse before the store).  The reference's (I-KC)P is symmetric only to
// rounding (~1e-16 relative); packed batches keep the owner-row value P[r][c], r <= c, of each pair.
// LAYOUT_SEPARABLE: only the entries of P inside an axis group are stored (the others are
// structurally zero); thread per target, dedicated kernel (kf_step_sep.hpp).
template <class M, typename T, int G_, int LAYOUT_ = LAYOUT_FULL>
struct Cfg {
  static constexpr int G = G_;
  static constexpr int LAYOUT = LAYOUT_;
  static constexpr bool PK = LAYOUT_ == LAYOUT_PACKED;
  static constexpr bool SEP = LAYOUT_ == LAYOUT_SEPARABLE || LAYOUT_ == LAYOUT_SEPARABLE_PACKED;
  static constexpr bool SEPPK = LAYOUT_ == LAYOUT_SEPARABLE_PACKED;   // group blocks stored as upper triangles
  static_assert(!(LAYOUT_ == LAYOUT_SEPARABLE || LAYOUT_ == LAYOUT_SEPARABLE_PACKED) || G_ == 1, "separable storage uses the thread-per-target mapping");
  static constexpr int N = M::N, K = M::K, NB = M::NB;
  static_assert(K % G == 0, "lanes per target must divide the block size");
  static constexpr int RPL = N / G;           // rows of x / P per lane
  static constexpr int KPL = K / G;           // rows of one block (and of S) per lane
  static constexpr int TPW = 64 / G;          // targets per wavefront (= per tile)
  static constexpr int LPT = TPW * G;         // active lanes per tile
  static constexpr int UW = M::ANGULAR ? (3 + G - 1) / G : 0;  // unwrap-memory words per lane
  static constexpr int sep_count() {
    int k = 0;
    for (int r = 0; r < M::N; ++r)
      for (int c = (LAYOUT_ == LAYOUT_SEPARABLE_PACKED ? r : 0); c < M::N; ++c) k += group_of(M::TYPE, r) == group_of(M::TYPE, c) ? 1 : 0;
    return k;
  }
  static constexpr int TRI = N * (N + 1) / 2;                  // words of the upper triangle
  // Packed with G > 1: lane i holds the i-th slice of the triangle.  When TRI is not a multiple of G the slices were rounded
  // up -- angular_rates on 6 lanes: 29 words for 171 / 6 = 28.5, and with the unwrap slot every lane carries (three of the six
  // unused) 198 words stored for 192.  Where the unused unwrap slots can take the remainder they do (TRI_FOLD): slices of
  // TRI / G words, the last TRI % G words of the triangle ride in the spare unwrap slots (fold_lane / fold_slot), the record
  // is exactly the triangle + x + unwrap memory (and, in fp64, a whole number of 16-byte chunks: no tail row).
  static constexpr int TRI_REM = (PK && G > 1) ? TRI % G : 0;
  static constexpr bool uw_slot_used(int lane, int u) {         // does an angle's unwrap word live in (lane, slot u)?  (kf_aux.hpp unwrap_ptr)
    for (int cc = 0; cc < 3; ++cc)
      if (M::ANGULAR && (3 + cc) % G == lane && cc / G == u) return true;
    return false;
  }
  static constexpr int uw_spare() {
    int k = 0;
    for (int lane = 0; lane < G; ++lane)
      for (int u = 0; u < UW; ++u) k += uw_slot_used(lane, u) ? 0 : 1;
    return k;
  }
  static constexpr bool TRI_FOLD = PK && G > 1 && TRI_REM > 0 && TRI_REM <= uw_spare();
  static constexpr int fold_find(int e, bool want_lane) {       // the e-th spare unwrap slot, lanes ascending
    int k = 0;
    for (int lane = 0; lane < G; ++lane)
      for (int u = 0; u < UW; ++u) {
        if (uw_slot_used(lane, u)) continue;
        if (k == e) return want_lane ? lane : u;
        ++k;
      }
    return -1;
  }
  static constexpr int fold_lane(int e) { return fold_find(e, true); }
  static constexpr int fold_slot(int e) { return fold_find(e, false); }
  static constexpr int PW = SEP ? sep_count() : PK ? (TRI_FOLD ? TRI / G : (TRI + G - 1) / G) : RPL * N;   // words of P per lane in HBM
  static constexpr int RW = PW + RPL + UW;                     // record words per lane in HBM
  static constexpr int FRW = RPL * (N + 1) + UW;               // words of the full register image
  static constexpr int VW = 16 / (int)sizeof(T);               // words per 16-byte chunk
  static constexpr int NC = RW / VW;                           // full chunks
  static constexpr int REM = RW % VW;                          // tail words (fp64: 0/1, fp32: 0..3)
  static constexpr int REM2 = (sizeof(T) == 4 && REM >= 2) ? 1 : 0;  // an 8-byte tail piece
  static constexpr int REM1 = (sizeof(T) == 8) ? REM : (REM & 1);    // a one-word tail piece
  static constexpr long TAIL2_OFF = (long)NC * LPT * 16;
  static constexpr long TAIL1_OFF = TAIL2_OFF + (long)REM2 * LPT * 8;
  static constexpr long TILE_PAYLOAD = (long)LPT * RW * (long)sizeof(T);
  static constexpr long TILE_BYTES = (TILE_PAYLOAD + 127) / 128 * 128;
  // word offsets inside a record
  static constexpr int X_OFF = PW;
  static constexpr int UW_OFF = PW + RPL;
  // upper-triangle index of (r, c), r <= c
  static constexpr int tri(int r, int c) { return r * N - r * (r - 1) / 2 + (c - r); }
  // word of P(r, c) in the record of the lane that owns row r (G = 1 for the non-full layouts),
  // -1 for a structural zero of the separable layout
  static constexpr int p_word(int r, int c) {
    if (SEP) {
      if (group_of(M::TYPE, r) != group_of(M::TYPE, c)) return -1;
      const int r0 = (SEPPK && c < r) ? c : r, c0 = (SEPPK && c < r) ? r : c;
      int k = 0;
      for (int rr = 0; rr < N; ++rr)
        for (int cc = (SEPPK ? rr : 0); cc < N; ++cc) {
          if (rr == r0 && cc == c0) return k;
          k += group_of(M::TYPE, rr) == group_of(M::TYPE, cc) ? 1 : 0;
        }
      return -1;
    }
    if (PK) {
      const int t = r <= c ? tri(r, c) : tri(c, r);
      if (TRI_FOLD && t >= G * PW) return UW_OFF + fold_slot(t - G * PW);
      return t % PW;
    }
    return ((r % K) / G + (r / K) * KPL) * N + c;
  }
  // which of the G lanes of a target holds P(r, c)
  static constexpr int p_lane(int r, int c) {
    if (PK) {
      const int t = r <= c ? tri(r, c) : tri(c, r);
      if (TRI_FOLD && t >= G * PW) return fold_lane(t - G * PW);
      return t / PW;
    }
    return r % G;
  }
  // LDS exchange words per target (G > 1 only)
  static constexpr int EXA = K * N;                 // top rows of P^- (AV: also the 6 mid rows)
  static constexpr int EXB = K * K;                 // S^-1
  static constexpr int EXC = M::EKF ? N : K;        // pivot row / innovation (EKF: x exchange)
  // the packed triangle while it is transposed (G > 1); it aliases the exchange areas above, which are
  // only live inside the predict/update section
  static constexpr int EXP = (PK && G > 1) ? TRI : 0;
  // LDS column stride: one spare column so that the idle lanes (>= LPT) of a wave, whose group
  // index is TPW, never alias a live target's scratch
  static constexpr int GS = (64 + G - 1) / G;
  static constexpr int EX_WORDS = (G == 1) ? 0 : GS * ((EXA + EXB + EXC) > EXP ? (EXA + EXB + EXC) : EXP);
  // p_word as a compile-time table (so that fully unrolled kernels index registers statically)
  struct WordTable { int v[N][N]; };
  static constexpr WordTable make_table() {
    WordTable t{};
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) t.v[r][c] = p_word(r, c);
    return t;
  }
  static constexpr WordTable PWORD = make_table();
  static constexpr int QR_WORDS = qr_words(M::TYPE, SEP);   // words of one parameter-class row
  struct QTable { int v[N][N]; };
  struct RTable { int v[K][K]; };
  static constexpr QTable make_qtable() {
    QTable t{};
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) t.v[r][c] = qr_q_word(M::TYPE, SEP, r, c);
    return t;
  }
  static constexpr RTable make_rtable() {
    RTable t{};
    for (int r = 0; r < K; ++r)
      for (int c = 0; c < K; ++c) t.v[r][c] = qr_r_word(M::TYPE, SEP, r, c);
    return t;
  }
  static constexpr QTable QWORD = make_qtable();
  static constexpr RTable RWORD = make_rtable();
  // wavefronts per workgroup: as many as keep static LDS under 64 KiB
  static constexpr long LDS4 = (long)(4 * (QR_WORDS + EX_WORDS) + 1) * (long)sizeof(T);
  static constexpr long LDS2 = (long)(2 * (QR_WORDS + EX_WORDS) + 1) * (long)sizeof(T);
  static constexpr int WPB = (LDS4 <= 65536) ? 4 : ((LDS2 <= 65536) ? 2 : 1);
};

// Byte offset (from the tile base) of word w of the record of lane `lane`.
template <class C, typename T>
__host__ __device__ inline long record_word_offset(int lane, int w) {
  if (w < C::NC * C::VW) return (long)(w / C::VW) * C::LPT * 16 + (long)lane * 16 + (long)(w % C::VW) * (long)sizeof(T);
  int rw = w - C::NC * C::VW;
  if (C::REM2 && rw < 2) return C::TAIL2_OFF + (long)lane * 8 + (long)rw * 4;
  return C::TAIL1_OFF + (long)lane * (long)sizeof(T);
}

// the doorbell word of a resident ("live") launch (kf_step.hpp StepArgs::live_posted): ticks posted | stop bit
constexpr long long kLiveStop = (long long)(1ull << 63);    // sign bit: "stop once the posted ticks are done"
constexpr long long kLiveCount = (long long)(~(1ull << 63));
constexpr int kLiveGroup = 32;           // worker wavefronts that share one copy of the relay's mirror word
constexpr int kLiveMirrorStride = 16;    // long longs between copies (128 bytes: a line of its own each)
constexpr long kLiveScan = 64 * 32;      // progress words the relay reads per round trip (32 per lane); the array is padded to whole scans with INT_MAX

struct LayoutInfo {
  int n, m, g, layout, tpw, lpt, record_words;
  long tile_bytes, tile_payload;
};

}  // namespace te
