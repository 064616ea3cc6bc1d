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

struct ModelUV { static constexpr int TYPE = UNIFORM_VELOCITY, N = 6, K = 3, NB = 2; static constexpr bool ANGULAR = false, EKF = false; };
struct ModelUA { static constexpr int TYPE = UNIFORM_ACCELERATION, N = 9, K = 3, NB = 3; static constexpr bool ANGULAR = false, EKF = false; };
struct ModelAR { static constexpr int TYPE = ANGULAR_RATES, N = 18, K = 6, NB = 3; static constexpr bool ANGULAR = true, EKF = false; };
struct ModelAV { static constexpr int TYPE = ANGULAR_VELOCITIES, N = 12, K = 6, NB = 2; static constexpr bool ANGULAR = true, EKF = true; };

constexpr int model_n(int type) { return type == UNIFORM_VELOCITY ? 6 : type == UNIFORM_ACCELERATION ? 9 : type == ANGULAR_RATES ? 18 : 12; }
constexpr int model_m(int type) { return (type == UNIFORM_VELOCITY || type == UNIFORM_ACCELERATION) ? 3 : 6; }

// PK_ = symmetric-packed storage: only the upper triangle of P (r <= c, row-major) is kept in HBM
// and mirrored into the registers after the load.  Offered for G = 1 (thread per target), where
// the mirror is a register rename.  The reference's (I-KC)P is symmetric only to rounding
// (~1e-16 relative); packed batches keep the owner-row value P[r][c], r <= c, of each pair.
template <class M, typename T, int G_, bool PK_ = false>
struct Cfg {
  static constexpr int G = G_;
  static constexpr bool PK = PK_;
  static_assert(!PK_ || G_ == 1, "packed storage is implemented for the thread-per-target mapping");
  static constexpr int N = M::N, K = M::K, NB = M::NB;
  static_assert(K % G == 0, "lanes per target must divide the block size");
  static constexpr int RPL = N / G;           // rows of x / P per lane
  static constexpr int KPL = K / G;           // rows of one block (and of S) per lane
  static constexpr int TPW = 64 / G;          // targets per wavefront (= per tile)
  static constexpr int LPT = TPW * G;         // active lanes per tile
  static constexpr int UW = M::ANGULAR ? (3 + G - 1) / G : 0;  // unwrap-memory words per lane
  static constexpr int PW = PK ? N * (N + 1) / 2 : RPL * N;    // words of P per lane in HBM
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
  // LDS exchange words per target (G > 1 only)
  static constexpr int EXA = K * N;                 // top rows of P^- (AV: also the 6 mid rows)
  static constexpr int EXB = K * K;                 // S^-1
  static constexpr int EXC = M::EKF ? N : K;        // pivot row / innovation (EKF: x exchange)
  // LDS column stride: one spare column so that the idle lanes (>= LPT) of a wave, whose group
  // index is TPW, never alias a live target's scratch
  static constexpr int GS = (64 + G - 1) / G;
  static constexpr int EX_WORDS = (G == 1) ? 0 : GS * (EXA + EXB + EXC);
  static constexpr int QR_WORDS = N * N + K * K;
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

struct LayoutInfo {
  int n, m, g, packed, tpw, lpt, record_words;
  long tile_bytes, tile_payload;
};

}  // namespace te
