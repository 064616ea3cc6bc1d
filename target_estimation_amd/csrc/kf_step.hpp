// kf_step.hpp -- the fused predict+update kernel of the batched Kalman path (gfx950).
//
// What one launch does, per target (reference call it replaces in brackets):
//   updateA(dt)                      [src/types/*.cpp updateA; AV: Jacobians at the previous posterior]
//   measurement conversion           [addMeasurement: xyz, or quat -> normalise -> rpy -> unwrap]
//   x^- = A x | f(x);  P^- = (A P) A^T + Q          [src/kalman.cpp:84-88 | :129-133]
//   S = P^-[0:m,0:m] + R;  K = P^-[:,0:m] S^-1      [src/kalman.cpp:92 | :137]
//   x^+ = x^- + K (y - x^-[0:m])                    [src/kalman.cpp:93 | :138]
//   P^+ = (I - K C) P^-   (NOT Joseph form)         [src/kalman.cpp:94 | :139]
// C = [I_m 0] in every model, so C P C^T / P C^T / K C are selections (zero flops).
//
// Execution model: G lanes cooperate on one target (te_layout.hpp); each lane keeps its rows of
// P in VGPRs for the whole step.  The banded / block transition is applied in registers; the
// only cross-lane traffic is, per step and target, through a per-wavefront LDS scratch:
// the pivot rows of the Gauss-Jordan inverse of S, S^-1 (K x K), the innovation (K) and the
// top K rows of P^- (K x N).  A wavefront owns its scratch, so no s_barrier is executed after
// the one that publishes Q and R.  With G = 1 (thread per target) there is no exchange at all.
//
// Summation order follows the reference's dense products (k ascending; the structural zeros
// of A and C contribute exact zeros), with fma in place of mul+add.  S is inverted by
// unpivoted Gauss-Jordan (S is SPD); the reference's Eigen inverse is partial-pivot LU: same
// result to rounding, covered by the stated tolerance.
#pragma once
#include <hip/hip_runtime.h>

#include "ekf_sym.hpp"
#include "kf_aux.hpp"
#include "te_device_math.hpp"
#include "te_layout.hpp"

namespace te {

template <typename T>
struct StepArgs {
  char* rec;                 // lane records, tile-major (te_layout.hpp)
  // A -> B ("ping-pong") ticks: rec_out != null makes a dense launch read every record from `rec` and write it, with
  // nontemporal stores, to the same place in `rec_out` (a second buffer of the same layout; the batch swaps the two after the
  // launch).  Once the state is several times the 256 MB Infinity Cache nothing of a tick survives to the next one anyway,
  // and keeping reads and writes in separate address streams -- writes that allocate nowhere on their way -- moves more
  // bytes per second than read-modify-write in place (profiles/r02_rmw_ceiling.txt: 1.9 GB of state 5.30 -> 5.74 TB/s).
  // Same arithmetic, same results.  null: in place.
  char* rec_out;
  const T* qr;               // Q (N*N row-major) then R (K*K row-major), compute precision; PERQR: a table of such blocks
  const int* cls;            // PERQR only: parameter class of slot s (its block of the qr table)
  long n;                    // dense: number of targets; indexed: number of entries
  const int* idx;            // indexed only: slot of entry e
  const T* meas;             // SoA [7][meas_ld]; row c = component c of [x y z qx qy qz qw]; may be null (predict only)
  long meas_ld;
  const unsigned char* has_meas;  // per entry; null = every entry has a measurement (if meas != null)
  const double* dt_per;      // indexed only, optional per-entry dt
  double dt;
  double* t_base;            // per-slot time offset       (touched by the indexed path only)
  int* nm_base;              // per-slot measurement count (touched when a mask is given or indexed)
  // temporal fusion: n_ticks > 1 runs that many consecutive ticks in ONE launch with the state
  // kept in registers; tick s reads meas + s * tick_stride and has_meas + s * has_stride
  int n_ticks;
  long tick_stride;
  long has_stride;
  // fused own-time sphere query after the step (QUERY variants of the separable kernel only):
  // q_delta [n], q_pose [n][7] or null (device doubles)
  double q_origin[3];
  double q_radius;
  double* q_delta;
  double* q_pose;
  // Zig-zag traversal: reverse != 0 walks the tiles from the last to the first.  Alternating the direction between
  // consecutive ticks leaves the part of the state touched last in the 256 MB Infinity Cache for the start of the next
  // tick (tools/zigzag_ceiling.hip: 480 MB of state in place 5.3 -> 6.6 TB/s, 960 MB 5.4 -> 6.0); results are
  // independent of the order (targets are independent).
  int reverse;
  // measurements are read once per tick: nt_meas != 0 loads them with the nontemporal policy so that they do not push
  // state out of the Infinity Cache (set for batches large enough to zig-zag; TE_NT_MEAS overrides)
  int nt_meas;
  // Indexed launches of the one-target ABI's queue (Batch::flush): o_pose != null makes the kernel also write the derived
  // outputs of the stepped slots into the per-slot host table (pose [.][7], twist [.][6], acceleration [.][6]; what
  // outputs_kernel would write) and then store done_seq to *done_flag (host-mapped): step, getter table and completion signal
  // in one launch.  More than one wavefront of entries: done_count (a device word, zero between launches) counts the
  // wavefronts that have written their rows, and the last of the launch's ceil(n / TPW) stores the flag (signal_done).
  double* o_pose;
  double* o_twist;
  double* o_acc;
  int* done_flag;
  int done_seq;
  int* done_count;
  // RESIDENT ("live") launches of small batches (LIVE variants of the separable kernel; Batch::live_start): the state stays in
  // registers while the kernel waits, tick after tick, for the host to post that the tick's measurements are in the ring.
  // Host memory is touched by ONE wavefront only, the RELAY (the extra, last workgroup of the grid): a store from the GPU to
  // host memory costs 0.6-1.6 us and they serialise (measured: a progress word per worker wavefront per tick made a
  // 157-wavefront tick 250 us long), so the workers talk to device memory and the relay carries two words over PCIe:
  //   live_posted   host-mapped, host -> relay: ticks posted so far; sign bit = "stop once they are done"
  //   live_mirror   device words, relay -> workers: the same value (the relay adds the stop bit itself when the host has
  //                 been silent for live_spin_limit polls: a dead host leaves no kernel behind, and every worker stops at
  //                 the same tick).  One copy per kLiveGroup wavefronts, each in a 128-byte line of its own: 1563 wavefronts
  //                 polling ONE word queue up at one memory channel (a paced 10^5-target tick took 21 us that way)
  //   live_progress device words [wavefronts], worker -> relay: ticks this wavefront has finished
  //   live_done     host-mapped, relay -> host: the minimum of live_progress
  //   live_ring / live_first   the measurement ring holds live_ring ticks; tick k of the session reads entry (live_first + k) % live_ring
  // n_ticks = the most ticks the launch will serve.
  const long long* live_posted;
  long long* live_mirror;
  int* live_progress;
  int* live_done;
  long live_ring;
  long live_first;
  unsigned live_spin_limit;
  unsigned long long live_idle_ticks;   // the idle limit on the device's wall clock (0: count relay rounds instead, live_spin_limit)
  int live_flags;   // experiments (TE_LIVE_FLAGS): 8 = relay scans slowly always, see "Ordering of the hand-offs" below
  // live_pose (or null): SoA [7][live_pose_ld] doubles in device memory that receives the estimated pose of every target after
  // every tick (what the reference's node publishes every tick, src/target_manager_ros.cpp:78-87), written THROUGH the caches
  // before the tick's progress word, so that a copy engine that reads it after `done` reached the tick sees that tick's poses
  double* live_pose;
  long live_pose_ld;
};

__device__ __forceinline__ long long wave_uniform_ll(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// Worker side: wait until tick number `need` (1-based) has been posted.  false: stop requested with nothing more to do (or
// the backstop limit ran out).  `seen` caches the last value read, so a wavefront that is behind catches up without
// polling.  ONE lane reads the mirror word (a device-scope atomic load on the vector path: the scalar cache may hold a
// stale copy; 64 lanes loading one address past the caches would be 64 requests).
// Every 4096th poll (a millisecond or two of waiting) the wavefront also looks at the host's own word, over PCIe: a stop there with
// nothing left to serve ends it even when the relay is not there to pass it on -- a grid that never became fully resident, which
// Batch::live_start gives up on (the relay is its LAST workgroup).  A busy session never waits that long between ticks.
//
// Ordering of the hand-offs (profiles/r04_live_ordering.txt has the measurements and the ISA).  Every edge that costs nothing is a
// real release / acquire; the two on the WORKER side cost 2x (paced) and 15x (back to back) and order nothing that is not
// ordered by construction, so there the relaxed form stays, with the reason next to it:
//   host      ring entries, then posted        store RELEASE                       (Batch::live_post)
//   relay     posted                           load relaxed system, ACQUIRE fence (system) behind the round's scan
//             mirror                           RELEASE fence (agent), then the relaxed stores of the copies
//   worker    mirror                           load relaxed (agent) while polling.  No acquire fence on admission: what the tick
//                                              then reads from outside the kernel -- the ring entry -- is read with SYSTEM-scope loads
//                                              (sc0 sc1: past every cache) issued after the poll's s_waitcnt in program order, so there
//                                              is no cached copy an acquire's buffer_inv would have to drop; the fence costs one cache
//                                              invalidation per wavefront per tick, on the critical path of a paced stream (10^5 targets:
//                                              9.4 -> 19.9 us per paced tick).  kLiveWorkerAcq switches it on.
//             outputs of the tick (poses, query results): system-scope write-through stores, waited for (vmcnt 0)
//             progress                         store relaxed agent BEHIND that wait.  Not a release: everything a consumer may read
//                                              after `done` has left the wavefront with write-through stores whose completion the
//                                              s_waitcnt has seen (a plain session writes nothing per tick at all: the state is in
//                                              registers), so a release would add only its L2 write-back (buffer_wbl2) of lines nobody
//                                              waits for (10^5 targets 1.12 -> 17.3 us per tick).  kLiveWorkerRel switches it on.
//   relay     progress scan                    loads relaxed, the round's ACQUIRE fence behind the scan
//             done                             store RELEASE  system
//   host      done                             load  ACQUIRE                        (Batch::live_done)
// live_flags (TE_LIVE_FLAGS, measurement only): kLiveRelaxed = round 3's form, relaxed everywhere; kLiveWorkerAcq / kLiveWorkerRel
// switch the worker's two edges ON; kLiveNoRelayAcq / kLiveNoDoneRel / kLiveNoMirrorRel switch single relay edges OFF.
constexpr int kLiveRelaxed = 16;
constexpr int kLiveWorkerAcq = 32, kLiveWorkerRel = 64;
constexpr int kLiveNoRelayAcq = 128, kLiveNoDoneRel = 256, kLiveNoMirrorRel = 512;
__device__ __forceinline__ bool live_wait_tick(const long long* mirror, const long long* host_posted, long long need, unsigned limit, long long& seen, int lane, int flags = 0) {
  if ((seen & kLiveCount) >= need) return true;
  for (unsigned spins = 0;; ++spins) {
    long long v = 0;
    if (lane == 0) v = __hip_atomic_load(mirror, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    seen = wave_uniform_ll(v);
    if ((seen & kLiveCount) >= need) {
      if (flags & kLiveWorkerAcq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return true;
    }
    if (seen < 0) return false;        // stop, and every posted tick is done
    if (spins >= limit) return false;  // backstop (the relay stops the session long before)
    if ((spins & 4095u) == 4095u) {
      long long h = 0;
      if (lane == 0) h = __hip_atomic_load(host_posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      h = wave_uniform_ll(h);
      if (h < 0 && (h & kLiveCount) < need) return false;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

// The relay wavefront: host doorbell -> device mirror, worker progress -> host.  Leaves when a stop was requested (by the
// host, or by itself after `limit` polls without news from the host) and every worker has served the posted ticks.
__device__ __forceinline__ void live_relay(const long long* posted, long long* mirror, const int* progress, int* done, long waves,
                                           unsigned limit, unsigned long long idle_ticks, int lane, int flags) {
  long long last = 0;
  int last_done = 0;
  unsigned idle = 0;   // consecutive rounds in which nothing happened: no news from the host, no progress of the workers
  // ... and since when, on the device's constant-rate clock: the idle limit is a time (idle_ticks of wall_clock64; a round takes
  // 1.5 - 5 us depending on the grid, so a count of rounds was 20 % short of the seconds asked for); the count remains the fallback
  unsigned long long t_active = wall_clock64();
  // "running": the host set the word to -1 before the launch.  This is the LAST workgroup of the grid and workgroups are dispatched
  // in order, so the word also says that every worker has been given its wave slot (Batch::live_start waits for it).
  if (lane == 0) __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (;;) {
    // One PCIe read per round, requested FIRST and consumed LAST: the scan of the workers' words below goes out behind it and
    // the round costs the longer of the two round trips, not their sum.  The scan keeps up to 32 independent loads in flight per
    // lane (one after the other, 25 dependent round trips made a 1563-wavefront round 17 us long).
    long long v = 0;
    if (lane == 0) v = __hip_atomic_load(posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (its acquire: the fence behind the scan)
    int mn = 0x7fffffff;
    for (long w0 = 0; w0 < waves; w0 += kLiveScan) {
      int p[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        // every load unconditional: a load behind a per-element condition gets a branch and an `s_waitcnt vmcnt(0)` of its own
        // -- 25 serial round trips again.  The array is padded to whole rounds (kLiveScan words) with INT_MAX, which a minimum
        // ignores, so the 32 addresses are ONE base plus constants (clamped indices cost an address pair each, and those
        // registers -- this wavefront shares the kernel with the workers -- cost every batch its resident capacity).
        p[k] = __hip_atomic_load(&progress[w0 + (long)k * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) mn = p[k] < mn ? p[k] : mn;
    }
    v = wave_uniform_ll(v);
    // one acquire fence for everything this round has read: the host's word (system) and the workers' progress words (agent)
    if (!(flags & (kLiveRelaxed | kLiveNoRelayAcq))) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(mn, off, 64);
      mn = o < mn ? o : mn;
    }
    const bool progressed = mn != last_done;
    if (progressed) {
      if (lane == 0) {   // one PCIe write per change
        if (flags & (kLiveRelaxed | kLiveNoDoneRel)) __hip_atomic_store(done, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else __hip_atomic_store(done, mn, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      last_done = mn;
    }
    const bool caught_up = (long long)mn >= (last & kLiveCount);
    const bool idle_over = idle_ticks ? (idle > 0 && wall_clock64() - t_active >= idle_ticks) : idle >= limit;
    if (last < 0) v = last;                                    // stopping: the host's word no longer matters
    else if (caught_up && idle_over) v = last | kLiveStop;      // everything served and a silent host: stop at what was posted
    if (v != last) {
      const long groups = (waves + kLiveGroup - 1) / kLiveGroup;
      if (!(flags & (kLiveRelaxed | kLiveNoMirrorRel))) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // one release for all copies of the word
      for (long g = lane; g < groups; g += 64)
        __hip_atomic_store(&mirror[g * kLiveMirrorStride], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = v;
      idle = 0; t_active = wall_clock64();
    } else if (progressed) {
      idle = 0; t_active = wall_clock64();   // the workers are busy with ticks already posted: that is not an idle host
    } else {
      ++idle;
    }
    if (last < 0) {
      if ((long long)mn >= (last & kLiveCount)) break;   // every worker has served the posted ticks and is leaving
      if (idle_ticks ? (idle > 0 && wall_clock64() - t_active >= idle_ticks) : idle >= limit) break;   // (a worker that never ran: nothing more to wait for)
    }
    // The workers' progress stores and this wavefront's scan meet in the same lines: scanned back to back, a busy session's
    // ticks get slower (10^5 UA fp32: 1.2 -> 2.0 us per tick).  While the workers are more than a tick behind what is posted
    // nobody is waiting for the next completion word, so the scan can take its time; with at most one tick outstanding (a paced
    // stream) it stays tight.
    const long long behind = (last & kLiveCount) - (long long)mn;
    if ((flags & 8) || behind > 1) __builtin_amdgcn_s_sleep(96);
    else __builtin_amdgcn_s_sleep(2);
  }
  // "ended" (done[2], host-mapped like done[0]): the host can tell a session that is over -- stopped, or given up on an idle
  // host -- from one that is waiting, without asking the runtime about the stream
  if (lane == 0) __hip_atomic_store(done + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// measurement word of a live tick: written by a copy engine or the host while the kernel runs, so it is read past the caches
template <typename T> __device__ __forceinline__ T load_meas_live(const T* p);
template <> __device__ __forceinline__ double load_meas_live<double>(const double* p) {
  const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return __longlong_as_double((long long)b);
}
template <> __device__ __forceinline__ float load_meas_live<float>(const float* p) {
  const unsigned b = __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return __uint_as_float(b);
}

template <typename T> __device__ __forceinline__ T load_meas(const T* p, int nt) { return nt ? __builtin_nontemporal_load(p) : *p; }

template <typename T> struct Vec16;
template <> struct Vec16<double> { using type = double2; };
template <> struct Vec16<float> { using type = float4; };

__device__ __forceinline__ void wave_lds_fence() {
  // same-wavefront LDS hand-off: DS instructions of one wave execute in order, so this only has
  // to stop the compiler from moving LDS accesses across it
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A copy the register coalescer cannot see through.  The 16-byte loads and stores of a record define and use 4-register
// tuples; a value that is updated in place would have to be allocated inside the tuple it was loaded into AND inside the
// tuple it is stored from, for its whole life, and the allocator answers by keeping both images.  Copying every word out of
// its load tuple (and into its store tuple) with an opaque move lets the words live and die one by one.
__device__ __forceinline__ float opaque_copy(float v) { float o; asm("v_mov_b32 %0, %1" : "=v"(o) : "v"(v)); return o; }
__device__ __forceinline__ double opaque_copy(double v) { double o; asm("v_mov_b64 %0, %1" : "=v"(o) : "v"(v)); return o; }

template <class C, typename T>
__device__ __forceinline__ void load_record(const char* tb, int lane, T* rec) {
  using V = typename Vec16<T>::type;
#pragma unroll
  for (int c = 0; c < C::NC; ++c) {
    V v = *reinterpret_cast<const V*>(tb + (long)c * C::LPT * 16 + (long)lane * 16);
    if constexpr (sizeof(T) == 8) { rec[c * 2] = v.x; rec[c * 2 + 1] = v.y; }
    else { rec[c * 4] = v.x; rec[c * 4 + 1] = v.y; rec[c * 4 + 2] = v.z; rec[c * 4 + 3] = v.w; }
  }
  if constexpr (C::REM2) {
    float2 v = *reinterpret_cast<const float2*>(tb + C::TAIL2_OFF + (long)lane * 8);
    rec[C::NC * C::VW] = v.x; rec[C::NC * C::VW + 1] = v.y;
  }
  if constexpr (C::REM1) rec[C::RW - 1] = *reinterpret_cast<const T*>(tb + C::TAIL1_OFF + (long)lane * (long)sizeof(T));
}

template <class C, typename T, bool OPAQUE = false, bool NT = false>
__device__ __forceinline__ void store_record(char* tb, int lane, const T* rec) {
  using V = typename Vec16<T>::type;
#pragma unroll
  for (int c = 0; c < C::NC; ++c) {
    V v;
    if constexpr (OPAQUE) {   // words gathered into their store tuple four at a time, never all 4-tuples at once
      if constexpr (sizeof(T) == 8) { v.x = opaque_copy(rec[c * 2]); v.y = opaque_copy(rec[c * 2 + 1]); }
      else { v.x = opaque_copy(rec[c * 4]); v.y = opaque_copy(rec[c * 4 + 1]); v.z = opaque_copy(rec[c * 4 + 2]); v.w = opaque_copy(rec[c * 4 + 3]); }
    } else if constexpr (sizeof(T) == 8) { v.x = rec[c * 2]; v.y = rec[c * 2 + 1]; }
    else { v.x = rec[c * 4]; v.y = rec[c * 4 + 1]; v.z = rec[c * 4 + 2]; v.w = rec[c * 4 + 3]; }
    if constexpr (NT) {   // the builtin takes native vectors, not HIP's wrapper types
      typedef float nt4 __attribute__((ext_vector_type(4)));
      nt4 raw;
      __builtin_memcpy(&raw, &v, 16);
      __builtin_nontemporal_store(raw, reinterpret_cast<nt4*>(tb + (long)c * C::LPT * 16 + (long)lane * 16));
    } else {
      *reinterpret_cast<V*>(tb + (long)c * C::LPT * 16 + (long)lane * 16) = v;
    }
    if constexpr (OPAQUE) {
      if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (C::REM2) {
    float2 v; v.x = rec[C::NC * C::VW]; v.y = rec[C::NC * C::VW + 1];
    *reinterpret_cast<float2*>(tb + C::TAIL2_OFF + (long)lane * 8) = v;
  }
  if constexpr (C::REM1) *reinterpret_cast<T*>(tb + C::TAIL1_OFF + (long)lane * (long)sizeof(T)) = rec[C::RW - 1];
}

template <typename T> __device__ __forceinline__ T sel3(int c, T a, T b, T d) { return c == 0 ? a : (c == 1 ? b : d); }

// StepArgs::o_pose: the row of the per-slot getter table for one stepped target (outputs_kernel's arithmetic on the same
// posterior state); the completion flag follows (signal_done).
template <class M, typename T>
__device__ __forceinline__ void write_outputs_row(const T* x, long slot, double* o_pose, double* o_twist, double* o_acc) {
  T pose7[7], twist6[6], acc6[6];
  derive_outputs<M, T>(x, false, (T)0, pose7, twist6, acc6);
#pragma unroll
  for (int c = 0; c < 7; ++c) o_pose[slot * 7 + c] = (double)pose7[c];
#pragma unroll
  for (int c = 0; c < 6; ++c) o_twist[slot * 6 + c] = (double)twist6[c];
#pragma unroll
  for (int c = 0; c < 6; ++c) o_acc[slot * 6 + c] = (double)acc6[c];
}

// The completion signal of such a launch.  Each lane makes its own rows visible to the host (system-scope fence), then the lanes
// meet at a wave barrier -- an explicit ordering of the other lanes' fences before lane 0's next step, instead of relying on the
// wave running in lockstep.  A launch of one wavefront: lane 0 publishes the sequence number the host spins on.  A launch of
// several: lane 0 counts its wavefront in -- a relaxed add BEHIND the system-scope fence above, which is the release -- and the
// wavefront that finds itself last (an acquire fence then puts every other wavefront's fence, and so its rows, before what
// follows) sets the counter back to zero for the next launch and publishes the number.
__device__ __forceinline__ void signal_done(int* flag, int seq, int lane, int* count, long n_entries, int tpw) {
  __threadfence_system();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane == 0) {
    const int total = (int)((n_entries + tpw - 1) / tpw);
    bool last = true;
    if (total > 1) {
      last = __hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (last) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// QUERY: the own-time sphere query of the target runs after the store, on the posterior state (kf_aux.hpp,
// sphere_query); with G > 1 the state is first collected from the G lanes through the wave's LDS scratch.
// PERQR: every target reads the Q and R of its own parameter class (TargetManager::init takes Q, R per target,
// target_manager.hpp:85-87) from a table in HBM (L2-resident for 10^3 classes) instead of the one pair staged in LDS.
// Minimum wavefronts per SIMD the register allocation must leave room for.  1 = no constraint, except angular_rates fp64 on
// the upper triangle with 6 lanes per target: unconstrained it takes 262 registers (one wavefront per SIMD, 1034 us per
// 10^6-target tick); held to 256 it parks four doubles in scratch (32-56 B per lane) and runs two wavefronts: 647 us.
// (The thread-per-target symmetric EKF got under the limit by other means: opaque_copy above and kf_model_av_sym.hip.)
// see LATE_MEAS in kf_step_kernel
template <class M, typename T, int G, int LAYOUT>
constexpr bool kLateMeas = (M::TYPE == ANGULAR_RATES && LAYOUT == LAYOUT_PACKED && G == 6 && sizeof(T) == 8);

template <class M, typename T, int G, int LAYOUT>
constexpr int step_min_waves() { return (M::TYPE == ANGULAR_RATES && LAYOUT == LAYOUT_PACKED && G == 6 && sizeof(T) == 8) ? 2 : 1; }

// AB: an A -> B tick (StepArgs::rec_out).  Its own instantiation, not a run-time branch around the record stores: with the
// branch some kernels kept both store sequences' operands alive and fell to one wavefront per SIMD (angular_rates fp32 on
// the upper triangle, 3 lanes per target: 242 -> 299 registers, 304 -> 513 us per 10^6-target tick).
template <class M, typename T, int G, int LAYOUT, bool INDEXED, bool FUSED = false, bool QUERY = false, bool PERQR = false, bool AB = false>
__global__ void __launch_bounds__((Cfg<M, T, G, LAYOUT>::WPB * 64), (step_min_waves<M, T, G, LAYOUT>()))
kf_step_kernel(const StepArgs<T> a) {
  static_assert(!AB || (!INDEXED && !FUSED && !QUERY), "A -> B ticks are dense single-tick launches without the fused query");
  using C = Cfg<M, T, G, LAYOUT>;
  constexpr bool PK = C::PK;
  // the EKF on a symmetric-packed covariance, thread per target: works on the triangle in place (ekf_sym.hpp)
  constexpr bool EKF_SYM = M::EKF && PK && G == 1;
  static_assert(!(QUERY && (INDEXED || FUSED)), "the fused query is for dense single-tick launches");
  static_assert(!(PERQR && (FUSED || QUERY)), "per-class Q/R: single-tick launches without the fused query");
  static_assert(!C::SEP, "the separable layout has its own kernel (kf_step_sep.hpp)");
  constexpr int N = C::N, K = C::K, RPL = C::RPL, KPL = C::KPL, TPW = C::TPW, GS = C::GS;
  constexpr int kStepWaves = C::WPB, kStepThreads = C::WPB * 64;
  using F = Mth<T>;

  // Q and R are staged per wavefront (no workgroup barrier anywhere in the kernel), after the
  // record loads have been issued so that both round trips overlap.
  __shared__ T s_qr[C::QR_WORDS * kStepWaves];
  __shared__ T s_ex[(C::EX_WORDS > 0 ? C::EX_WORDS : 1) * kStepWaves];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // the launcher may use fewer wavefronts per workgroup than the LDS arrays are sized for (small grids)
  long wg = (long)blockIdx.x * (blockDim.x >> 6) + wave;  // wavefront-global index
  if (wg * TPW >= a.n) return;                           // wave-uniform
  if (a.reverse) wg = (a.n + TPW - 1) / TPW - 1 - wg;
  const int g = lane / G;
  const int i = (G == 1) ? 0 : lane % G;
  const long entry = wg * TPW + g;
  bool valid = (lane < C::LPT) && (entry < a.n);

  long tile;
  int lt;  // lane inside the tile
  long slot_of = entry;
  if constexpr (INDEXED) {
    long slot = valid ? (long)a.idx[entry] : -1;   // a negative slot = "skip this entry" (an id the device table did not resolve)
    valid = slot >= 0;
    if (!valid) slot = 0;
    slot_of = slot;
    tile = slot / TPW;
    lt = (int)(slot % TPW) * G + i;
  } else {
    tile = wg;
    lt = lane;
  }
  char* tb = a.rec + tile * C::TILE_BYTES;
  const T* qr_own = a.qr;    // PERQR: this target's block of the table (per-lane vector loads)
  if constexpr (PERQR) {
    if (valid) qr_own = a.qr + (long)a.cls[slot_of] * C::QR_WORDS;
  }

  // `mem` is the HBM image of the record, `rec` the full register image the step works on; they
  // are the same array unless the batch stores P symmetric-packed.
  T mem[C::RW];
  if (valid) {
    load_record<C, T>(tb, lt, mem);
  } else {
#pragma unroll
    for (int w = 0; w < C::RW; ++w) mem[w] = 0;
  }
  if constexpr (EKF_SYM) {   // updated in place: the words must not stay tied to their load tuples (opaque_copy)
#pragma unroll
    for (int w = 0; w < C::RW; ++w) mem[w] = opaque_copy(mem[w]);
  }
  T* sQw = s_qr + C::QR_WORDS * wave;
  if constexpr (!PERQR) {
    for (int e = lane; e < C::QR_WORDS; e += 64) sQw[e] = a.qr[e];
    wave_lds_fence();
  }
  const T* sQ = PERQR ? qr_own : sQw;
  const T* sR = sQ + N * N;
  // per-wave LDS scratch: element `idx` of this lane's target at [idx*GS + g]
  T* sx = s_ex + (C::EX_WORDS > 0 ? C::EX_WORDS : 1) * wave;
#define EXA_(idx) sx[(idx) * GS + g]
#define EXB_(idx) sx[(C::EXA + (idx)) * GS + g]
#define EXC_(idx) sx[(C::EXA + C::EXB + (idx)) * GS + g]
#define EXP_(idx) sx[(idx) * GS + g]   /* aliases EXA/EXB/EXC: used only outside the predict/update section */

  T rec[C::FRW];
  // packed with G > 1: the triangle passes through the wave's LDS scratch.  Row r = i + G q of this lane;
  // element (r, c) of the triangle is tri(r, c) = tri_base[q] + c for c >= r, tri(c, r) = tri_col(c) + r below.
  int tri_base[RPL];
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    const int r = i + G * q;
    tri_base[q] = r * N - r * (r - 1) / 2 - r;
  }
  auto rows_from_lds = [&]() {   // gather the full rows of this lane from the triangle in LDS
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int r = i + G * q;
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const int idx = (c >= r) ? tri_base[q] + c : (c * N - c * (c - 1) / 2 - c) + r;
        rec[q * N + c] = EXP_(idx);
      }
    }
  };
  auto rows_to_lds = [&]() {     // scatter the upper part (c >= r) of this lane's rows into the triangle
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int r = i + G * q;
#pragma unroll
      for (int c = 0; c < N; ++c)
        if (c >= r) EXP_(tri_base[q] + c) = rec[q * N + c];
    }
  };
  if constexpr (EKF_SYM) {
    // no register image: the step works on `mem` directly
  } else if constexpr (PK && G == 1) {
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) rec[r * N + c] = mem[r <= c ? C::tri(r, c) : C::tri(c, r)];
#pragma unroll
    for (int w = 0; w < RPL + C::UW; ++w) rec[RPL * N + w] = mem[C::X_OFF + w];
  } else if constexpr (PK) {
#pragma unroll
    for (int k = 0; k < C::PW; ++k) {
      const int t = i * C::PW + k;
      if (t < C::TRI) EXP_(t) = mem[k];
    }
    if constexpr (C::TRI_FOLD) {   // the remainder of the triangle rides in the unwrap slots no angle uses (te_layout.hpp)
#pragma unroll
      for (int e = 0; e < C::TRI_REM; ++e)
        if (i == C::fold_lane(e)) EXP_(G * C::PW + e) = mem[C::UW_OFF + C::fold_slot(e)];
    }
    wave_lds_fence();
    rows_from_lds();
    wave_lds_fence();
#pragma unroll
    for (int w = 0; w < RPL + C::UW; ++w) rec[RPL * N + w] = mem[C::X_OFF + w];
  } else {
#pragma unroll
    for (int w = 0; w < C::RW; ++w) rec[w] = mem[w];
  }
  // views into the register image (compile-time indices only)
#define P_(q, c) rec[(q) * N + (c)]
#define X_(q) rec[RPL * N + (q)]
#define UW_(s) rec[RPL * N + RPL + (s)]

  double dtd = a.dt;
  if constexpr (INDEXED) {
    if (a.dt_per && valid) dtd = a.dt_per[entry];
  }
  const T dt = (T)dtd;
  int n_has = 0;
  const int n_ticks = FUSED ? a.n_ticks : 1;   // the single-tick kernels are compiled without the loop
  for (int tick = 0; tick < n_ticks; ++tick) {
  const T* meas_t = a.meas ? a.meas + (long)tick * a.tick_stride : nullptr;
  const unsigned char* has_t = a.has_meas ? a.has_meas + (long)tick * a.has_stride : nullptr;
  // Every measurement word this lane needs is requested up front, right behind the record loads and
  // regardless of the mask (one round trip instead of one per use; see kf_step_sep.hpp).
  // LATE_MEAS (angular_rates fp64 on the upper triangle, 6 lanes per target): the kernel sits at the two-wave register limit,
  // and five measurement doubles (then three angles and a position word) held across the whole predict were what pushed
  // four doubles of the record into scratch -- 10 % more HBM traffic than the record itself.  There the measurement is
  // requested behind the predict instead; the other wavefront of the SIMD covers the round trip.
  constexpr bool LATE_MEAS = kLateMeas<M, T, G, LAYOUT> && !FUSED;
  T ymeas_own[KPL], qmeas[4] = {0, 0, 0, 1};
#pragma unroll
  for (int qq = 0; qq < KPL; ++qq) ymeas_own[qq] = 0;
  unsigned char hmask = 1;
  auto load_measurement = [&]() {
#pragma unroll
    for (int qq = 0; qq < KPL; ++qq) {
      const int r = i + G * qq;
      if (!M::ANGULAR || r < 3) ymeas_own[qq] = load_meas(&meas_t[(long)r * a.meas_ld + entry], a.nt_meas);
    }
    if constexpr (M::ANGULAR) {
#pragma unroll
      for (int c = 0; c < 4; ++c) qmeas[c] = load_meas(&meas_t[(long)(3 + c) * a.meas_ld + entry], a.nt_meas);
    }
  };
  if (valid && meas_t != nullptr) {
    if constexpr (!LATE_MEAS) load_measurement();
    if (has_t != nullptr) hmask = has_t[entry];
  }
  const bool has = valid && meas_t != nullptr && hmask != 0;
  n_has += has ? 1 : 0;

  // ------------------------------------------------------------------ measurement conversion
  // angular models: quaternion -> normalise -> rpy (every lane of the group redundantly)
  T mrpy[3] = {0, 0, 0};
  if constexpr (M::ANGULAR && !LATE_MEAS) {
    if (has) {
      T q[4];
      q[0] = qmeas[0];
      q[1] = qmeas[1];
      q[2] = qmeas[2];
      q[3] = qmeas[3];
      quat_normalize(q);
      quat_to_rpy(q, mrpy);
    }
  }

  if constexpr (EKF_SYM) {
    ekf_sym_tick<C, T>(mem, sQ, sR, dt, has, ymeas_own, mrpy);
  } else {
  // ------------------------------------------------------------------ predict
  if constexpr (!M::EKF) {
    const T hdt = (T)0.5 * dt * dt;  // angular_rates.cpp:114 / uniform_acceleration.cpp:98
    // x^- = A x  and  AP = A*P: rows r, r+K, r+2K are q, q+KPL, q+2KPL of this lane
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int b = q / KPL;  // block of this row
      if (b + 1 < C::NB) {
        X_(q) = F::fma(dt, X_(q + KPL), X_(q));
        if (C::NB == 3 && b == 0) X_(q) = F::fma(hdt, X_(q + 2 * KPL), X_(q));
#pragma unroll
        for (int c = 0; c < N; ++c) {
          T v = F::fma(dt, P_(q + KPL, c), P_(q, c));
          if (C::NB == 3 && b == 0) v = F::fma(hdt, P_(q + 2 * KPL, c), v);
          P_(q, c) = v;
        }
      }
    }
    // (AP) * A^T, column blocks, then + Q
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int r = i + G * (q % KPL) + K * (q / KPL);
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const int bc = c / K;
        T v = P_(q, c);
        if (bc + 1 < C::NB) {
          v = F::fma(dt, P_(q, c + K), v);
          if (C::NB == 3 && bc == 0) v = F::fma(hdt, P_(q, c + 2 * K), v);
        }
        P_(q, c) = v + sQ[r * N + c];
      }
    }
  } else {
    // ---- EKF (angular velocities): A blocks of angular_velocities.cpp:116-124 evaluated at
    // the previous posterior, f of :126-140.  Rows: [xyz rpy | vel omega], K = 6, NB = 2.
    T rpy[3], om[3];
    if constexpr (G == 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { rpy[c] = X_(3 + c); om[c] = X_(9 + c); }
    } else {
#pragma unroll
      for (int q = 0; q < RPL; ++q) EXC_(i + G * (q % KPL) + K * (q / KPL)) = X_(q);
      wave_lds_fence();
#pragma unroll
      for (int c = 0; c < 3; ++c) { rpy[c] = EXC_(3 + c); om[c] = EXC_(9 + c); }
    }
    T s_r, c_r, s_p, c_p;
    F::sincos(rpy[0], &s_r, &c_r);
    F::sincos(rpy[1], &s_p, &c_p);
    const T wy = om[1], wz = om[2];
    // geometry.hpp:394-410 EarBaseInvJacobianRpy and :412-426 EarBaseInvJacobianOmega
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
    // geometry.hpp:359-374 rpyToEarBaseInv
    Ei[0][0] = 1; Ei[0][1] = (s_p * s_r) / c_p; Ei[0][2] = (c_r * s_p) / c_p;
    Ei[1][0] = 0; Ei[1][1] = c_r;               Ei[1][2] = -s_r;
    Ei[2][0] = 0; Ei[2][1] = s_r / c_p;         Ei[2][2] = c_r / c_p;

    // rows 3..5 and 9..11 of P are needed by the rpy rows of A*P
    T mid[6][N];
    if constexpr (G == 1) {
#pragma unroll
      for (int c = 0; c < N; ++c) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { mid[k][c] = P_(3 + k, c); mid[3 + k][c] = P_(9 + k, c); }
      }
    } else {
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const int r = i + G * (q % KPL) + K * (q / KPL);
        const int rr = r % 6;  // 3..5 for the rows of interest
        if (rr >= 3) {
          const int m6 = (r >= 6 ? 3 : 0) + (rr - 3);
#pragma unroll
          for (int c = 0; c < N; ++c) EXA_(m6 * N + c) = P_(q, c);
        }
      }
      wave_lds_fence();
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < N; ++c) mid[k][c] = EXA_(k * N + c);
    }
    // x^- = f(x) and the top half of A*P (rows 6..11 are unchanged)
#pragma unroll
    for (int q = 0; q < KPL; ++q) {
      const int r = i + G * q;  // < 6
      const int cc = r - 3;
      const bool rot = r >= 3;
      const T e0 = sel3(cc, Ei[0][0], Ei[1][0], Ei[2][0]);
      const T e1 = sel3(cc, Ei[0][1], Ei[1][1], Ei[2][1]);
      const T e2 = sel3(cc, Ei[0][2], Ei[1][2], Ei[2][2]);
      T acc = (dt * e0) * om[0];
      acc = F::fma(dt * e1, om[1], acc);
      acc = F::fma(dt * e2, om[2], acc);
      const T xlin = F::fma(dt, X_(q + KPL), X_(q));
      X_(q) = rot ? X_(q) + acc : xlin;
      const T jr0 = sel3(cc, Jr[0][0], Jr[1][0], Jr[2][0]), jr1 = sel3(cc, Jr[0][1], Jr[1][1], Jr[2][1]),
              jr2 = sel3(cc, Jr[0][2], Jr[1][2], Jr[2][2]);
      const T jw0 = sel3(cc, Jw[0][0], Jw[1][0], Jw[2][0]), jw1 = sel3(cc, Jw[0][1], Jw[1][1], Jw[2][1]),
              jw2 = sel3(cc, Jw[0][2], Jw[1][2], Jw[2][2]);
#pragma unroll
      for (int c = 0; c < N; ++c) {
        T v = jr0 * mid[0][c];
        v = F::fma(jr1, mid[1][c], v);
        v = F::fma(jr2, mid[2][c], v);
        v = F::fma(jw0, mid[3][c], v);
        v = F::fma(jw1, mid[4][c], v);
        v = F::fma(jw2, mid[5][c], v);
        const T lin = F::fma(dt, P_(q + KPL, c), P_(q, c));
        P_(q, c) = rot ? v : lin;
      }
    }
    // (AP) * A^T in registers, + Q
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int r = i + G * (q % KPL) + K * (q / KPL);
      T nw[6];
#pragma unroll
      for (int j = 0; j < 3; ++j) nw[j] = F::fma(dt, P_(q, j + 6), P_(q, j));
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        T v = P_(q, 3) * Jr[cc][0];
        v = F::fma(P_(q, 4), Jr[cc][1], v);
        v = F::fma(P_(q, 5), Jr[cc][2], v);
        v = F::fma(P_(q, 9), Jw[cc][0], v);
        v = F::fma(P_(q, 10), Jw[cc][1], v);
        v = F::fma(P_(q, 11), Jw[cc][2], v);
        nw[3 + cc] = v;
      }
#pragma unroll
      for (int c = 0; c < N; ++c) P_(q, c) = (c < 6 ? nw[c < 6 ? c : 0] : P_(q, c)) + sQ[r * N + c];
    }
  }

  // ------------------------------------------------------------------ update (estimate)
  if constexpr (LATE_MEAS) {
    if (has) {
      load_measurement();
      T q[4] = {qmeas[0], qmeas[1], qmeas[2], qmeas[3]};
      quat_normalize(q);
      quat_to_rpy(q, mrpy);
    }
  }
  if (has) {
    // S = P^-[0:K,0:K] + R and its inverse by unpivoted in-place Gauss-Jordan (S is SPD).
    //  LOCAL_INV (G == 1, or K == 3): every lane holds all of S and inverts it privately (one LDS
    //    round trip to collect the rows when G > 1; 27 fma for K = 3);
    //  otherwise (K == 6, G > 1): rows stay distributed, the scaled pivot row of each of the K
    //    elimination steps goes through LDS, and S^-1 is published for the gain.
    // Both orders of operations are the same, so every G gives bit-identical results.
    constexpr bool LOCAL_INV = (G == 1) || (K == 3);
    T S[LOCAL_INV ? K : KPL][K];
    if constexpr (LOCAL_INV) {
      if constexpr (G == 1) {
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
          for (int c = 0; c < K; ++c) S[r][c] = P_(r, c) + sR[r * K + c];
      } else {
#pragma unroll
        for (int qq = 0; qq < KPL; ++qq)
#pragma unroll
          for (int c = 0; c < K; ++c) EXB_((i + G * qq) * K + c) = P_(qq, c) + sR[(i + G * qq) * K + c];
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
          for (int c = 0; c < K; ++c) S[r][c] = EXB_(r * K + c);
      }
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
    } else {
#pragma unroll
      for (int qq = 0; qq < KPL; ++qq)
#pragma unroll
        for (int c = 0; c < K; ++c) S[qq][c] = P_(qq, c) + sR[(i + G * qq) * K + c];
#pragma unroll
      for (int p = 0; p < K; ++p) {
        const int ip = p % G, qp = p / G;
        T prow[K];
        if (i == ip) {
          const T inv = (T)1 / S[qp][p];
          S[qp][p] = 1;
#pragma unroll
          for (int c = 0; c < K; ++c) { S[qp][c] *= inv; EXC_(c) = S[qp][c]; }
        }
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < K; ++c) prow[c] = EXC_(c);
        wave_lds_fence();
#pragma unroll
        for (int qq = 0; qq < KPL; ++qq) {
          const bool is_piv = (qq == qp) && (i == ip);
          const T f = is_piv ? (T)0 : S[qq][p];
          if (!is_piv) S[qq][p] = 0;
#pragma unroll
          for (int c = 0; c < K; ++c) S[qq][c] = F::fma(-f, prow[c], S[qq][c]);
        }
      }
#pragma unroll
      for (int qq = 0; qq < KPL; ++qq)
#pragma unroll
        for (int c = 0; c < K; ++c) EXB_((i + G * qq) * K + c) = S[qq][c];
    }
    // publish the top K rows of P^- (needed by every lane of the group for the covariance update)
    if constexpr (G > 1) {
#pragma unroll
      for (int qq = 0; qq < KPL; ++qq)
#pragma unroll
        for (int c = 0; c < N; ++c) EXA_((i + G * qq) * N + c) = P_(qq, c);
    }
    // innovation y - x^-[0:K]; y = xyz | unwrapped rpy (angular_rates.cpp:81-88)
    T nu[K];
    {
      T nu_own[KPL];
#pragma unroll
      for (int qq = 0; qq < KPL; ++qq) {
        const int r = i + G * qq;
        T y;
        if (!M::ANGULAR || r < 3) {
          y = ymeas_own[qq];
        } else {
          const int cc = r - 3;
          const int us = cc / G;
          T prev = 0;
#pragma unroll
          for (int s = 0; s < C::UW; ++s) prev = (s == us) ? UW_(s) : prev;
          y = unwrap_angle(prev, sel3(cc, mrpy[0], mrpy[1], mrpy[2]));
#pragma unroll
          for (int s = 0; s < C::UW; ++s) UW_(s) = (s == us) ? y : UW_(s);
        }
        nu_own[qq] = y - X_(qq);
        if constexpr (G > 1) EXC_(r) = nu_own[qq];
      }
      if constexpr (G == 1) {
#pragma unroll
        for (int l = 0; l < K; ++l) nu[l] = nu_own[l];
      }
    }
    if constexpr (G > 1) {
      wave_lds_fence();
#pragma unroll
      for (int l = 0; l < K; ++l) nu[l] = EXC_(l);
    }
    // K = (P^- C^T) S^-1, rows of this lane
    T Kg[RPL][K];
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
      for (int c = 0; c < K; ++c) {
        T v;
        if constexpr (LOCAL_INV) v = S[c][l]; else v = EXB_(c * K + l);
#pragma unroll
        for (int q = 0; q < RPL; ++q) Kg[q][l] = (c == 0) ? P_(q, 0) * v : F::fma(P_(q, c), v, Kg[q][l]);
      }
    // x^+ = x^- + K nu
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      T acc = Kg[q][0] * nu[0];
#pragma unroll
      for (int l = 1; l < K; ++l) acc = F::fma(Kg[q][l], nu[l], acc);
      X_(q) += acc;
    }
    // P^+ = (I - K C) P^-: D = I - K C has columns 0..K-1 and the diagonal
    T D[RPL][K];
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int r = i + G * (q % KPL) + K * (q / KPL);
#pragma unroll
      for (int j = 0; j < K; ++j) D[q][j] = ((r == j) ? (T)1 : (T)0) - Kg[q][j];
    }
    T top[K];  // column c of the top K rows of P^-
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if constexpr (G == 1) top[j] = P_(j, c); else top[j] = EXA_(j * N + c);
      }
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        T acc = D[q][0] * top[0];
#pragma unroll
        for (int j = 1; j < K; ++j) acc = F::fma(D[q][j], top[j], acc);
        P_(q, c) = (q < KPL) ? acc : acc + P_(q, c);
      }
    }
  }

  }  // !EKF_SYM
  if constexpr (FUSED && PK && !EKF_SYM) {
    // a packed batch re-symmetrises P every tick (store upper triangle, reload mirrored)
    if constexpr (G == 1) {
#pragma unroll
      for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = r + 1; c < N; ++c) rec[c * N + r] = rec[r * N + c];
    } else {
      wave_lds_fence();
      rows_to_lds();
      wave_lds_fence();
      rows_from_lds();
      wave_lds_fence();
    }
  }
  }  // tick loop
  if constexpr (PK && G > 1) {   // all lanes of the wave take part in the transposition
    wave_lds_fence();
    rows_to_lds();
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < C::PW; ++k) {
      const int t = i * C::PW + k;
      mem[k] = (t < C::TRI) ? EXP_(t) : (T)0;
    }
#pragma unroll
    for (int w = 0; w < RPL + C::UW; ++w) mem[C::X_OFF + w] = rec[RPL * N + w];
    if constexpr (C::TRI_FOLD) {   // (after the line above: in these lanes the slot's register image is the stale word it was loaded with)
#pragma unroll
      for (int e = 0; e < C::TRI_REM; ++e)
        if (i == C::fold_lane(e)) mem[C::UW_OFF + C::fold_slot(e)] = EXP_(G * C::PW + e);
    }
  }
  if (valid) {
    if constexpr ((PK && G > 1) || EKF_SYM) {
      // mem was filled above / is what the step worked on
    } else if constexpr (PK) {
#pragma unroll
      for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = r; c < N; ++c) mem[C::tri(r, c)] = rec[r * N + c];
#pragma unroll
      for (int w = 0; w < RPL + C::UW; ++w) mem[C::X_OFF + w] = rec[RPL * N + w];
    } else {
#pragma unroll
      for (int w = 0; w < C::RW; ++w) mem[w] = rec[w];
    }
    if constexpr (AB) store_record<C, T, EKF_SYM, true>(a.rec_out + tile * C::TILE_BYTES, lt, mem);   // A -> B tick (StepArgs::rec_out)
    else store_record<C, T, EKF_SYM>(tb, lt, mem);
    if (i == 0) {
      if constexpr (INDEXED) {
        const long slot = slot_of;
        a.t_base[slot] += dtd * n_ticks;
        a.nm_base[slot] += n_has;
      } else {
        if (a.has_meas != nullptr) a.nm_base[entry] += n_has;
      }
    }
  }
  if constexpr (QUERY || INDEXED) {
    bool want = QUERY;
    if constexpr (INDEXED) want = a.o_pose != nullptr;   // uniform
    if (want) {
      T xq[N];
      if constexpr (EKF_SYM) {
#pragma unroll
        for (int r = 0; r < N; ++r) xq[r] = mem[C::X_OFF + r];
      } else if constexpr (G == 1) {
#pragma unroll
        for (int r = 0; r < N; ++r) xq[r] = X_(r);
      } else {
        // local row q of lane i is row (q / KPL) K + (q % KPL) G + i of the state
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < RPL; ++q) EXA_((q / KPL) * K + (q % KPL) * G + i) = X_(q);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < N; ++r) xq[r] = EXA_(r);
      }
      if constexpr (QUERY) {
        if (valid && i == 0)
          sphere_query<M, T>(xq, true, 0.0, 0.0, a.q_origin, a.q_radius, &a.q_delta[entry], a.q_pose ? &a.q_pose[entry * 7] : nullptr);
      } else {
        if (valid && i == 0) write_outputs_row<M, T>(xq, slot_of, a.o_pose, a.o_twist, a.o_acc);
        signal_done(a.done_flag, a.done_seq, lane, a.done_count, a.n, TPW);
      }
    }
  }
#undef P_
#undef X_
#undef UW_
#undef EXA_
#undef EXB_
#undef EXC_
#undef EXP_
}

}  // namespace te
