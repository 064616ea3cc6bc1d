// batch_store.hpp -- device-resident store of all targets of one motion model / parameter set /
// precision, and the launches that act on it.  Replaces, for N targets at once, the per-target
// objects of the reference (TargetInterface + its KalmanFilterInterface:
// include/target_estimation/target_interface.hpp:193-282, kalman.hpp:95-150): instead of
// (x, P, A, C, Q, R, P0, K, I) per target there is one shared (Q, R) and one lane record
// (x, P, unwrap memory) per target in HBM (te_layout.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "kf_ops.hpp"

namespace te {

class Batch {
 public:
  // `owner_lock`: the mutex of the manager that owns the batch (TargetManager::target_lock_); the C boundary takes it
  // around every call made through a batch handle, so that those calls and the manager's own are serialised
  Batch(int type, int dtype, int lanes, const double* Q, const double* R, hipStream_t stream, std::mutex* owner_lock = nullptr);
  ~Batch();
  Batch(const Batch&) = delete;
  Batch& operator=(const Batch&) = delete;

  std::mutex* owner_lock() const { return owner_lock_; }
  int type() const { return type_; }
  int lanes_code() const { return lanes_code_; }
  int dtype() const { return dtype_; }
  int n_state() const { return ops_->L.n; }
  int n_meas() const { return ops_->L.m; }
  const LayoutInfo& layout() const { return ops_->L; }
  long size() const { return n_; }
  bool getter_table_is_cheap() const { return n_ <= kCacheMax; }   // (its one-target getters read a host-resident table from the first call after a change)
  size_t elem_size() const { return dtype_ == F64 ? 8 : 4; }
  // Parameter classes: the (Q, R) pairs of the batch's targets (TargetManager::init takes them per target,
  // target_manager.hpp:85-87).  One class: Q, R are shared by every lane (scalar cache / LDS).  Several: a table in
  // HBM plus one class index per slot, and the per-class kernels (PERQR) -- still one launch per tick.
  int find_class(const double* Q, const double* R) const;   // -1: not there
  int add_class(const double* Q, const double* R);
  int n_classes() const { return n_classes_; }
  void set_stream(hipStream_t s) { stream_ = s; }
  hipStream_t stream() const { return stream_; }
  unsigned slot_id(long slot) const { return slot_ids_[slot]; }
  const std::vector<unsigned>& slot_ids() const { return slot_ids_; }
  double clock() const { return t_acc_; }

  // Construct `count` new targets in slots [size(), size()+count).  Host arrays.
  // cls: the class of every new target, or cls_of [count] one class each; P0_index [count] (with P0 = a table of
  // P0_count matrices) gives every target the initial covariance of its row of the table.
  long append(long count, const unsigned* ids, double t0, const double* P0, bool per_target_P0,
              const double* p0, const double* v0, const double* a0, int cls = 0, const int* cls_of = nullptr,
              const int* P0_index = nullptr, long P0_count = 0);
  // Remove a slot by moving the last slot into it; returns the id that now lives in `slot`
  // (or the erased id if it was the last one).
  unsigned erase_slot(long slot);
  // Erase many slots at once (distinct slots, any order): survivors from the tail fill the holes below
  // the new size in ONE launch.  moves_out lists (id, new slot) of every target that changed slot.
  void erase_slots(const int* slots, long k, std::vector<std::pair<unsigned, int>>& moves_out);

  // One tick over every target, device-resident inputs (the fast path).
  //   meas_dev: SoA [7][ld] in the batch precision, or null (predict only = TargetInterface::update)
  //   has_dev : per-slot mask or null (all have a measurement)
  void step_dense(double dt, const void* meas_dev, long ld, const unsigned char* has_dev);
  // n_ticks consecutive dense ticks, tick s reading meas_base + s * tick_stride (elements of the
  // batch precision) and has_base + s * has_stride: exactly n_ticks launches of the step kernel,
  // enqueued from C++ (use_graph 1: recorded once into a hipGraph and replayed, which removes the
  // per-launch host cost when a recorded stream is replayed; 2: record only, launch nothing).
  void step_sequence(long n_ticks, double dt, const void* meas_base, long tick_stride, long ld,
                     const unsigned char* has_base, long has_stride, int use_graph, long ring_ticks = 0);
  // n_ticks ticks in ONE launch: every target's state stays in registers across the ticks and only
  // the measurements are read per tick.  Same results as n_ticks single ticks; a different
  // ("effective", temporally fused) cost model -- for replaying recorded streams.
  void step_fused(long n_ticks, double dt, const void* meas_base, long tick_stride, long ld,
                  const unsigned char* has_base, long has_stride);
  // RESIDENT ("live") mode for small batches: ONE launch stays on the device with the batch's state in registers and serves
  // tick after tick as the host posts them -- no per-tick dispatch (a dependent launch costs 1.5-2 us, more than the tick of a
  // 10^4-target batch itself).  Protocol:
  //   live_start(dt, ring...)   launch; the ring (device memory, SoA ticks as for step_sequence) receives tick k's measurements
  //                             in entry (first_entry + k) % ring_ticks BEFORE tick k is posted
  //   live_post(n)              "n more ticks are in the ring": one store to a host-mapped word the wavefronts poll
  //   live_done()               ticks every wavefront has finished (one host-mapped word, kept by the kernel's relay wavefront)
  //   live_wait(tick, timeout)  spin until live_done() >= tick
  //   live_stop()               ask the kernel to finish the posted ticks, store the records and exit; accounts the ticks
  // Results are those of single ticks, bit for bit.  While a session is open the records in HBM are stale; every other call
  // on the batch (steps, getters, erase, ...) ends the session first.  The whole grid must be resident (the wavefronts wait
  // for the host, not for each other, but one that never starts would miss its ticks): refused beyond live_capacity().
  // A wavefront that sees no news for idle_limit_s gives up on its own (a dead host leaves no kernel behind).
  // q_delta_dev [size] (+ q_pose_dev [size][7] or null): also run the own-time sphere query of every target after every tick
  // (outputs overwritten every tick), as enqueue_tick's fused query does
  void live_start(double dt, const void* meas_ring, long tick_stride, long ld, const unsigned char* has_ring, long has_stride,
                  long ring_ticks, long first_entry, long max_ticks, double idle_limit_s, const double* q_origin = nullptr,
                  double q_radius = 0.0, double* q_delta_dev = nullptr, double* q_pose_dev = nullptr);
  // pose_soa_dev [7][ld] doubles (device memory; null switches it off): the sessions started AFTER this call also write the
  // estimated pose of every target after every tick there (the reference's node publishes them every tick,
  // src/target_manager_ros.cpp:78-87), through the caches and before the tick counts as done -- a consumer that copies the
  // buffer on another stream after live_done() reached tick k reads the poses of a tick >= k (of tick k if it posts one tick at a time)
  void live_set_pose_output(double* pose_soa_dev, long ld);
  void live_post(long n_ticks);
  long live_done() const;
  bool live_wait(long tick, double timeout_s) const;
  long live_stop();            // returns the ticks served
  bool live_active() const { return live_.active; }
  // targets a session can hold; with_outputs: one with the per-tick query or pose output (a larger kernel: fewer)
  long live_capacity_targets(bool with_outputs = false) const { return ops_->live_capacity ? ops_->live_capacity(with_outputs ? 1 : 0) * ops_->L.tpw : 0; }
  bool live_pose_output_set() const { return live_.pose_out != nullptr; }
  bool live_running() const { return live_.active && __atomic_load_n(live_.h_done + 2, __ATOMIC_ACQUIRE) == 0; }   // the relay has not left (its last store)
  // One tick over the listed slots, host inputs (meas rows follow the order of `slots`).
  void step_indexed(const int* slots, long n, double dt, const double* meas_aos, const unsigned char* has);
  // The same with everything already on the device: idx_dev [n] = slot of entry e or a negative number (entry skipped);
  // meas_soa_dev SoA [7][ld] in the batch precision, rows in entry order.  Asynchronous.
  void step_indexed_dev(const int* idx_dev, long n, double dt, const void* meas_soa_dev, long ld, const unsigned char* has_dev);
  // derived outputs of the listed entries into device arrays (row e; entries with a negative slot are left untouched)
  void outputs_indexed_dev(const int* idx_dev, long n, double* pose_dev, double* twist_dev, double* acc_dev, bool at_time, double t1);
  // One-target call of the reference's C ABI (target_manager_update / update_meas).  The step is
  // QUEUED: it runs, together with every other queued one-target step, as a single indexed launch
  // when anything reads or otherwise touches the batch (flush()).  A slot queued twice flushes first,
  // so the order of steps per target is the caller's order.
  void step_one(long slot, double dt, const double* meas7);
  void flush();
  // One tick over every slot in slot order, host inputs (rows of meas_aos / has follow the slot order)
  void step_dense_host(double dt, const double* meas_aos, const unsigned char* has);
  // The same from SoA host rows in the BATCH precision (row c of meas_soa = component c of every slot,
  // ld_host elements apart): only the rows the model reads cross PCIe (3 for the linear models, 7 for the
  // angular ones) and no conversion kernel runs.  Pinned host memory makes the copies asynchronous DMA.
  void step_dense_host_soa(double dt, const void* meas_soa, long ld_host, const unsigned char* has);

  // Derived outputs to host arrays; slots == null means all slots in order.
  void outputs(const int* slots, long n, double* pose, double* twist, double* acc, bool at_time, double t1);
  // Dense, to device arrays [size()][7] / [size()][6] / [size()][6] (doubles).
  void outputs_dev(double* pose_dev, double* twist_dev, double* acc_dev, bool at_time, double t1);
  void outputs_one(long slot, double* pose, double* twist, double* acc, bool at_time, double t1);
  // AoS doubles [n][7] on device -> SoA [7][ld] in the batch precision on device
  void pack_meas_dev(const double* aos_dev, long n, void* soa_dev, long ld);

  // IntersectionSolver::getIntersectionTime/PoseWithSphere (src/intersection_solver.cpp:42-104)
  // for the listed slots (host arrays; slots == null: all) or, _dev, every slot to device arrays.
  // t1 is absolute; NaN means each target's own current time.
  void intersect(const int* slots, long n, double t1, const double* origin, double radius, double* delta, double* pose);
  void intersect_dev(double t1, const double* origin, double radius, double* delta_dev, double* pose_dev);

  // the same with the reference's per-solver convergence gate kept per target (intersect_gate.hpp);
  // converged [n] bytes; filt [n][2] (filtered position / angle error) optional.  `window` is the
  // reference's filters_length (intersection_solver.hpp:63, default 250); changing it resets the gates.
  void intersect_gated(const int* slots, long n, double t1, const double* origin, double radius, double pos_th,
                       double ang_th, int window, double* delta, double* pose, unsigned char* converged, double* filt);
  void intersect_gated_dev(double t1, const double* origin, double radius, double pos_th, double ang_th, int window,
                           double* delta_dev, double* pose_dev, unsigned char* converged_dev);
  // the gate alone, fed with query results that are already on the device (e.g. the outputs of the per-tick query
  // fused into the step kernels): every slot, delta [size], pose [size][7]; filt / var [size][2] optional
  void gate_update_dev(const double* delta_dev, const double* pose_dev, double pos_th, double ang_th, int window,
                       unsigned char* converged_dev, double* filt_dev, double* var_dev);

  // Building blocks of the manager's all-batches sequence (TargetManager::stepSequenceAll): enqueue
  // n_ticks ticks -- each optionally followed by the own-time sphere query of every slot -- on `st`
  // without touching the batch clock, then account for them.
  struct SeqSpec {
    const void* meas_base; long tick_stride; long ld;   // tick s reads meas_base + s*tick_stride elements, SoA [7][ld]
    const unsigned char* has_base; long has_stride;     // optional masks
    double* delta_dev; double* pose_dev;                // query outputs [size] / [size][7] (overwritten every tick)
    long ring_ticks;                                    // > 0: the measurements are a ring, tick s reads entry s % ring_ticks
  };
  // tick s of the spec on `st`, without touching the batch clock.  With query: the own-time sphere
  // query of every slot runs inside the step kernel (one launch).
  // ab: 1 = an A -> B tick (records read from the current buffer, written to the alternate one, buffers swapped; never
  // inside a stream capture: a recorded graph bakes its pointers), 0 = in place.
  void enqueue_tick(hipStream_t st, long s, double dt, const SeqSpec& spec, bool query, const double* origin, double radius,
                    bool reverse = false, bool ab = false);
  // The same tick as launch parameters only, for the manager's one-launch population tick (kf_population.hpp): possible when
  // population_ready() -- separable layout with packed groups, one (Q, R) class, no measured-pose rows to keep.  With ab the
  // caller calls swap_records() once the launch is queued (and only if the returned parameters carry rec_out).
  bool population_ready() const;
  StepParams tick_params(long s, double dt, const SeqSpec& spec, bool query, const double* origin, double radius, bool ab);
  void swap_records() { std::swap(d_rec_, d_rec_alt_); }
  void account_sequence(long n_ticks, double dt, bool all_measured);
  // identity of everything a recorded launch sequence refers to
  struct DevIdentity { const void* rec; const void* qr; const void* tbase; const void* nmbase; long n; };
  DevIdentity dev_identity() const { return DevIdentity{d_rec_, d_qr_, d_tbase_, d_nmbase_, n_}; }
  void prepare() { touch(); }

  // TargetInterface::getMeasuredPose (target_interface.cpp:117-121), optional: see measured_pose.hpp
  void set_keep_measurement(bool on);
  bool keep_measurement() const { return keep_meas_; }
  void measured_poses(const int* slots, long n, double* pose7_rows);   // host rows [n][7]; slots == null: all
  // KalmanFilterInterface::getQ / getR / getP0 of one target (kalman.hpp:74-89): the matrices it was created with, as
  // given (doubles, row-major).  P0 is kept on the host as a table of the distinct matrices seen (up to kP0TableMax;
  // beyond that initial_covariance() returns false).
  void class_matrices(long slot, double* Q, double* R);
  bool initial_covariance(long slot, double* P0);
  void get_state(const int* slots, long n, double* x, double* P);
  void set_state(const int* slots, long n, const double* x, const double* P, const double* unwrap);
  long long n_measurements(long slot);
  double time(long slot);
  void times(double* out);   // filter time of every slot (host array [size()])
  void synchronize();

  // bytes of HBM one predict+update cycle must move for one target (state read+write + the
  // measurement words the model reads); used by the roofline accounting
  long algorithmic_bytes_per_cycle() const;
  long state_bytes() const { return (n_ + ops_->L.tpw - 1) / ops_->L.tpw * ops_->L.tile_bytes; }
  static long zigzag_min_bytes();   // default 128 MB (env TE_ZIGZAG_MIN_MB)
  // A -> B ticks with nontemporal stores (kf_step.hpp StepArgs::rec_out) instead of in-place read-modify-write: the policy of
  // eager dense ticks once the state is this large (default 1536 MB, env TE_PINGPONG_MIN_MB; a negative value switches it off).
  // Below it the zig-zag in place wins (the Infinity Cache still holds a useful share of the state); costs a second record buffer.
  static long pingpong_min_bytes();
  bool pingpong() const { return pingpong_min_bytes() >= 0 && state_bytes() >= pingpong_min_bytes(); }
  char* records_dev() const { return d_rec_; }

 private:
  void reserve(long n);
  void stage_reserve(long n);
  void upload_slots(const int* slots, long n);

  int type_, dtype_, lanes_code_;
  std::mutex* owner_lock_ = nullptr;
  const Ops* ops_;
  hipStream_t stream_;
  std::unordered_map<std::string, int> class_index_;   // raw bytes of [Q | R] -> class
  int n_classes_ = 0, qr_cap_ = 0;
  void* d_qr_ = nullptr;           // [qr_cap_][N*N + K*K] in the batch precision
  int* d_cls_ = nullptr;           // [cap_] class of every slot
  bool flip_ = false;              // the next dense tick walks the tiles backwards (zig-zag, kf_step.hpp StepArgs::reverse)
  // Zig-zag only pays when the state does not fit the Infinity Cache, and it costs when the state is L2-resident: a tile
  // walked backwards lands on another XCD, whose L2 does not hold it (10^5 UA fp32: 4.7 -> 8.2 us per tick).
  bool zigzag() const { return state_bytes() >= zigzag_min_bytes(); }
  StepParams base_params() const;
  void launch_step(const StepParams& p, hipStream_t st, int meas_rows = 7);   // ops_->step + the measured-pose rows when kept
  bool keep_meas_ = false;
  int host_meas_rows_ = 7;         // measurement rows the host SoA path transported for the tick being enqueued
  double* d_lastmeas_ = nullptr;   // [cap_][7] doubles (measured_pose.hpp), null unless keep_meas_
  std::vector<std::vector<double>> class_qr_;   // [Q | R] of every class as given (doubles)
  static constexpr size_t kP0TableMax = 4096;
  std::vector<std::vector<double>> p0_tab_;
  std::unordered_map<std::string, int> p0_index_;
  std::vector<int> slot_p0_;       // slot -> row of p0_tab_
  bool p0_kept_ = true;
  int intern_p0(const double* P0);
  char* d_rec_ = nullptr;          // the CURRENT records (A -> B ticks swap it with d_rec_alt_ after every launch)
  char* d_rec_alt_ = nullptr;      // second record buffer of the same capacity, allocated on the first A -> B tick
  char* alt_records();             // null if the device has no room for it (the batch then stays in place)
  bool alt_failed_ = false;
  double* d_tbase_ = nullptr;
  int* d_nmbase_ = nullptr;
  long cap_ = 0;  // slots
  long n_ = 0;
  std::vector<unsigned> slot_ids_;
  double t_acc_ = 0.0;      // batch clock: target time = t_base[slot] + t_acc_
  long long nm_acc_ = 0;    // batch measurement counter
  // staging for the host-array paths
  long stage_cap_ = 0;
  int* d_idx_ = nullptr;
  double* d_aos_ = nullptr;        // [stage_cap][19] doubles: inputs (p0|v0|a0, meas) and outputs
  void* d_meas_ = nullptr;         // SoA [7][stage_cap] in the batch precision
  unsigned char* d_mask_ = nullptr;
  double* d_P0_ = nullptr;
  long P0_cap_ = 0;
  // convergence gates (allocated on first use)
  int gate_window_ = 0;
  long gate_cap_ = 0;
  double* d_gate_ring_ = nullptr;
  double* d_gate_sum_ = nullptr;
  int* d_gate_state_ = nullptr;
  double* d_gate_prev_ = nullptr;
  void gate_reserve(int window);
  void gate_move(long src, long dst);
  void gate_reset(long first, long count);
  struct GraphEntry {
    long n_ticks, tick_stride, ld, has_stride, n;
    double dt;
    const void* meas_base;
    const unsigned char* has_base;
    char* rec;
    hipGraphExec_t exec;
    hipGraph_t graph;
    long ring_ticks;
    unsigned long last_use = 0;
  };
  std::vector<GraphEntry> graphs_;
  unsigned long graph_clock_ = 0;   // recorded sequences are evicted least-recently-used first (64 kept)
  hipStream_t cap_stream_ = nullptr;
  void drop_graphs();
  // queue of one-target creations (append with count == 1): consecutive ones with the same (t0, P0, class) become one init launch
  struct InitQueue { std::vector<unsigned> ids; std::vector<double> p0, v0, a0, P0; double t0 = 0.0; int cls = 0; long first = 0; } initq_;
  void flush_inits();
  long append_now(long first, long count, const unsigned* ids, double t0, const double* P0, bool per_target_P0, const double* p0,
                  const double* v0, const double* a0, int cls, const int* cls_of, const int* P0_index, long P0_count, bool bookkeeping);
  // queue of one-target steps (reference C ABI) and the cache that serves the one-target getters
  struct Pending { int slot; unsigned char has; double dt; double meas[7]; };
  std::vector<Pending> pending_;
  std::vector<unsigned char> pending_mark_;
  double* d_dtper_ = nullptr;
  // Both live in pinned, device-mapped host memory: the kernels read the queued inputs and write the
  // outputs there directly, so a flush needs no copies; when the getter table is current and the queue holds at most
  // kFusedFlushMax targets it is ONE launch -- the step kernel writes the stepped rows and then a completion flag -- and the
  // host does not synchronise the stream but spins on that flag (wait_done; tools/launch_latency.hip: launch + flag 6.7 us
  // against 13.3 us for two launches and the runtime's wait, which is what longer queues still pay).
  void pin_reserve(long k);
  char* h_pin_ = nullptr;                    // idx int[cap] | dt double[cap] | meas T[7][cap] | has uchar[cap]
  char* d_pin_ = nullptr;                    // the same memory as the device sees it
  char* bar_pin_ = nullptr;                  // the same sections for flushes of up to one wavefront, in fine-grained device memory behind the PCIe BAR (flush)
  bool bar_pin_failed_ = false;
  long pin_cap_ = 0;
  // The getter table.  Batches up to kCacheMax build it at the first one-target getter after a change.  A larger batch
  // (up to kCacheBigMax) builds it only for a caller that really sweeps it target by target: the first kBigDirect getters
  // after a change are served one by one (a launch and a copy each), the next one builds the table; a batch that was swept
  // last time builds it at once, one that was not (a dense step followed by a single getter) goes back to single reads.
  static constexpr long kCacheMax = 16384;
  static constexpr long kCacheBigMax = 1L << 22;   // 19 doubles per target: 640 MB of pinned host memory at most
  static constexpr int kBigDirect = 64;
  long epoch_getters_ = 0;                   // one-target getters since the last change
  bool big_sweeps_ = false;                  // the last epoch of this (large) batch was a sweep
  static constexpr size_t kStateScratchKeep = 16u << 20;   // get_state's device scratch kept between calls up to this size
  char* d_state_scratch_ = nullptr;
  size_t state_scratch_bytes_ = 0;
  static constexpr int kCounterDirect = 4;   // single reads of a measurement counter before all of them are copied to the host
  std::vector<int> h_nm_;
  bool nm_valid_ = false;
  int nm_reads_ = 0;
  void cache_reserve(long n);
  double* h_cache_ = nullptr;                // [n][7] pose | [n][6] twist | [n][6] acceleration
  double* d_cache_ = nullptr;
  long cache_cap_ = 0;
  bool cache_valid_ = false;
  int* h_done_ = nullptr;                    // completion flag of the last flush (host-mapped), and its device alias
  int* d_done_ = nullptr;
  int* d_done_count_ = nullptr;              // device word: wavefronts of a multi-wavefront flush that have written their rows (StepArgs::done_count)
  static constexpr long kFusedFlushMax = 1024;   // up to this many queued targets the flush is ONE launch (flush)
  unsigned done_seq_ = 0;
  void wait_done(int seq);                   // spin on *h_done_ == seq, falling back to hipStreamSynchronize
  void touch() {                             // call before anything that changes state
    flush();
    cache_valid_ = false;
    nm_valid_ = false; nm_reads_ = 0;
    if (epoch_getters_ > 0) big_sweeps_ = epoch_getters_ > kBigDirect;   // (changes with no getter in between keep the verdict)
    epoch_getters_ = 0;
  }
  static constexpr double kLiveStartTimeoutS = 2.0;
  struct Live {
    bool active = false;
    bool zombie = false;             // launched, never seen running, told to stop: synchronise its stream before touching records
    char* h_block = nullptr;         // host-mapped block: [0] the doorbell unless it lives behind the BAR, [64] the relay's words
    long long* bar_bell = nullptr;   // the doorbell in fine-grained DEVICE memory, written by the host through the PCIe BAR (or null)
    long long* h_posted = nullptr;   // the doorbell as the host writes it (count | stop bit)
    long long* d_posted = nullptr;   // the same block as the device sees it
    int* h_done = nullptr;
    int* d_done = nullptr;
    char* d_block = nullptr;         // device memory: [mirror word, padded to 64 B][progress words of the wavefronts]
    double* pose_out = nullptr;      // per-tick pose output (live_set_pose_output)
    long pose_ld = 0;
    hipStream_t stream = nullptr;    // the resident kernel's own stream
    hipEvent_t ready = nullptr;
    long waves = 0, cap_waves = 0;
    long posted = 0, max_ticks = 0;
    double dt = 0.0;
    bool all_measured = false;
    double share = 0.0;              // > 0 while the session is counted among the process's resident sessions (hip_check.hpp)
  } live_;
  void live_release();
};

}  // namespace te
