// target_manager.hpp -- host-side mirror of the reference's TargetManager
// (include/target_estimation/target_manager.hpp:33-203) over device-resident batches.
//
// Same method names, argument meaning, return values and messages as the reference; Eigen types
// are replaced by raw arrays (the reference's Eigen-typed header needs Eigen3, absent here):
//   Vector7d pose/meas  -> const double[7]  [x y z qx qy qz qw]   (target_manager.hpp:60)
//   Vector6d twist/acc  -> const double[6]
//   MatrixXd Q, R, P0   -> row-major const double[n*n] / [m*m]
// Differences, all deliberate:
//   * targets live in HBM, grouped into one Batch per (model, Q, R); the id -> (batch, slot)
//     map is a hash table (id_table.hpp) and enumeration sorts, so it stays ascending by id as
//     with the reference's std::map;
//   * the per-target constructor dump (printInfo, target_interface.cpp:57-78) and the per-target
//     type line (target_manager.cpp:161-173) are printed only when verbose (env
//     TARGET_ESTIMATION_VERBOSE=1): a million-target init must not write a million dumps;
//   * every filter step runs on the GPU; there is no CPU path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <functional>
#include <unordered_map>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "batch_store.hpp"
#include "id_resolve.hpp"
#include "id_table.hpp"

namespace te {

// true if Q, R and the n_P0 covariances have no entry between different axis groups (te_layout.hpp)
bool is_axis_separable(int type, const double* Q, const double* R, const double* P0, long n_P0);

class TargetManager {
 public:
  typedef std::shared_ptr<TargetManager> Ptr;
  // target_manager.hpp:38
  enum target_t { ANGULAR_RATES = 0, ANGULAR_VELOCITIES, UNIFORM_ACCELERATION, UNIFORM_VELOCITY };

  explicit TargetManager(int dtype = F64, int lanes_per_target = 0);
  // throws const char* "TargetManager default constructor failed!" like target_manager.cpp:111-118
  explicit TargetManager(const std::string& file, int dtype = F64, int lanes_per_target = 0);
  virtual ~TargetManager();

  // target_manager.cpp:135-142 (defaults from the YAML file; throws const char* if none loaded)
  void init(unsigned id, double dt0, double t0, const double* p0, const double* v0 = nullptr, const double* a0 = nullptr);
  // target_manager.cpp:144-179
  void init(target_t type, unsigned id, double dt0, double t0, const double* Q, const double* R, const double* P0,
            const double* p0, const double* v0 = nullptr, const double* a0 = nullptr);
  // target_manager.cpp:181-188
  void init(const std::string& file, unsigned id, double dt0, double t0, const double* p0, const double* v0 = nullptr,
            const double* a0 = nullptr);
  bool update(unsigned id, double dt, const double* meas);  // target_manager.cpp:190-202
  bool update(unsigned id, double dt);                      // :204-218
  virtual void update(double dt);                           // :220-225
  bool erase(unsigned id);                                  // :227-241
  bool getTargetPose(unsigned id, double* pose7);           // :252-261
  bool getTargetTwist(unsigned id, double* twist6);         // :263-272
  bool getTargetAcceleration(unsigned id, double* acc6);    // :274-283
  long long getNumberMeasurements(unsigned id);             // :285-295
  // :120-124.  The reference publishes five channels per target through rt_logger (an external ROS package):
  // measurement (measured_pose_), pose (pose_internal_ = [xyz rpy]), twist, acceleration, covariance (the full P),
  // target_interface.cpp:32-40,50-55.  Equivalent observability without ROS: with a log directory set (setLogDirectory
  // or env TARGET_ESTIMATION_LOG_DIR) every log() appends one row per SELECTED target to <dir>/
  //   time_<id>  meas_pose_<id>  est_pose_<id>  est_twist_<id>      the files the reference's test writes and its plot script
  //                                                                  loads (test/target_manager_test.cpp:164-168,
  //                                                                  matlab/plot_target_manager_test.m:9-13)
  //   pose_<id>  est_acc_<id>  covariance_<id>                       the remaining rt_logger channels ([xyz rpy]; acc6; P row-major)
  // in writeTxtFile's text format (utils.hpp:96-120: values separated by one space, one row per line).  Selected =
  // setLogTargets(ids), or every target while the manager holds at most kLogAutoSelect of them; the files stay open
  // between calls and each gets ONE buffered write per call.  A larger population without a selection gets one file per
  // channel, <channel>_all, rows prefixed by the id -- one write per channel per call.  Without a directory: a no-op, as
  // the reference without LOGGER_ON.  Setting a directory switches the measured-pose rows on (setKeepMeasurement).
  void log();
  void setLogDirectory(const std::string& dir);
  void setLogTargets(const unsigned* ids, long n);   // n == 0: back to the automatic selection
  static constexpr long kLogAutoSelect = 64;
  // TargetInterface::getMeasuredPose (target_interface.cpp:117-121): kept only on request (measured_pose.hpp)
  void setKeepMeasurement(bool on);
  bool keepMeasurement() const { return keep_meas_; }
  bool getTargetMeasuredPose(unsigned id, double* pose7);          // false: unknown id or not kept
  // TargetInterface::getPeriodEstimate (target_interface.cpp:80-87): 2 pi / |omega| of the current twist, -1 if not rotating
  bool getTargetPeriodEstimate(unsigned id, double& period);
  // TargetInterface::getEstimatedTransform (target_interface.cpp:95-98): T_ as a row-major 4x4 [R t; 0 1]
  bool getTargetTransform(unsigned id, double* T16);
  // getN() / getM() (target_interface.hpp:142,148)
  bool getTargetDims(unsigned id, int& n, int& m);
  // getTarget(id)->getEstimator()->getQ() / getR() / getP0() (kalman.hpp:74-89), row-major doubles as given at init;
  // any pointer may be null.  false: unknown id (or, for P0 only, more distinct P0 matrices than the host mirror keeps)
  bool getTargetModelMatrices(unsigned id, double* Q, double* R, double* P0);
  std::vector<unsigned> getAvailableTargets();              // :126-133
  bool selectTargetType(const std::string& type_str, target_t& type);  // :52-65

  // TargetInterface getters reached through getTarget(id)-> in the reference
  bool getTargetPoseAt(unsigned id, double t1, double* pose7);      // getEstimatedPose(t)
  bool getTargetTwistAt(unsigned id, double t1, double* twist6);    // getEstimatedTwist(t)
  bool getTargetAccelerationAt(unsigned id, double t1, double* a6); // getEstimatedAcceleration(t)
  bool getTargetTime(unsigned id, double& t);                       // getTime()
  // getTarget(id)->getEstimator()->getState()/getP() (kalman.hpp:69-89); x [n], P [n*n] row-major
  int getTargetState(unsigned id, double* x, double* P);
  bool hasTarget(unsigned id);
  size_t size();

  // ---- batched extension (not in the reference) ----------------------------------------------
  long initBatch(const unsigned* ids, long n, double dt0, double t0, const double* p0, const double* v0, const double* a0);
  long initBatch(target_t type, const unsigned* ids, long n, double dt0, double t0, const double* Q, const double* R,
                 const double* P0, bool per_target_P0, const double* p0, const double* v0, const double* a0);
  // n targets whose (Q, R, P0) come from a table of n_classes parameter sets: class_of[i] is the row of target i
  // (the reference's init takes Q, R, P0 per target, target_manager.hpp:85-87; a table + index is the same thing
  // without n copies).  All classes of one layout share ONE batch, i.e. one launch per tick.
  long initBatchClasses(target_t type, const unsigned* ids, long n, double dt0, double t0, long n_classes, const double* Q,
                        const double* R, const double* P0, const unsigned* class_of, const double* p0, const double* v0,
                        const double* a0);
  long updateBatch(const unsigned* ids, long n, double dt, const double* meas, const unsigned char* has_meas);
  // erase many targets in one call (one compaction launch per batch); unknown or repeated ids are reported
  // like erase() does and skipped; returns the number erased
  long eraseBatch(const unsigned* ids, long n);
  long getPoseBatch(const unsigned* ids, long n, double* pose, double* twist, double* acc, unsigned char* found,
                    bool at_time = false, double t1 = 0.0);
  long getStateBatch(const unsigned* ids, long n, double* x, double* P);
  // IntersectionSolver::getIntersectionTimeWithSphere (src/intersection_solver.cpp:42-89): time from
  // t1 to the first crossing of the sphere, -1 if none or unknown id.
  double getIntersectionTimeWithSphere(unsigned id, double t1, const double* origin, double radius);
  // IntersectionSolver::getIntersectionPoseWithSphere without its moving-average convergence gate
  // (src/intersection_solver.cpp:91-104): true if an intersection exists; pose7 = pose at t1+delta.
  bool getIntersectionPoseWithSphere(unsigned id, double t1, const double* origin, double radius, double* pose7,
                                     double* delta = nullptr);
  // IntersectionSolver::getIntersectionPoseWithSphere with its convergence gate, same argument order
  // (intersection_solver.hpp:98-101); one gate per target (the reference has one per solver object).
  // Returns whether the filtered position / angle errors are below the thresholds.
  bool getIntersectionPoseWithSphere(unsigned id, double t1, double pos_th, double ang_th, const double* origin,
                                     double radius, double* pose7);
  void setIntersectionFiltersLength(int n) { filters_length_ = n; }   // IntersectionSolver ctor, default 250
  long intersectGatedBatch(const unsigned* ids, long n, double t1, double pos_th, double ang_th, const double* origin,
                           double radius, double* delta, double* pose, unsigned char* converged, unsigned char* found,
                           double* filt = nullptr);
  long intersectBatch(const unsigned* ids, long n, double t1, const double* origin, double radius, double* delta,
                      double* pose, unsigned char* found);

  // n_ticks ticks of EVERY batch (specs in batch order), device-resident inputs: one step launch per
  // batch per tick, optionally followed by the own-time sphere query of every target.  The batches are
  // independent, so with use_graph != 0 each batch's chain of launches is its own branch of one hipGraph
  // and the branches run concurrently; the query runs inside the step kernel (QUERY variants).
  // use_graph == 2 records without launching.  use_graph == 0 issues the same launches eagerly, batch after batch per tick.
  bool populationTickNow() { std::lock_guard<std::mutex> lg(target_lock_); return populationTick(); }
  void stepSequenceAll(long n_ticks, double dt, const Batch::SeqSpec* specs, long n_specs, bool query,
                       const double* origin, double radius, int use_graph);

  // Resident ("live") mode for EVERY batch of the manager at once (Batch::live_start per batch, each kernel on its own
  // stream so that they are resident together): BASELINE configs[3] / configs[4] put two motion models on every GPU, and
  // their per-GPU share (62 500 + 62 500 targets) is launch-bound.  specs as for stepSequenceAll (ring_ticks > 0 required).
  // The wavefronts of all sessions must fit the device together: sum over batches of waves / capacity <= 1.
  // query: also the own-time sphere query of every target after every tick into specs[b].delta_dev / pose_dev (configs[4])
  void liveStartAll(double dt, const Batch::SeqSpec* specs, long n_specs, long first_entry, long max_ticks, double idle_limit_s,
                    bool query = false, const double* origin = nullptr, double radius = 0.0);
  void livePostAll(long n_ticks, bool one_doorbell_per_tick);
  long liveDoneAll();                       // ticks every wavefront of every batch has finished
  bool liveWaitAll(long tick, double timeout_s);
  long liveStopAll();                       // returns the ticks served (the same for every batch)
  int numBatches() const { return (int)batches_.size(); }
  Batch* batch(int i) { return batches_[(size_t)i].get(); }
  Batch* batchOfType(int type);
  void setStream(hipStream_t s);
  hipStream_t stream() const { return stream_; }
  // pose7 rows (doubles) of every target, batch after batch in slot order, into out_dev [size()][7] on stream `st`
  // (the gather's send side, pose_gather.hpp); out_dev == null only counts.  Returns the number of rows.
  long posesToDevice(double* out_dev, long capacity, hipStream_t st);
  // The gather's enqueue step, atomic with respect to init / erase / setStream: under the manager's lock, checks that the
  // manager holds expect_rows rows, calls prepare(rows, stream) -- which may enqueue waits on that stream and returns the
  // destination [rows][7] -- and launches the outputs kernels into it.  Returns the row count.
  long posesForGather(long expect_rows, const std::function<double*(long, hipStream_t)>& prepare);
  void synchronize();
  int dtype() const { return dtype_; }
  bool defaultsLoaded() const { return default_values_loaded_; }
  int defaultType() const { return (int)default_type_; }

 protected:
  using Loc = TargetLoc;
  bool loadYamlFile(const std::string& file, std::vector<double>& Q, std::vector<double>& R, std::vector<double>& P,
                    target_t& type);  // target_manager.cpp:67-104
  // lanes code of a new target's batch: the manager's explicit choice, or (auto) the axis-separable
  // layout when Q, R and every P0 allow it
  int chooseLayout(int type, const double* Q, const double* R, const double* P0, long n_P0) const;
  // the batch of (model, layout) -- created on first use -- and the parameter class of (Q, R) inside it
  int findOrCreateBatch(int type, const double* Q, const double* R, int lanes_code, int& cls);
  bool find(unsigned id, Loc& loc);

  IdTable targets_;   // id -> (batch, slot); the reference's std::map<unsigned, TargetPtr> (target_manager.hpp:201)
  std::vector<std::unique_ptr<Batch>> batches_;
  std::mutex target_lock_;
  std::vector<double> default_Q_, default_P_, default_R_;
  target_t default_type_ = UNIFORM_VELOCITY;
  bool default_values_loaded_ = false;
  int dtype_, lanes_;
  hipStream_t stream_ = nullptr;
  bool verbose_ = false;
  int filters_length_ = 250;
  std::string log_dir_;
  bool keep_meas_ = false;
  std::vector<unsigned> log_ids_;                       // explicit selection (sorted); empty = automatic
  struct LogFiles { std::FILE* f[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; };
  std::unordered_map<unsigned, LogFiles> log_files_;    // per selected target, kept open
  std::FILE* log_all_[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void closeLogFiles();
  // recorded all-batches sequences (stepSequenceAll)
  struct SeqGraph {
    long n_ticks; double dt; bool query; double origin[3]; double radius;
    std::vector<Batch::SeqSpec> specs;
    std::vector<Batch::DevIdentity> ident;
    hipGraph_t graph; hipGraphExec_t exec;
  };
  std::vector<SeqGraph> seq_graphs_;
  std::vector<hipStream_t> branch_streams_;   // [0]: the capture stream
  bool populationTick() const;   // the tick of all batches as one launch (kf_population.hpp); caller holds target_lock_
  void enqueuePopulationTick(hipStream_t st, long s, double dt, const Batch::SeqSpec* specs, bool query, const double* origin, double radius,
                             bool reverse, bool ab);
  std::vector<hipEvent_t> branch_events_;
  void dropSeqGraphs();
  // device-side id resolution for the array-of-ids calls (id_resolve.hpp): the table and the staging of one call
  struct DevIds {
    unsigned* keys = nullptr; unsigned* vals = nullptr; int* seen = nullptr;
    int log2cap = 0; bool dirty = true; int epoch = 0;
    long cap = 0;                       // entries the staging holds
    unsigned* ids = nullptr; int* loc = nullptr; int* idx = nullptr;
    double* aos = nullptr; void* soa = nullptr; unsigned char* mask = nullptr; unsigned char* found = nullptr;
    double* out = nullptr;              // [cap][7 + 6 + 6] getter outputs
    ResolveCounters* counters = nullptr;
    ResolveCounters* h_counters = nullptr;   // pinned
  } dev_ids_;
  static constexpr long kDevResolveMin = 8192;   // below this the host table is faster than the extra launches
  static constexpr long kSmallBatchQueue = 1024; // host-array calls of at most this many targets go through the one-target queue (updateBatch)
  bool smallBatchPath(const unsigned* ids, long n) const;
  void devIdsReserve(long n);
  void devIdsRebuild();
  // loc[e] of every id on the device + the per-batch counts on the host; false: not applicable (too many batches)
  bool resolveOnDevice(const unsigned* ids, long n, ResolveCounters& out);
  void devIdsFree();
  bool seq_flip_ = false;   // zig-zag across the whole tick: the next eager all-batches tick runs last batch first, tiles backwards
};

}  // namespace te
