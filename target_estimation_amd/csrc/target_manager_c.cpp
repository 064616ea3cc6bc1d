// target_manager_c.cpp -- extern "C" boundary over te::TargetManager.
// The first ten functions are the reference's C wrapper (src/target_manager_c.cpp:15-76,
// declared in include/target_estimation/target_manager_c.h:28-37); the rest is the batched
// extension declared in include/target_estimation_amd/target_batch_c.h.
#include <cstdio>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/target_estimation_amd/target_batch_c.h"
#include "intersection_solver.hpp"
#include "measurement_ingest.hpp"
#include "pose_gather.hpp"
#include "stream_gen.hpp"
#include "target_manager.hpp"

using te::Batch;
using te::TargetManager;

namespace {
thread_local std::string g_last_error;

void set_error(const char* where, const char* what) {
  g_last_error = std::string(where) + ": " + what;
  std::fprintf(stderr, "[target_estimation_amd] %s\n", g_last_error.c_str());
}

// run f(); map the reference's thrown `const char*` / std::exception to an error code
template <class F>
int guarded(const char* where, F&& f) {
  try {
    f();
    return 0;
  } catch (const char* msg) {
    set_error(where, msg);
  } catch (const std::exception& e) {
    set_error(where, e.what());
  } catch (...) {
    set_error(where, "unknown exception");
  }
  return -1;
}

// f()'s value, or `fallback` (with the error recorded) if it throws
template <class R, class F>
R guarded_value(const char* where, R fallback, F&& f) {
  R out = fallback;
  guarded(where, [&] { out = f(); });
  return out;
}

// a NULL handle is reported like any other error (the reference dereferences it)
inline TargetManager* M(const target_manager_c* self) {
  if (!self) throw std::invalid_argument("NULL manager handle");
  return (TargetManager*)self;
}
inline Batch* B(target_batch_c* b) {
  if (!b) throw std::invalid_argument("NULL batch handle");
  return (Batch*)b;
}
// Calls through a batch handle take the owning manager's mutex (the same one every TargetManager method takes), so a
// thread on the ten-symbol ABI and a thread driving a batch handle are serialised; Batch itself is not locked.
struct BatchLock {
  std::unique_lock<std::mutex> l;
  explicit BatchLock(Batch* b) { if (b->owner_lock()) l = std::unique_lock<std::mutex>(*b->owner_lock()); }
};
inline te::MeasurementIngest* I(target_ingest_c* i) {
  if (!i) throw std::invalid_argument("NULL ingest handle");
  return (te::MeasurementIngest*)i;
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- the reference's ten symbols
target_manager_c* target_manager_new(const char* file) {
  return target_manager_new_ex(file, TARGET_DTYPE_F64, 0);
}

void target_manager_init(const target_manager_c* self, const unsigned int id, const double dt0, double p0[], const double t0) {
  guarded("target_manager_init", [&] { M(self)->init(id, dt0, t0, p0); });
}

void target_manager_update_meas(const target_manager_c* self, const unsigned int id, const double dt, double meas[]) {
  guarded("target_manager_update_meas", [&] { M(self)->update(id, dt, meas); });
}

void target_manager_update(const target_manager_c* self, const unsigned int id, const double dt) {
  guarded("target_manager_update", [&] { M(self)->update(id, dt); });
}

bool target_manager_get_est_pose(const target_manager_c* self, const unsigned int id, double pose[]) {
  bool res = false;
  guarded("target_manager_get_est_pose", [&] { res = M(self)->getTargetPose(id, pose); });
  return res;
}

bool target_manager_get_est_twist(const target_manager_c* self, const unsigned int id, double twist[]) {
  bool res = false;
  guarded("target_manager_get_est_twist", [&] { res = M(self)->getTargetTwist(id, twist); });
  return res;
}

bool target_manager_get_est_acceleration(const target_manager_c* self, const unsigned int id, double acceleration[]) {
  bool res = false;
  guarded("target_manager_get_est_acceleration", [&] { res = M(self)->getTargetAcceleration(id, acceleration); });
  return res;
}

int target_manager_get_n_measurements(const target_manager_c* self, const unsigned int id) {
  int n = 0;
  guarded("target_manager_get_n_measurements", [&] { n = (int)M(self)->getNumberMeasurements(id); });
  return n;
}

void target_manager_log(const target_manager_c* self) {
  guarded("target_manager_log", [&] { M(self)->log(); });
}

void target_manager_delete(target_manager_c* self) {
  if (self) guarded("target_manager_delete", [&] { delete M(self); });
}

// ---------------------------------------------------------------- batched extension
target_manager_c* target_manager_new_ex(const char* file, int dtype, int lanes_per_target) {
  TargetManager* m = nullptr;
  guarded("target_manager_new", [&] {
    m = file ? new TargetManager(std::string(file), dtype, lanes_per_target) : new TargetManager(dtype, lanes_per_target);
  });
  return (target_manager_c*)m;
}

int target_manager_set_stream(target_manager_c* self, void* hip_stream) {
  return guarded("target_manager_set_stream", [&] { M(self)->setStream((hipStream_t)hip_stream); });
}

int target_manager_set_log_directory(target_manager_c* self, const char* dir) {
  return guarded("target_manager_set_log_directory", [&] { M(self)->setLogDirectory(dir ? dir : ""); });
}

int target_manager_synchronize(target_manager_c* self) {
  return guarded("target_manager_synchronize", [&] { M(self)->synchronize(); });
}

const char* target_manager_last_error(void) { return g_last_error.c_str(); }

int target_manager_init_typed(target_manager_c* self, int type, unsigned int id, double dt0, double t0, const double* Q,
                              const double* R, const double* P0, const double* p0, const double* v0, const double* a0) {
  return guarded("target_manager_init_typed", [&] {
    M(self)->init((TargetManager::target_t)type, id, dt0, t0, Q, R, P0, p0, v0, a0);
  });
}

long target_manager_init_batch(target_manager_c* self, const unsigned int* ids, long n, double dt0, double t0,
                               const double* p0, const double* v0, const double* a0) {
  long k = -1;
  guarded("target_manager_init_batch", [&] { k = M(self)->initBatch(ids, n, dt0, t0, p0, v0, a0); });
  return k;
}

long target_manager_init_batch_typed(target_manager_c* self, int type, const unsigned int* ids, long n, double dt0,
                                     double t0, const double* Q, const double* R, const double* P0, int per_target_P0,
                                     const double* p0, const double* v0, const double* a0) {
  long k = -1;
  guarded("target_manager_init_batch_typed", [&] {
    k = M(self)->initBatch((TargetManager::target_t)type, ids, n, dt0, t0, Q, R, P0, per_target_P0 != 0, p0, v0, a0);
  });
  return k;
}

long target_manager_init_batch_classes(target_manager_c* self, int type, const unsigned int* ids, long n, double dt0, double t0,
                                       long n_classes, const double* Q, const double* R, const double* P0,
                                       const unsigned int* class_of, const double* p0, const double* v0, const double* a0) {
  long k = -1;
  guarded("target_manager_init_batch_classes", [&] {
    if (!ids || !Q || !R || !P0 || !class_of || !p0) throw std::invalid_argument("NULL argument");
    k = M(self)->initBatchClasses((TargetManager::target_t)type, ids, n, dt0, t0, n_classes, Q, R, P0, class_of, p0, v0, a0);
  });
  return k;
}

int target_batch_num_classes(target_batch_c* b) {
  return guarded_value<int>("target_batch_num_classes", -1, [&]() -> int { return B(b)->n_classes(); });
}

int target_manager_erase(target_manager_c* self, unsigned int id) {
  int r = -1;
  guarded("target_manager_erase", [&] { r = M(self)->erase(id) ? 1 : 0; });
  return r;
}

long target_manager_erase_batch(target_manager_c* self, const unsigned int* ids, long n) {
  long k = -1;
  guarded("target_manager_erase_batch", [&] { k = M(self)->eraseBatch(ids, n); });
  return k;
}

long target_manager_size(target_manager_c* self) {
  long n = -1;
  guarded("target_manager_size", [&] { n = (long)M(self)->size(); });
  return n;
}

long target_manager_get_available_targets(target_manager_c* self, unsigned int* ids_out, long capacity) {
  long n = -1;
  guarded("target_manager_get_available_targets", [&] {
    auto ids = M(self)->getAvailableTargets();
    n = (long)ids.size();
    for (long i = 0; i < n && i < capacity; ++i) ids_out[i] = ids[(size_t)i];
  });
  return n;
}

long target_manager_update_meas_batch(target_manager_c* self, const unsigned int* ids, long n, double dt,
                                      const double* meas, const unsigned char* has_meas) {
  long k = -1;
  guarded("target_manager_update_meas_batch", [&] { k = M(self)->updateBatch(ids, n, dt, meas, has_meas); });
  return k;
}

int target_manager_update_all(target_manager_c* self, double dt) {
  return guarded("target_manager_update_all", [&] { M(self)->update(dt); });
}

long target_manager_get_est_batch(target_manager_c* self, const unsigned int* ids, long n, double* pose, double* twist,
                                  double* acceleration, unsigned char* found) {
  long k = -1;
  guarded("target_manager_get_est_batch", [&] { k = M(self)->getPoseBatch(ids, n, pose, twist, acceleration, found); });
  return k;
}

long target_manager_get_est_at_batch(target_manager_c* self, const unsigned int* ids, long n, double t1, double* pose,
                                     double* twist, double* acceleration, unsigned char* found) {
  long k = -1;
  guarded("target_manager_get_est_at_batch", [&] {
    k = M(self)->getPoseBatch(ids, n, pose, twist, acceleration, found, true, t1);
  });
  return k;
}

long target_manager_get_state_batch(target_manager_c* self, const unsigned int* ids, long n, double* x, double* P) {
  long k = -1;
  guarded("target_manager_get_state_batch", [&] { k = M(self)->getStateBatch(ids, n, x, P); });
  return k;
}

int target_manager_get_time(target_manager_c* self, unsigned int id, double* t) {
  int r = -1;
  guarded("target_manager_get_time", [&] { r = M(self)->getTargetTime(id, *t) ? 0 : -2; });
  return r;
}

double target_manager_get_intersection_time_with_sphere(target_manager_c* self, unsigned int id, double t1,
                                                         const double* origin, double radius) {
  double d = -1;
  guarded("target_manager_get_intersection_time_with_sphere", [&] {
    d = M(self)->getIntersectionTimeWithSphere(id, t1, origin, radius);
  });
  return d;
}

bool target_manager_get_intersection_pose_with_sphere(target_manager_c* self, unsigned int id, double t1,
                                                      const double* origin, double radius, double* pose,
                                                      double* delta) {
  bool r = false;
  guarded("target_manager_get_intersection_pose_with_sphere", [&] {
    r = M(self)->getIntersectionPoseWithSphere(id, t1, origin, radius, pose, delta);
  });
  return r;
}

long target_manager_intersect_sphere_batch(target_manager_c* self, const unsigned int* ids, long n, double t1,
                                           const double* origin, double radius, double* delta, double* pose,
                                           unsigned char* found) {
  long k = -1;
  guarded("target_manager_intersect_sphere_batch", [&] {
    k = M(self)->intersectBatch(ids, n, t1, origin, radius, delta, pose, found);
  });
  return k;
}

long target_manager_intersect_sphere_converged_batch(target_manager_c* self, const unsigned int* ids, long n, double t1,
                                                     double pos_th, double ang_th, const double* origin, double radius,
                                                     int filters_length, double* delta, double* pose,
                                                     unsigned char* converged, unsigned char* found, double* filtered_errors) {
  long k = -1;
  guarded("target_manager_intersect_sphere_converged_batch", [&] {
    M(self)->setIntersectionFiltersLength(filters_length > 0 ? filters_length : 250);
    k = M(self)->intersectGatedBatch(ids, n, t1, pos_th, ang_th, origin, radius, delta, pose, converged, found, filtered_errors);
  });
  return k;
}

int target_batch_intersect_sphere_converged_dev(target_batch_c* b, double t1, double pos_th, double ang_th,
                                                const double* origin, double radius, int filters_length,
                                                double* delta_dev, double* pose_dev, unsigned char* converged_dev) {
  return guarded("target_batch_intersect_sphere_converged_dev", [&] { BatchLock lk(B(b));
    B(b)->intersect_gated_dev(t1, origin, radius, pos_th, ang_th, filters_length > 0 ? filters_length : 250, delta_dev,
                              pose_dev, converged_dev);
  });
}

int target_batch_gate_update_dev(target_batch_c* b, const double* delta_dev, const double* pose_dev, double pos_th, double ang_th,
                                 int filters_length, unsigned char* converged_dev, double* filtered_dev, double* variance_dev) {
  return guarded("target_batch_gate_update_dev", [&] { BatchLock lk(B(b));
    if (!delta_dev || !pose_dev || !converged_dev) throw std::invalid_argument("delta, pose and converged are required");
    B(b)->gate_update_dev(delta_dev, pose_dev, pos_th, ang_th, filters_length > 0 ? filters_length : 250, converged_dev, filtered_dev,
                          variance_dev);
  });
}

int target_batch_intersect_sphere_dev(target_batch_c* b, double t1, const double* origin, double radius,
                                      double* delta_dev, double* pose_dev) {
  return guarded("target_batch_intersect_sphere_dev", [&] { BatchLock lk(B(b)); B(b)->intersect_dev(t1, origin, radius, delta_dev, pose_dev); });
}

int target_manager_num_batches(target_manager_c* self) {
  return guarded_value<int>("target_manager_num_batches", -1, [&]() -> int { return M(self)->numBatches(); });
}

target_batch_c* target_manager_get_batch(target_manager_c* self, int index) {
  return guarded_value<target_batch_c*>("target_manager_get_batch", nullptr, [&]() -> target_batch_c* {
    if (index < 0 || index >= M(self)->numBatches()) return nullptr;
    return (target_batch_c*)M(self)->batch(index);
  });
}

target_batch_c* target_manager_get_batch_of_type(target_manager_c* self, int type) {
  return guarded_value<target_batch_c*>("target_manager_get_batch_of_type", nullptr,
                                        [&]() -> target_batch_c* { return (target_batch_c*)M(self)->batchOfType(type); });
}

long target_batch_size(target_batch_c* b) {
  return guarded_value<long>("target_batch_size", -1, [&]() -> long { return B(b)->size(); });
}
int target_batch_type(target_batch_c* b) {
  return guarded_value<int>("target_batch_type", -1, [&]() -> int { return B(b)->type(); });
}
int target_batch_dtype(target_batch_c* b) {
  return guarded_value<int>("target_batch_dtype", -1, [&]() -> int { return B(b)->dtype(); });
}
int target_batch_state_dim(target_batch_c* b) {
  return guarded_value<int>("target_batch_state_dim", -1, [&]() -> int { return B(b)->n_state(); });
}
int target_batch_meas_dim(target_batch_c* b) {
  return guarded_value<int>("target_batch_meas_dim", -1, [&]() -> int { return B(b)->n_meas(); });
}
int target_batch_lanes_per_target(target_batch_c* b) {
  return guarded_value<int>("target_batch_lanes_per_target", -1, [&]() -> int { return B(b)->layout().g; });
}
int target_batch_is_symmetric_packed(target_batch_c* b) {
  return guarded_value<int>("target_batch_is_symmetric_packed", -1, [&]() -> int { return B(b)->layout().layout == te::LAYOUT_PACKED; });
}
int target_batch_layout(target_batch_c* b) {
  return guarded_value<int>("target_batch_layout", -1, [&]() -> int { return B(b)->layout().layout; });
}
long target_batch_algorithmic_bytes(target_batch_c* b) {
  return guarded_value<long>("target_batch_algorithmic_bytes", -1, [&]() -> long { return B(b)->algorithmic_bytes_per_cycle(); });
}
double target_batch_resident_bytes_per_target(target_batch_c* b) {
  return guarded_value<double>("target_batch_resident_bytes_per_target", -1.0,
                               [&]() -> double { return (double)B(b)->layout().tile_bytes / (double)B(b)->layout().tpw; });
}

long target_batch_slot_ids(target_batch_c* b, unsigned int* ids_out, long capacity) {
  return guarded_value<long>("target_batch_slot_ids", -1, [&]() -> long {
    BatchLock lk(B(b));
    const auto& ids = B(b)->slot_ids();
    const long n = (long)ids.size();
    for (long i = 0; i < n && i < capacity; ++i) ids_out[i] = ids[(size_t)i];
    return n;
  });
}

int target_batch_step(target_batch_c* b, double dt, const void* meas_dev, long ld, const unsigned char* has_meas_dev) {
  return guarded("target_batch_step", [&] { BatchLock lk(B(b)); B(b)->step_dense(dt, meas_dev, ld, has_meas_dev); });
}

int target_batch_step_host(target_batch_c* b, double dt, const void* meas_soa_host, long ld_host, const unsigned char* has_meas_host) {
  return guarded("target_batch_step_host", [&] { BatchLock lk(B(b)); B(b)->step_dense_host_soa(dt, meas_soa_host, ld_host, has_meas_host); });
}

int target_batch_step_sequence(target_batch_c* b, long n_ticks, double dt, const void* meas_dev, long tick_stride, long ld,
                               const unsigned char* has_meas_dev, long has_stride, int use_graph) {
  return guarded("target_batch_step_sequence", [&] { BatchLock lk(B(b));
    B(b)->step_sequence(n_ticks, dt, meas_dev, tick_stride, ld, has_meas_dev, has_stride, use_graph);
  });
}

int target_batch_step_sequence_ring(target_batch_c* b, long n_ticks, double dt, const void* meas_dev, long tick_stride, long ld,
                                    const unsigned char* has_meas_dev, long has_stride, long ring_ticks, int use_graph) {
  return guarded("target_batch_step_sequence_ring", [&] { BatchLock lk(B(b));
    if (ring_ticks <= 0) throw std::invalid_argument("ring_ticks must be positive");
    B(b)->step_sequence(n_ticks, dt, meas_dev, tick_stride, ld, has_meas_dev, has_stride, use_graph, ring_ticks);
  });
}

int target_manager_step_sequence_all(target_manager_c* m, long n_ticks, double dt, const target_batch_sequence_c* per_batch,
                                     long n_batches, int query, const double* origin, double radius, int use_graph) {
  return guarded("target_manager_step_sequence_all", [&] {
    std::vector<te::Batch::SeqSpec> specs((size_t)(n_batches > 0 ? n_batches : 0));
    for (long i = 0; i < n_batches; ++i) {
      const target_batch_sequence_c& s = per_batch[i];
      specs[(size_t)i] = te::Batch::SeqSpec{s.meas_dev, s.tick_stride, s.ld, s.has_meas_dev, s.has_stride, s.delta_dev, s.pose_dev, s.ring_ticks};
    }
    M(m)->stepSequenceAll(n_ticks, dt, specs.data(), n_batches, query != 0, origin, radius, use_graph);
  });
}

int target_manager_population_tick(target_manager_c* m) {
  int on = 0;
  const int rc = guarded("target_manager_population_tick", [&] { on = M(m)->populationTickNow() ? 1 : 0; });
  return rc < 0 ? -1 : on;
}

int target_batch_step_fused(target_batch_c* b, long n_ticks, double dt, const void* meas_dev, long tick_stride, long ld,
                            const unsigned char* has_meas_dev, long has_stride) {
  return guarded("target_batch_step_fused", [&] { BatchLock lk(B(b));
    B(b)->step_fused(n_ticks, dt, meas_dev, tick_stride, ld, has_meas_dev, has_stride);
  });
}

// ---- resident ("live") mode
int target_batch_live_start(target_batch_c* b, double dt, const void* meas_ring_dev, long tick_stride, long ld,
                            const unsigned char* has_ring_dev, long has_stride, long ring_ticks, long first_entry, long max_ticks,
                            double idle_limit_s) {
  return guarded("target_batch_live_start", [&] { BatchLock lk(B(b));
    B(b)->live_start(dt, meas_ring_dev, tick_stride, ld, has_ring_dev, has_stride, ring_ticks, first_entry, max_ticks, idle_limit_s);
  });
}
int target_batch_live_set_pose_output(target_batch_c* b, double* pose_soa_dev, long ld) {
  return guarded("target_batch_live_set_pose_output", [&] { BatchLock lk(B(b)); B(b)->live_set_pose_output(pose_soa_dev, ld); });
}
int target_batch_live_post(target_batch_c* b, long n_ticks) {
  return guarded("target_batch_live_post", [&] { BatchLock lk(B(b)); B(b)->live_post(n_ticks); });
}
int target_batch_live_post_each(target_batch_c* b, long n_ticks) {
  return guarded("target_batch_live_post_each", [&] { BatchLock lk(B(b)); for (long i = 0; i < n_ticks; ++i) B(b)->live_post(1); });
}
long target_batch_live_done(target_batch_c* b) {
  return guarded_value<long>("target_batch_live_done", -1L, [&] { return B(b)->live_done(); });   // reads host-mapped words: no lock
}
int target_batch_live_wait(target_batch_c* b, long tick, double timeout_s) {
  int late = 0;
  const int rc = guarded("target_batch_live_wait", [&] { late = B(b)->live_wait(tick, timeout_s) ? 0 : 1; });
  return rc != 0 ? rc : late;
}
long target_batch_live_stop(target_batch_c* b) {
  return guarded_value<long>("target_batch_live_stop", -1L, [&] { BatchLock lk(B(b)); return B(b)->live_stop(); });
}
int target_batch_live_running(target_batch_c* b) {
  return guarded_value<int>("target_batch_live_running", -1, [&] { return B(b)->live_running() ? 1 : 0; });   // reads a host-mapped word: no lock
}
long target_batch_live_capacity(target_batch_c* b) {
  return guarded_value<long>("target_batch_live_capacity", -1L, [&] { return B(b)->live_capacity_targets(B(b)->live_pose_output_set()); });
}

int target_manager_live_start_all(target_manager_c* m, double dt, const target_batch_sequence_c* per_batch, long n_batches, long first_entry,
                                  long max_ticks, double idle_limit_s, int query, const double* origin, double radius) {
  return guarded("target_manager_live_start_all", [&] {
    if (!per_batch || n_batches <= 0) throw std::invalid_argument("NULL per-batch description");
    std::vector<Batch::SeqSpec> specs((size_t)n_batches);
    for (long i = 0; i < n_batches; ++i)
      specs[(size_t)i] = Batch::SeqSpec{per_batch[i].meas_dev, per_batch[i].tick_stride, per_batch[i].ld, per_batch[i].has_meas_dev,
                                        per_batch[i].has_stride, per_batch[i].delta_dev, per_batch[i].pose_dev, per_batch[i].ring_ticks};
    M(m)->liveStartAll(dt, specs.data(), n_batches, first_entry, max_ticks, idle_limit_s, query != 0, origin, radius);
  });
}
int target_manager_live_post_all(target_manager_c* m, long n_ticks, int one_doorbell_per_tick) {
  return guarded("target_manager_live_post_all", [&] { M(m)->livePostAll(n_ticks, one_doorbell_per_tick != 0); });
}
long target_manager_live_done_all(target_manager_c* m) {
  return guarded_value<long>("target_manager_live_done_all", -1L, [&] { return M(m)->liveDoneAll(); });
}
int target_manager_live_wait_all(target_manager_c* m, long tick, double timeout_s) {
  int late = 0;
  const int rc = guarded("target_manager_live_wait_all", [&] { late = M(m)->liveWaitAll(tick, timeout_s) ? 0 : 1; });
  return rc != 0 ? rc : late;
}
long target_manager_live_stop_all(target_manager_c* m) {
  return guarded_value<long>("target_manager_live_stop_all", -1L, [&] { return M(m)->liveStopAll(); });
}

int target_batch_get_est_dev(target_batch_c* b, double* pose_dev, double* twist_dev, double* acc_dev, int at_time, double t1) {
  return guarded("target_batch_get_est_dev", [&] { BatchLock lk(B(b)); B(b)->outputs_dev(pose_dev, twist_dev, acc_dev, at_time != 0, t1); });
}

int target_batch_pack_meas_dev(target_batch_c* b, const double* meas_aos_dev, long n, void* meas_soa_dev, long ld) {
  return guarded("target_batch_pack_meas_dev", [&] { BatchLock lk(B(b)); B(b)->pack_meas_dev(meas_aos_dev, n, meas_soa_dev, ld); });
}

// ---------------------------------------------------------------- pose gather over xGMI (RCCL)
int target_comm_unique_id(char* out128) {
  return guarded("target_comm_unique_id", [&] {
    if (!out128) throw std::invalid_argument("NULL buffer");
    te::PoseComm::unique_id(out128);
  });
}

target_comm_c* target_comm_new(const char* id128, int rank, int world) {
  te::PoseComm* c = nullptr;
  guarded("target_comm_new", [&] {
    if (!id128) throw std::invalid_argument("NULL id");
    c = new te::PoseComm(id128, rank, world);
  });
  return (target_comm_c*)c;
}

void target_comm_delete(target_comm_c* comm) {
  if (comm) guarded("target_comm_delete", [&] { delete (te::PoseComm*)comm; });
}

int target_manager_gather_pose_begin(target_manager_c* self, target_comm_c* comm, int root, const long* counts, double* recv_dev) {
  return guarded("target_manager_gather_pose_begin", [&] {
    if (!comm || !counts) throw std::invalid_argument("NULL communicator or counts");
    ((te::PoseComm*)comm)->begin(M(self), root, counts, recv_dev);
  });
}

int target_manager_gather_pose_wait(target_comm_c* comm, float* device_ms) {
  return guarded("target_manager_gather_pose_wait", [&] {
    if (!comm) throw std::invalid_argument("NULL communicator");
    ((te::PoseComm*)comm)->wait();
    if (device_ms) *device_ms = ((te::PoseComm*)comm)->last_ms();
  });
}

int target_manager_gather_pose_wait_for(target_comm_c* comm, double timeout_s, float* device_ms) {
  int pending = 0;
  const int rc = guarded("target_manager_gather_pose_wait_for", [&] {
    if (!comm) throw std::invalid_argument("NULL communicator");
    if (!((te::PoseComm*)comm)->wait_for(timeout_s)) { pending = 1; return; }
    if (device_ms) *device_ms = ((te::PoseComm*)comm)->last_ms();
  });
  return rc != 0 ? rc : pending;
}

// ---------------------------------------------------------------- measurement ingest
target_ingest_c* target_ingest_new(target_manager_c* manager, int type, const double* Q, const double* R, const double* P0) {
  te::MeasurementIngest* ing = nullptr;
  guarded("target_ingest_new", [&] { ing = new te::MeasurementIngest(M(manager), type, Q, R, P0); });
  return (target_ingest_c*)ing;
}

void target_ingest_delete(target_ingest_c* ingest) {
  if (ingest) guarded("target_ingest_delete", [&] { delete (te::MeasurementIngest*)ingest; });
}

void target_ingest_set_expiration_time(target_ingest_c* ingest, double seconds) {
  guarded("target_ingest_set_expiration_time", [&] { I(ingest)->setExpirationTime(seconds); });
}

void target_ingest_set_token_name(target_ingest_c* ingest, const char* token) {
  guarded("target_ingest_set_token_name", [&] { I(ingest)->setTargetTokenName(token); });
}

int target_ingest_push(target_ingest_c* ingest, unsigned int id, double stamp, const double* pose) {
  return guarded("target_ingest_push", [&] { I(ingest)->push(id, stamp, pose); });
}

int target_ingest_push_named(target_ingest_c* ingest, const char* child_frame_id, double stamp, const double* pose) {
  int r = -2;
  guarded("target_ingest_push_named", [&] { r = I(ingest)->push_named(child_frame_id, stamp, pose); });
  return r;
}

long target_ingest_tick(target_ingest_c* ingest, double dt, double now, unsigned int* ids_out, double* poses_out, long capacity) {
  long n = -1;
  guarded("target_ingest_tick", [&] {
    std::vector<unsigned> ids;
    std::vector<double> poses;
    n = I(ingest)->tick(dt, now, ids, poses);
    for (long i = 0; i < n && i < capacity; ++i) {
      if (ids_out) ids_out[i] = ids[(size_t)i];
      if (poses_out) for (int c = 0; c < 7; ++c) poses_out[i * 7 + c] = poses[(size_t)i * 7 + c];
    }
  });
  return n;
}

// ---------------------------------------------------------------- IntersectionSolver as an object
target_intersection_solver_c* target_intersection_solver_new(target_manager_c* manager, unsigned int filters_length) {
  te::IntersectionSolver* sv = nullptr;
  guarded("target_intersection_solver_new", [&] { sv = new te::IntersectionSolver(M(manager), filters_length); });
  return (target_intersection_solver_c*)sv;
}
void target_intersection_solver_delete(target_intersection_solver_c* solver) {
  if (solver) guarded("target_intersection_solver_delete", [&] { delete (te::IntersectionSolver*)solver; });
}
double target_intersection_solver_get_time_with_sphere(target_intersection_solver_c* solver, unsigned int id, double t1,
                                                        const double* origin, double radius) {
  return guarded_value<double>("target_intersection_solver_get_time_with_sphere", -1.0, [&] {
    if (!solver || !origin) throw std::invalid_argument("NULL argument");
    return ((te::IntersectionSolver*)solver)->getIntersectionTimeWithSphere(id, t1, origin, radius);
  });
}
bool target_intersection_solver_get_pose_with_sphere(target_intersection_solver_c* solver, unsigned int id, double t1, double pos_th,
                                                     double ang_th, const double* origin, double radius, double* pose7) {
  return guarded_value<bool>("target_intersection_solver_get_pose_with_sphere", false, [&] {
    if (!solver || !origin || !pose7) throw std::invalid_argument("NULL argument");
    return ((te::IntersectionSolver*)solver)->getIntersectionPoseWithSphere(id, t1, pos_th, ang_th, origin, radius, pose7);
  });
}
void target_intersection_solver_last_errors(target_intersection_solver_c* solver, double* pos_error_filtered, double* ang_error_filtered) {
  if (!solver) return;
  if (pos_error_filtered) *pos_error_filtered = ((te::IntersectionSolver*)solver)->lastPositionErrorFiltered();
  if (ang_error_filtered) *ang_error_filtered = ((te::IntersectionSolver*)solver)->lastAngleErrorFiltered();
}

// ---------------------------------------------------------------- TargetInterface / estimator getters
int target_manager_set_keep_measurement(target_manager_c* self, int on) {
  return guarded("target_manager_set_keep_measurement", [&] { M(self)->setKeepMeasurement(on != 0); });
}
bool target_manager_get_measured_pose(target_manager_c* self, unsigned int id, double* pose7) {
  return guarded_value<bool>("target_manager_get_measured_pose", false, [&] { return M(self)->getTargetMeasuredPose(id, pose7); });
}
bool target_manager_get_period_estimate(target_manager_c* self, unsigned int id, double* period) {
  return guarded_value<bool>("target_manager_get_period_estimate", false, [&] {
    double p = -1.0;
    const bool ok = M(self)->getTargetPeriodEstimate(id, p);
    if (ok && period) *period = p;
    return ok;
  });
}
bool target_manager_get_estimated_transform(target_manager_c* self, unsigned int id, double* T16) {
  return guarded_value<bool>("target_manager_get_estimated_transform", false, [&] { return M(self)->getTargetTransform(id, T16); });
}
int target_manager_get_n(target_manager_c* self, unsigned int id) {
  return guarded_value<int>("target_manager_get_n", -1, [&] { int n = 0, m = 0; return M(self)->getTargetDims(id, n, m) ? n : 0; });
}
int target_manager_get_m(target_manager_c* self, unsigned int id) {
  return guarded_value<int>("target_manager_get_m", -1, [&] { int n = 0, m = 0; return M(self)->getTargetDims(id, n, m) ? m : 0; });
}
bool target_manager_get_model_matrices(target_manager_c* self, unsigned int id, double* Q, double* R, double* P0) {
  return guarded_value<bool>("target_manager_get_model_matrices", false, [&] { return M(self)->getTargetModelMatrices(id, Q, R, P0); });
}
int target_manager_set_log_targets(target_manager_c* self, const unsigned int* ids, long n) {
  return guarded("target_manager_set_log_targets", [&] {
    if (n > 0 && !ids) throw std::invalid_argument("NULL id list");
    M(self)->setLogTargets(ids, n);
  });
}

// ---------------------------------------------------------------- synthetic measurement streams
int target_stream_fill_dev(const target_stream_c* sp, long n_targets, long first_tick, long n_ticks, int dtype, void* meas_dev,
                           long tick_stride, long ld, unsigned char* has_meas_dev, long has_stride, void* hip_stream) {
  return guarded("target_stream_fill_dev", [&] {
    if (!sp) throw std::invalid_argument("NULL stream description");
    if (dtype != TARGET_DTYPE_F64 && dtype != TARGET_DTYPE_F32) throw std::invalid_argument("unknown dtype");
    te::stream_fill(te::StreamSpec{sp->model, (uint64_t)sp->seed, sp->first_target, sp->dt, sp->availability, sp->rpy_noise}, n_targets,
                    first_tick, n_ticks, dtype == TARGET_DTYPE_F32, meas_dev, tick_stride, ld, has_meas_dev, has_stride, (hipStream_t)hip_stream);
  });
}

int target_stream_truth_dev(const target_stream_c* sp, long n_targets, double* pose0_dev, double* truth_dev, void* hip_stream) {
  return guarded("target_stream_truth_dev", [&] {
    if (!sp) throw std::invalid_argument("NULL stream description");
    te::stream_truth(te::StreamSpec{sp->model, (uint64_t)sp->seed, sp->first_target, sp->dt, sp->availability, sp->rpy_noise}, n_targets,
                     pose0_dev, truth_dev, (hipStream_t)hip_stream);
  });
}

}  // extern "C"
