// target_manager.cpp -- see target_manager.hpp.
#include "target_manager.hpp"
#include "kf_population.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "hip_check.hpp"
#include "yaml_mini.hpp"

namespace te {

using std::lock_guard;
using std::mutex;

static const double kZero6[6] = {0, 0, 0, 0, 0, 0};

TargetManager::TargetManager(int dtype, int lanes_per_target) : dtype_(dtype), lanes_(lanes_per_target) {
  const char* v = std::getenv("TARGET_ESTIMATION_VERBOSE");
  verbose_ = v && v[0] && v[0] != '0';
  const char* ld = std::getenv("TARGET_ESTIMATION_LOG_DIR");
  if (ld && ld[0]) { log_dir_ = ld; keep_meas_ = true; }   // batches created later inherit the measured-pose rows
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    throw std::runtime_error("target_estimation_amd: no HIP device available; this library has no CPU path");
}

TargetManager::TargetManager(const std::string& file, int dtype, int lanes_per_target)
    : TargetManager(dtype, lanes_per_target) {
  if (!loadYamlFile(file, default_Q_, default_R_, default_P_, default_type_))
    throw "TargetManager default constructor failed!";
  else
    default_values_loaded_ = true;
}

void TargetManager::devIdsFree() {
  DevIds& d = dev_ids_;
  device_free(d.keys); device_free(d.vals); device_free(d.seen);
  device_free(d.ids); device_free(d.loc); device_free(d.idx); device_free(d.aos); device_free(d.soa);
  device_free(d.mask); device_free(d.found); device_free(d.out); device_free(d.counters);
  if (d.h_counters) (void)hipHostFree(d.h_counters);
  d = DevIds();
}

void TargetManager::devIdsReserve(long n) {
  DevIds& d = dev_ids_;
  if (!d.counters) {
    TE_HIP_CHECK(hipMalloc((void**)&d.counters, sizeof(ResolveCounters)));
    TE_HIP_CHECK(hipHostMalloc((void**)&d.h_counters, sizeof(ResolveCounters), hipHostMallocDefault));
  }
  if (n <= d.cap) return;
  const long want = std::max(n, d.cap * 2);
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  device_free(d.ids); device_free(d.loc); device_free(d.idx); device_free(d.aos); device_free(d.soa);
  device_free(d.mask); device_free(d.found); device_free(d.out);
  TE_HIP_CHECK(hipMalloc((void**)&d.ids, sizeof(unsigned) * want));
  TE_HIP_CHECK(hipMalloc((void**)&d.loc, sizeof(int) * want));
  TE_HIP_CHECK(hipMalloc((void**)&d.idx, sizeof(int) * want));
  TE_HIP_CHECK(hipMalloc((void**)&d.aos, sizeof(double) * 7 * want));
  TE_HIP_CHECK(hipMalloc((void**)&d.soa, (dtype_ == F64 ? 8 : 4) * 7 * (size_t)want));
  TE_HIP_CHECK(hipMalloc((void**)&d.mask, (size_t)want));
  TE_HIP_CHECK(hipMalloc((void**)&d.found, (size_t)want));
  TE_HIP_CHECK(hipMalloc((void**)&d.out, sizeof(double) * 19 * want));
  d.cap = want;
}

void TargetManager::devIdsRebuild() {
  DevIds& d = dev_ids_;
  const size_t total = targets_.size();
  int log2cap = 4;
  while ((size_t(1) << log2cap) < 2 * total + 16) ++log2cap;
  if (log2cap > 31) throw std::runtime_error("target_estimation_amd: too many targets for the device id table");
  if (log2cap != d.log2cap) {
    TE_HIP_CHECK(hipStreamSynchronize(stream_));
    device_free(d.keys); device_free(d.vals); device_free(d.seen);
    const size_t cap = size_t(1) << log2cap;
    TE_HIP_CHECK(hipMalloc((void**)&d.keys, sizeof(unsigned) * cap));
    TE_HIP_CHECK(hipMalloc((void**)&d.vals, sizeof(unsigned) * cap));
    TE_HIP_CHECK(hipMalloc((void**)&d.seen, sizeof(int) * cap));
    d.log2cap = log2cap;
  }
  const size_t cap = size_t(1) << d.log2cap;
  TE_HIP_CHECK(hipMemsetAsync(d.vals, 0xFF, sizeof(unsigned) * cap, stream_));
  TE_HIP_CHECK(hipMemsetAsync(d.seen, 0, sizeof(int) * cap, stream_));
  d.epoch = 0;
  for (size_t b = 0; b < batches_.size(); ++b) {
    const long n = batches_[b]->size();
    if (!n) continue;
    devIdsReserve(n);
    TE_HIP_CHECK(hipMemcpyAsync(d.ids, batches_[b]->slot_ids().data(), sizeof(unsigned) * n, hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(id_table_insert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream_, d.keys, d.vals, d.log2cap,
                       d.ids, n, (unsigned)b);
    TE_HIP_CHECK(hipGetLastError());
    TE_HIP_CHECK(hipStreamSynchronize(stream_));   // d.ids is reused by the next batch; slot_ids() is pageable memory
  }
  d.dirty = false;
}

bool TargetManager::resolveOnDevice(const unsigned* ids, long n, ResolveCounters& out) {
  if (batches_.empty() || batches_.size() > (size_t)kIdMaxBatches) return false;
  for (auto& b : batches_)
    if (b->size() >= (1L << kIdSlotBits)) return false;
  DevIds& d = dev_ids_;
  if (d.dirty) devIdsRebuild();
  devIdsReserve(n);
  if (++d.epoch == 0x7fffffff) { TE_HIP_CHECK(hipMemsetAsync(d.seen, 0, sizeof(int) * (size_t(1) << d.log2cap), stream_)); d.epoch = 1; }
  TE_HIP_CHECK(hipMemcpyAsync(d.ids, ids, sizeof(unsigned) * n, hipMemcpyHostToDevice, stream_));
  TE_HIP_CHECK(hipMemsetAsync(d.counters, 0, sizeof(ResolveCounters), stream_));
  hipLaunchKernelGGL(id_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream_, d.keys, d.vals, d.seen, d.log2cap,
                     d.ids, n, d.epoch, d.loc, d.counters);
  TE_HIP_CHECK(hipGetLastError());
  TE_HIP_CHECK(hipMemcpyAsync(d.h_counters, d.counters, sizeof(ResolveCounters), hipMemcpyDeviceToHost, stream_));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  out = *d.h_counters;
  return true;
}

TargetManager::~TargetManager() {
  closeLogFiles();
  devIdsFree();
  dropSeqGraphs();
  for (auto st : branch_streams_) (void)hipStreamDestroy(st);
  for (auto ev : branch_events_) (void)hipEventDestroy(ev);
}

bool TargetManager::selectTargetType(const std::string& type_str, target_t& type) {
  if (type_str == "angular_rates") type = ANGULAR_RATES;
  else if (type_str == "angular_velocities") type = ANGULAR_VELOCITIES;
  else if (type_str == "uniform_acceleration") type = UNIFORM_ACCELERATION;
  else if (type_str == "uniform_velocity") type = UNIFORM_VELOCITY;
  else return false;
  return true;
}

bool TargetManager::loadYamlFile(const std::string& file, std::vector<double>& Q, std::vector<double>& R,
                                 std::vector<double>& P, target_t& type) {
  bool success = true;
  ModelFile mf;
  std::string err;
  if (!load_model_file(file, mf, err)) {
    std::cerr << err << std::endl;
    return false;
  }
  if (!selectTargetType(mf.type, type)) {
    std::cerr << "Can not parse type: " << mf.type << std::endl;
    std::cerr << "Can not load type from file: " << file << std::endl;
    return false;
  }
  const size_t n = (size_t)model_n((int)type), m = (size_t)model_m((int)type);
  const char* names[3] = {"Q", "R", "P"};
  std::vector<double>* dst[3] = {&Q, &R, &P};
  const size_t want[3] = {n * n, m * m, n * n};
  for (int k = 0; k < 3; ++k) {
    auto it = mf.seqs.find(names[k]);
    if (it == mf.seqs.end() || it->second.size() != want[k]) {
      // the reference maps any square list; its models then assert n (uniform_velocity.cpp:34 ...)
      std::cerr << "Can not load matrix " << names[k] << " from file: " << file << std::endl;
      success = false;
      continue;
    }
    // The reference maps the row-major YAML list column-major (target_manager.cpp:25), i.e. it
    // reads the transpose.  Reproduce that exactly (immaterial for the symmetric shipped models).
    const size_t s = (size_t)std::llround(std::sqrt((double)want[k]));
    dst[k]->assign(want[k], 0.0);
    for (size_t r = 0; r < s; ++r)
      for (size_t c = 0; c < s; ++c) (*dst[k])[r * s + c] = it->second[c * s + r];
  }
  return success;
}

bool is_axis_separable(int type, const double* Q, const double* R, const double* P0, long n_P0) {
  const int n = model_n(type), m = model_m(type);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) {
      if (group_of(type, r) == group_of(type, c)) continue;
      if (Q[r * n + c] != 0.0) return false;
      for (long k = 0; k < n_P0; ++k)
        if (P0[k * n * n + r * n + c] != 0.0) return false;
    }
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < m; ++c)
      if (group_of(type, r) != group_of(type, c) && R[r * m + c] != 0.0) return false;
  return true;
}

// exact symmetry of Q, R and every P0 (covariances are; the reference accepts any matrix)
static bool all_symmetric(int type, const double* Q, const double* R, const double* P0, long n_P0) {
  const int n = model_n(type), m = model_m(type);
  for (int r = 0; r < n; ++r)
    for (int c = r + 1; c < n; ++c) {
      if (Q[r * n + c] != Q[c * n + r]) return false;
      for (long k = 0; k < n_P0; ++k)
        if (P0[k * n * n + r * n + c] != P0[k * n * n + c * n + r]) return false;
    }
  for (int r = 0; r < m; ++r)
    for (int c = r + 1; c < m; ++c)
      if (R[r * m + c] != R[c * m + r]) return false;
  return true;
}

int TargetManager::chooseLayout(int type, const double* Q, const double* R, const double* P0, long n_P0) const {
  constexpr int kSeparable = 201;        // 1 + TARGET_LAYOUT_AXIS_SEPARABLE
  constexpr int kSeparablePacked = 301;  // 1 + TARGET_LAYOUT_AXIS_SEPARABLE_PACKED
  const bool sep = is_axis_separable(type, Q, R, P0, n_P0);
  // automatic: the smallest record the matrices allow -- per-axis-group blocks when nothing couples the
  // groups, their upper triangles only when everything is symmetric as well
  if (lanes_ == 0) {
    const bool sym = all_symmetric(type, Q, R, P0, n_P0);
    if (sep) return sym ? kSeparablePacked : kSeparable;
    if (!sym) return 0;   // general matrices: dense kernel, full P, tuned lanes per target
    // coupled but symmetric: dense kernel on the upper triangle (100 + lanes per target): per (model, precision) the
    // fastest packed form at 10^6 targets (profiles/r02_layout_sweep.txt, profiles/r02_kernel_resources.txt).
    //   angular_velocities: 101 = thread per target on the triangle in place (ekf_sym.hpp): 309 us fp64 / 190 us fp32 per
    //     10^6-target tick against 493 / 212 us for the best lanes-per-target form (106 / 103);
    //   angular_rates: fp64 106 held to two wavefronts per SIMD (kf_step.hpp step_min_waves: 647 us; 103 takes 714 us at
    //     one wavefront), fp32 103 (299 us).
    // The AV picks run at one wavefront per SIMD: they are the fastest forms measured, their two-wave alternatives lose 10-60 %.
    switch (type) {
      case ANGULAR_RATES: return dtype_ == F32 ? 103 : 106;
      case ANGULAR_VELOCITIES: return 101;
      case UNIFORM_ACCELERATION: return dtype_ == F32 ? 103 : 101;
      default: return 101;
    }
  }
  if ((lanes_ == kSeparable || lanes_ == kSeparablePacked) && !sep)
    throw std::runtime_error("target_estimation_amd: the axis-separable layout was requested but Q, R or P0 couple different axes");
  return lanes_;
}

int TargetManager::findOrCreateBatch(int type, const double* Q, const double* R, int lanes_code, int& cls) {
  // a batch per (model, layout): at most a handful, so a scan; the (Q, R) class inside it is a hash lookup
  for (size_t b = 0; b < batches_.size(); ++b)
    if (batches_[b]->type() == type && batches_[b]->lanes_code() == lanes_code) {
      cls = batches_[b]->find_class(Q, R);
      if (cls < 0) cls = batches_[b]->add_class(Q, R);
      return (int)b;
    }
  batches_.emplace_back(new Batch(type, dtype_, lanes_code, Q, R, stream_, &target_lock_));
  if (keep_meas_) batches_.back()->set_keep_measurement(true);
  cls = 0;
  return (int)batches_.size() - 1;
}

bool TargetManager::find(unsigned id, Loc& loc) {
  return targets_.find(id, loc);
}

namespace {
// channel order of LogFiles::f / log_all_
const char* const kLogChannel[7] = {"time", "meas_pose", "est_pose", "est_twist", "pose", "est_acc", "covariance"};

// unit quaternion [x y z w] -> rotation matrix (row-major), Eigen's Quaterniond::toRotationMatrix
void host_quat_to_rot(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
// rotToRpy, geometry.hpp:191-196
void host_rot_to_rpy(const double* R, double* rpy) {
  rpy[0] = std::atan2(R[7], R[8]);
  rpy[1] = std::atan2(-R[6], std::sqrt(R[7] * R[7] + R[8] * R[8]));
  rpy[2] = std::atan2(R[3], R[0]);
}
// one row in writeTxtFile's format (utils.hpp:96-120: `myfile << value << " "` per column, then "\n"; default ostream
// formatting = %g with 6 significant digits)
void append_row(std::string& out, const double* v, long w) {
  char buf[40];
  for (long c = 0; c < w; ++c) {
    std::snprintf(buf, sizeof buf, "%g ", v[c]);
    out += buf;
  }
  out += "\n";
}
}  // namespace

void TargetManager::closeLogFiles() {
  for (auto& kv : log_files_)
    for (std::FILE* f : kv.second.f) if (f) std::fclose(f);
  log_files_.clear();
  for (std::FILE*& f : log_all_) { if (f) std::fclose(f); f = nullptr; }
}

void TargetManager::setLogDirectory(const std::string& dir) {
  {
    lock_guard<mutex> lg(target_lock_);
    closeLogFiles();
    log_dir_ = dir;
  }
  if (!dir.empty()) setKeepMeasurement(true);   // the "measurement" channel needs the rows
}

void TargetManager::setLogTargets(const unsigned* ids, long n) {
  lock_guard<mutex> lg(target_lock_);
  closeLogFiles();
  log_ids_.assign(ids, ids + (n > 0 ? n : 0));
  std::sort(log_ids_.begin(), log_ids_.end());
  log_ids_.erase(std::unique(log_ids_.begin(), log_ids_.end()), log_ids_.end());
}

void TargetManager::setKeepMeasurement(bool on) {
  lock_guard<mutex> lg(target_lock_);
  keep_meas_ = on;
  for (auto& b : batches_) b->set_keep_measurement(on);
  dropSeqGraphs();
}

void TargetManager::log() {
  if (log_dir_.empty()) return;
  lock_guard<mutex> lg(target_lock_);
  // what to log: the explicit selection, or everything while the population is small
  std::vector<unsigned> ids = log_ids_;
  const bool per_target = !ids.empty() || (long)targets_.size() <= kLogAutoSelect;
  if (ids.empty()) ids = targets_.sorted_ids();
  // group the ids by batch (slot lists), keep the id order inside a batch
  std::vector<std::vector<int>> slots(batches_.size());
  std::vector<std::vector<unsigned>> who(batches_.size());
  for (unsigned id : ids) {
    Loc loc;
    if (!find(id, loc)) continue;   // a selected target that does not exist (yet, or any more)
    slots[(size_t)loc.batch].push_back(loc.slot);
    who[(size_t)loc.batch].push_back(id);
  }
  std::string all[7];
  for (size_t bi = 0; bi < batches_.size(); ++bi) {
    Batch& b = *batches_[bi];
    const long n = (long)slots[bi].size();
    if (!n) continue;
    const int N = b.n_state();
    std::vector<double> pose((size_t)n * 7), twist((size_t)n * 6), acc((size_t)n * 6), x((size_t)n * N), P((size_t)n * N * N), meas((size_t)n * 7);
    b.outputs(slots[bi].data(), n, pose.data(), twist.data(), acc.data(), false, 0.0);
    b.get_state(slots[bi].data(), n, x.data(), P.data());
    if (b.keep_measurement()) b.measured_poses(slots[bi].data(), n, meas.data());
    else for (long s = 0; s < n; ++s) for (int c = 0; c < 7; ++c) meas[(size_t)s * 7 + c] = c == 6 ? 1.0 : 0.0;
    for (long s = 0; s < n; ++s) {
      const unsigned id = who[bi][(size_t)s];
      const double t = b.time(slots[bi][(size_t)s]);
      double R[9], pose6[6];
      host_quat_to_rot(&pose[(size_t)s * 7 + 3], R);
      for (int c = 0; c < 3; ++c) pose6[c] = pose[(size_t)s * 7 + c];
      host_rot_to_rpy(R, pose6 + 3);   // isometryToPose6d, geometry.hpp:602-608
      const double* row[7] = {&t, &meas[(size_t)s * 7], &pose[(size_t)s * 7], &twist[(size_t)s * 6], pose6, &acc[(size_t)s * 6],
                              &P[(size_t)s * N * N]};
      const long width[7] = {1, 7, 7, 6, 6, 6, (long)N * N};
      if (per_target) {
        LogFiles& lf = log_files_[id];
        for (int ch = 0; ch < 7; ++ch) {
          if (!lf.f[ch]) {
            lf.f[ch] = std::fopen((log_dir_ + "/" + kLogChannel[ch] + "_" + std::to_string(id)).c_str(), "a");
            if (!lf.f[ch]) { std::cerr << "Unable to open file : [" << log_dir_ << "/" << kLogChannel[ch] << "_" << id << "]" << std::endl; continue; }
          }
          std::string line;
          append_row(line, row[ch], width[ch]);
          std::fwrite(line.data(), 1, line.size(), lf.f[ch]);   // one buffered write per channel per call ...
        }
      } else {
        const double idd = (double)id;
        for (int ch = 0; ch < 7; ++ch) {
          char buf[24];
          std::snprintf(buf, sizeof buf, "%g ", idd);
          all[ch] += buf;
          append_row(all[ch], row[ch], width[ch]);
        }
      }
    }
  }
  if (per_target) {
    for (auto& kv : log_files_)
      for (std::FILE* f : kv.second.f) if (f) std::fflush(f);   // ... made visible to readers at the end of the call
  } else {
    for (int ch = 0; ch < 7; ++ch) {
      if (!log_all_[ch]) log_all_[ch] = std::fopen((log_dir_ + "/" + kLogChannel[ch] + "_all").c_str(), "a");
      if (!log_all_[ch]) continue;
      std::fwrite(all[ch].data(), 1, all[ch].size(), log_all_[ch]);
      std::fflush(log_all_[ch]);
    }
  }
}

bool TargetManager::getTargetMeasuredPose(unsigned id, double* pose7) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc) || !batches_[(size_t)loc.batch]->keep_measurement()) return false;
  const int slot = loc.slot;
  batches_[(size_t)loc.batch]->measured_poses(&slot, 1, pose7);
  return true;
}

bool TargetManager::getTargetPeriodEstimate(unsigned id, double& period) {
  double twist[6];
  if (!getTargetTwist(id, twist)) return false;
  const double omega_norm = std::sqrt(twist[3] * twist[3] + twist[4] * twist[4] + twist[5] * twist[5]);
  period = omega_norm > 0 ? 2 * M_PI / omega_norm : -1.0;   // target_interface.cpp:82-86
  return true;
}

bool TargetManager::getTargetTransform(unsigned id, double* T) {
  double pose[7];
  if (!getTargetPose(id, pose)) return false;
  double R[9];
  host_quat_to_rot(pose + 3, R);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[r * 4 + c] = R[r * 3 + c];
    T[r * 4 + 3] = pose[r];
  }
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
  return true;
}

bool TargetManager::getTargetDims(unsigned id, int& n, int& m) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  n = batches_[(size_t)loc.batch]->n_state();
  m = batches_[(size_t)loc.batch]->n_meas();
  return true;
}

bool TargetManager::getTargetModelMatrices(unsigned id, double* Q, double* R, double* P0) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  Batch& b = *batches_[(size_t)loc.batch];
  if (Q || R) b.class_matrices(loc.slot, Q, R);
  if (P0 && !b.initial_covariance(loc.slot, P0)) return false;
  return true;
}

std::vector<unsigned> TargetManager::getAvailableTargets() {
  std::vector<unsigned> ids;
  lock_guard<mutex> lg(target_lock_);
  ids = targets_.sorted_ids();   // ascending, as the reference's std::map iteration
  return ids;
}

size_t TargetManager::size() {
  lock_guard<mutex> lg(target_lock_);
  return targets_.size();
}

bool TargetManager::hasTarget(unsigned id) {
  lock_guard<mutex> lg(target_lock_);
  return targets_.contains(id);
}

void TargetManager::init(unsigned id, double dt0, double t0, const double* p0, const double* v0, const double* a0) {
  if (default_values_loaded_)
    init(default_type_, id, dt0, t0, default_Q_.data(), default_R_.data(), default_P_.data(), p0, v0, a0);
  else
    throw "TargetManager::init failed, can not find default values to load!";
}

void TargetManager::init(target_t type, unsigned id, double dt0, double t0, const double* Q, const double* R,
                         const double* P0, const double* p0, const double* v0, const double* a0) {
  (void)dt0;  // only shapes the constructor's A, which every step rebuilds (uniform_velocity.cpp:40,67)
  lock_guard<mutex> lg(target_lock_);
  if (!targets_.contains(id)) {
    int cls = 0;
    const int b = findOrCreateBatch((int)type, Q, R, chooseLayout((int)type, Q, R, P0, 1), cls);
    const long slot = batches_[(size_t)b]->append(1, &id, t0, P0, false, p0, v0 ? v0 : kZero6, a0 ? a0 : kZero6, cls);
    targets_.set(id, Loc{b, (int)slot});
    dev_ids_.dirty = true;
    if (verbose_) {
      switch (type) {
        case ANGULAR_RATES: std::cout << "Using angular rates for the orientation" << std::endl; break;
        case ANGULAR_VELOCITIES: std::cout << "Using angular velocities for the orientation" << std::endl; break;
        case UNIFORM_ACCELERATION: std::cout << "Uniformly accelerated motion" << std::endl; break;
        case UNIFORM_VELOCITY: std::cout << "Uniform rectilinear motion" << std::endl; break;
      }
    }
  } else
    std::cout << "Target(" << id << ") already exists!" << std::endl;
}

void TargetManager::init(const std::string& file, unsigned id, double dt0, double t0, const double* p0,
                         const double* v0, const double* a0) {
  std::vector<double> Q, R, P;
  target_t type = UNIFORM_VELOCITY;
  if (!loadYamlFile(file, Q, R, P, type)) throw "TargetManager::init failed, can not load the model file!";
  init(type, id, dt0, t0, Q.data(), R.data(), P.data(), p0, v0, a0);
}

long TargetManager::initBatch(const unsigned* ids, long n, double dt0, double t0, const double* p0, const double* v0,
                              const double* a0) {
  if (!default_values_loaded_) throw "TargetManager::init failed, can not find default values to load!";
  return initBatch(default_type_, ids, n, dt0, t0, default_Q_.data(), default_R_.data(), default_P_.data(), false, p0, v0, a0);
}

long TargetManager::initBatch(target_t type, const unsigned* ids, long n, double dt0, double t0, const double* Q,
                              const double* R, const double* P0, bool per_target_P0, const double* p0,
                              const double* v0, const double* a0) {
  (void)dt0;
  lock_guard<mutex> lg(target_lock_);
  const int N = model_n((int)type);
  // keep only ids that do not exist yet (existing ones are left untouched, as in init())
  std::vector<long> keep;
  keep.reserve((size_t)n);
  {
    IdTable seen;
    seen.reserve((size_t)n);
    for (long i = 0; i < n; ++i) {
      if (targets_.contains(ids[i]) || seen.contains(ids[i])) {
        if (verbose_) std::cout << "Target(" << ids[i] << ") already exists!" << std::endl;
        continue;
      }
      seen.set(ids[i], Loc{0, 0});
      keep.push_back(i);
    }
  }
  if (keep.empty()) return 0;
  const long k = (long)keep.size();
  int cls = 0;
  const int b = findOrCreateBatch((int)type, Q, R, chooseLayout((int)type, Q, R, P0, per_target_P0 ? n : 1), cls);
  long first;
  if (k == n) {
    first = batches_[(size_t)b]->append(n, ids, t0, P0, per_target_P0, p0, v0, a0, cls);
  } else {
    std::vector<unsigned> ids2((size_t)k);
    std::vector<double> p2((size_t)k * 7), v2, a2, P2;
    if (v0) v2.resize((size_t)k * 6);
    if (a0) a2.resize((size_t)k * 6);
    if (per_target_P0) P2.resize((size_t)k * N * N);
    for (long j = 0; j < k; ++j) {
      const long i = keep[(size_t)j];
      ids2[(size_t)j] = ids[i];
      std::memcpy(&p2[(size_t)j * 7], p0 + i * 7, sizeof(double) * 7);
      if (v0) std::memcpy(&v2[(size_t)j * 6], v0 + i * 6, sizeof(double) * 6);
      if (a0) std::memcpy(&a2[(size_t)j * 6], a0 + i * 6, sizeof(double) * 6);
      if (per_target_P0) std::memcpy(&P2[(size_t)j * N * N], P0 + i * N * N, sizeof(double) * N * N);
    }
    first = batches_[(size_t)b]->append(k, ids2.data(), t0, per_target_P0 ? P2.data() : P0, per_target_P0, p2.data(),
                                        v0 ? v2.data() : nullptr, a0 ? a2.data() : nullptr, cls);
  }
  targets_.reserve(targets_.size() + (size_t)k);
  for (long j = 0; j < k; ++j) targets_.set(ids[keep[(size_t)j]], Loc{b, (int)(first + j)});
  dev_ids_.dirty = true;
  return k;
}

long TargetManager::initBatchClasses(target_t type, const unsigned* ids, long n, double dt0, double t0, long n_classes,
                                     const double* Q, const double* R, const double* P0, const unsigned* class_of,
                                     const double* p0, const double* v0, const double* a0) {
  (void)dt0;
  lock_guard<mutex> lg(target_lock_);
  if (n <= 0) return 0;
  if (n_classes <= 0) throw std::invalid_argument("target_estimation_amd: initBatchClasses needs at least one class");
  const int N = model_n((int)type), M = model_m((int)type);
  // every class: its layout (the matrices decide) -> batch, and its index inside that batch
  std::vector<int> cls_batch((size_t)n_classes), cls_idx((size_t)n_classes);
  for (long c = 0; c < n_classes; ++c) {
    const double* Qc = Q + c * N * N;
    const double* Rc = R + c * M * M;
    const double* Pc = P0 + c * N * N;
    cls_batch[(size_t)c] = findOrCreateBatch((int)type, Qc, Rc, chooseLayout((int)type, Qc, Rc, Pc, 1), cls_idx[(size_t)c]);
  }
  // new ids only (existing ones are left untouched, as in init()), grouped by destination batch in input order
  std::vector<std::vector<long>> rows(batches_.size());
  {
    IdTable seen;
    seen.reserve((size_t)n);
    for (long i = 0; i < n; ++i) {
      if (class_of[i] >= (unsigned long)n_classes) throw std::invalid_argument("target_estimation_amd: class index out of range");
      if (targets_.contains(ids[i]) || seen.contains(ids[i])) {
        if (verbose_) std::cout << "Target(" << ids[i] << ") already exists!" << std::endl;
        continue;
      }
      seen.set(ids[i], Loc{0, 0});
      rows[(size_t)cls_batch[class_of[i]]].push_back(i);
    }
  }
  long created = 0;
  for (size_t b = 0; b < rows.size(); ++b) {
    const long k = (long)rows[b].size();
    if (!k) continue;
    std::vector<unsigned> ids2((size_t)k);
    std::vector<double> p2((size_t)k * 7), v2(v0 ? (size_t)k * 6 : 0), a2(a0 ? (size_t)k * 6 : 0);
    std::vector<int> cls2((size_t)k), pidx((size_t)k);
    for (long j = 0; j < k; ++j) {
      const long i = rows[b][(size_t)j];
      ids2[(size_t)j] = ids[i];
      std::memcpy(&p2[(size_t)j * 7], p0 + i * 7, sizeof(double) * 7);
      if (v0) std::memcpy(&v2[(size_t)j * 6], v0 + i * 6, sizeof(double) * 6);
      if (a0) std::memcpy(&a2[(size_t)j * 6], a0 + i * 6, sizeof(double) * 6);
      cls2[(size_t)j] = cls_idx[class_of[i]];
      pidx[(size_t)j] = (int)class_of[i];
    }
    const long first = batches_[b]->append(k, ids2.data(), t0, P0, false, p2.data(), v0 ? v2.data() : nullptr,
                                           a0 ? a2.data() : nullptr, 0, cls2.data(), pidx.data(), n_classes);
    targets_.reserve(targets_.size() + (size_t)k);
    for (long j = 0; j < k; ++j) targets_.set(ids2[(size_t)j], Loc{(int)b, (int)(first + j)});
    created += k;
  }
  dev_ids_.dirty = true;
  return created;
}

bool TargetManager::update(unsigned id, double dt, const double* meas) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) {
    std::cout << "Target(" << id << ") does not exist!" << std::endl;
    return false;
  }
  batches_[(size_t)loc.batch]->step_one(loc.slot, dt, meas);
  return true;
}

bool TargetManager::update(unsigned id, double dt) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) {
    std::cout << "Target(" << id << ") does not exist!" << std::endl;
    return false;
  }
  batches_[(size_t)loc.batch]->step_one(loc.slot, dt, nullptr);
  return true;
}

void TargetManager::update(double dt) {
  lock_guard<mutex> lg(target_lock_);
  for (auto& b : batches_) b->step_dense(dt, nullptr, 0, nullptr);
}

bool TargetManager::erase(unsigned id) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) {
    std::cout << "Target(" << id << ") does not exist!" << std::endl;
    return false;
  }
  Batch* b = batches_[(size_t)loc.batch].get();
  const bool was_last = loc.slot == b->size() - 1;
  const unsigned moved = b->erase_slot(loc.slot);
  targets_.erase(id);
  dev_ids_.dirty = true;
  if (!was_last) targets_.set(moved, Loc{loc.batch, loc.slot});
  auto lf = log_files_.find(id);   // a logged target that goes away closes its files (a later target of that id appends)
  if (lf != log_files_.end()) {
    for (std::FILE* f : lf->second.f) if (f) std::fclose(f);
    log_files_.erase(lf);
  }
  return true;
}

long TargetManager::eraseBatch(const unsigned* ids, long n) {
  lock_guard<mutex> lg(target_lock_);
  std::vector<std::vector<int>> slots(batches_.size());
  std::vector<unsigned> erased;
  for (long i = 0; i < n; ++i) {
    Loc loc;
    if (!find(ids[i], loc)) {     // unknown, or already taken by an earlier entry of this call
      std::cout << "Target(" << ids[i] << ") does not exist!" << std::endl;
      continue;
    }
    slots[(size_t)loc.batch].push_back(loc.slot);
    erased.push_back(ids[i]);
    targets_.erase(ids[i]);
  }
  std::vector<std::pair<unsigned, int>> moves;
  dev_ids_.dirty = true;
  for (size_t b = 0; b < batches_.size(); ++b) {
    if (slots[b].empty()) continue;
    batches_[b]->erase_slots(slots[b].data(), (long)slots[b].size(), moves);
    for (auto const& mv : moves) targets_.set(mv.first, Loc{(int)b, mv.second});
  }
  if (!log_files_.empty())
    for (unsigned id : erased) {   // logged targets that went away close their files
      auto lf = log_files_.find(id);
      if (lf == log_files_.end()) continue;
      for (std::FILE* f : lf->second.f) if (f) std::fclose(f);
      log_files_.erase(lf);
    }
  return (long)erased.size();
}

bool TargetManager::getTargetPose(unsigned id, double* pose7) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  batches_[(size_t)loc.batch]->outputs_one(loc.slot, pose7, nullptr, nullptr, false, 0.0);
  return true;
}

bool TargetManager::getTargetTwist(unsigned id, double* twist6) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  batches_[(size_t)loc.batch]->outputs_one(loc.slot, nullptr, twist6, nullptr, false, 0.0);
  return true;
}

bool TargetManager::getTargetAcceleration(unsigned id, double* acc6) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  batches_[(size_t)loc.batch]->outputs_one(loc.slot, nullptr, nullptr, acc6, false, 0.0);
  return true;
}

bool TargetManager::getTargetPoseAt(unsigned id, double t1, double* pose7) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  batches_[(size_t)loc.batch]->outputs_one(loc.slot, pose7, nullptr, nullptr, true, t1);
  return true;
}

bool TargetManager::getTargetTwistAt(unsigned id, double t1, double* twist6) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  batches_[(size_t)loc.batch]->outputs_one(loc.slot, nullptr, twist6, nullptr, true, t1);
  return true;
}

bool TargetManager::getTargetAccelerationAt(unsigned id, double t1, double* a6) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  batches_[(size_t)loc.batch]->outputs_one(loc.slot, nullptr, nullptr, a6, true, t1);
  return true;
}

bool TargetManager::getTargetTime(unsigned id, double& t) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return false;
  t = batches_[(size_t)loc.batch]->time(loc.slot);
  return true;
}

int TargetManager::getTargetState(unsigned id, double* x, double* P) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return 0;
  Batch* b = batches_[(size_t)loc.batch].get();
  b->get_state(&loc.slot, 1, x, P);
  return b->n_state();
}

long long TargetManager::getNumberMeasurements(unsigned id) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (find(id, loc)) return batches_[(size_t)loc.batch]->n_measurements(loc.slot);
  std::cout << "Target(" << id << ") does not exist!" << std::endl;
  return 0;
}

// Node-tick sizes (a few to a thousand targets per call) are a LATENCY path: staging copies and separate launches cost more than
// the step itself (40 targets: 22 us for the dense host path below, 78 us with the getters behind it).  They go through the
// one-target queue instead -- host table look-up per id, one indexed launch at the next read, the queue behind the PCIe BAR and the
// getter table filled by the same launch for up to a wavefront of targets (Batch::flush) -- as a caller looping over the
// reference's own symbols would, minus the call overhead.  Same results (tests/test_gpu_by_id.py, tests/test_gpu_ingest.py).
bool TargetManager::smallBatchPath(const unsigned* ids, long n) const {
  // TE_SMALL_BATCH_QUEUE=<n>: the largest call that takes this path (0 switches it off; the comparison in profiles/)
  static const long most = [] { const char* e = std::getenv("TE_SMALL_BATCH_QUEUE"); return e && *e ? std::atol(e) : kSmallBatchQueue; }();
  if (n <= 0 || n > most) return false;
  for (const auto& b : batches_)
    if (!b->getter_table_is_cheap()) return false;   // (a batch too large for a host-resident getter table: the bulk paths below)
  (void)ids;
  return true;
}

long TargetManager::updateBatch(const unsigned* ids, long n, double dt, const double* meas, const unsigned char* has_meas) {
  lock_guard<mutex> lg(target_lock_);
  const size_t nb = batches_.size();
  if (smallBatchPath(ids, n)) {
    long done = 0;
    for (long i = 0; i < n; ++i) {
      Loc loc;
      if (!find(ids[i], loc)) {
        if (verbose_) std::cout << "Target(" << ids[i] << ") does not exist!" << std::endl;
        continue;
      }
      batches_[(size_t)loc.batch]->step_one(loc.slot, dt, (meas && (!has_meas || has_meas[i])) ? meas + 7 * i : nullptr);
      ++done;
    }
    return done;
  }
  // fast path: the caller passes exactly one batch's ids in slot order (the usual case when the same
  // id array is reused every tick): no per-id lookup, dense kernel
  for (size_t b = 0; b < nb; ++b) {
    Batch* bt = batches_[b].get();
    if (bt->size() == n && n > 0 && std::memcmp(ids, bt->slot_ids().data(), sizeof(unsigned) * (size_t)n) == 0) {
      bt->step_dense_host(dt, meas, has_meas);
      return n;
    }
  }
  // ids in any order, possibly several batches, possibly unknown ids: resolved on the device (id_resolve.hpp); a call
  // that names an id twice keeps the reference's "two consecutive steps" through the host path below
  if (n >= kDevResolveMin && !verbose_) {
    ResolveCounters rc;
    if (resolveOnDevice(ids, n, rc) && !rc.duplicate) {
      DevIds& d = dev_ids_;
      if (meas) {
        TE_HIP_CHECK(hipMemcpyAsync(d.aos, meas, sizeof(double) * 7 * n, hipMemcpyHostToDevice, stream_));
        batches_[0]->pack_meas_dev(d.aos, n, d.soa, n);
      }
      if (meas && has_meas) TE_HIP_CHECK(hipMemcpyAsync(d.mask, has_meas, (size_t)n, hipMemcpyHostToDevice, stream_));
      long total = 0;
      for (size_t b = 0; b < nb; ++b) {
        if (rc.found[b] <= 0) continue;
        total += rc.found[b];
        hipLaunchKernelGGL(id_select_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream_, d.loc, n, (int)b, d.idx,
                           (unsigned char*)nullptr);
        batches_[b]->step_indexed_dev(d.idx, n, dt, meas ? d.soa : nullptr, n, (meas && has_meas) ? d.mask : nullptr);
      }
      TE_HIP_CHECK(hipStreamSynchronize(stream_));   // the caller's host arrays may be reused after return
      return total;
    }
  }
  std::vector<std::vector<int>> slots(nb);
  std::vector<std::vector<long>> src(nb);
  std::vector<std::vector<unsigned char>> seen(nb);
  for (size_t b = 0; b < nb; ++b) seen[b].assign((size_t)batches_[b]->size(), 0);
  long done = 0;
  auto flush = [&](size_t b) {
    const long k = (long)slots[b].size();
    if (!k) return;
    std::vector<double> m2;
    std::vector<unsigned char> h2;
    if (meas) {
      m2.resize((size_t)k * 7);
      for (long j = 0; j < k; ++j) std::memcpy(&m2[(size_t)j * 7], meas + src[b][(size_t)j] * 7, sizeof(double) * 7);
    }
    if (has_meas) {
      h2.resize((size_t)k);
      for (long j = 0; j < k; ++j) h2[(size_t)j] = has_meas[src[b][(size_t)j]];
    }
    batches_[b]->step_indexed(slots[b].data(), k, dt, meas ? m2.data() : nullptr, has_meas ? h2.data() : nullptr);
    for (int s : slots[b]) seen[b][(size_t)s] = 0;
    slots[b].clear();
    src[b].clear();
  };
  for (long i = 0; i < n; ++i) {
    Loc loc;
    if (!find(ids[i], loc)) {
      if (verbose_) std::cout << "Target(" << ids[i] << ") does not exist!" << std::endl;
      continue;
    }
    const size_t b = (size_t)loc.batch;
    // the same id twice in one call = two consecutive steps, as the reference's loop over ids would
    // do: everything queued so far for that batch goes first
    if (seen[b][(size_t)loc.slot]) flush(b);
    seen[b][(size_t)loc.slot] = 1;
    slots[b].push_back(loc.slot);
    src[b].push_back(i);
    ++done;
  }
  for (size_t b = 0; b < nb; ++b) {
    const long k = (long)slots[b].size();
    if (!k) continue;
    const bool contiguous = (k == n);  // single batch, every id known: rows already in order
    if (contiguous) {
      batches_[b]->step_indexed(slots[b].data(), k, dt, meas, has_meas);
    } else {
      std::vector<double> m2;
      std::vector<unsigned char> h2;
      if (meas) {
        m2.resize((size_t)k * 7);
        for (long j = 0; j < k; ++j) std::memcpy(&m2[(size_t)j * 7], meas + src[b][(size_t)j] * 7, sizeof(double) * 7);
      }
      if (has_meas) {
        h2.resize((size_t)k);
        for (long j = 0; j < k; ++j) h2[(size_t)j] = has_meas[src[b][(size_t)j]];
      }
      batches_[b]->step_indexed(slots[b].data(), k, dt, meas ? m2.data() : nullptr, has_meas ? h2.data() : nullptr);
    }
  }
  return done;
}

long TargetManager::getPoseBatch(const unsigned* ids, long n, double* pose, double* twist, double* acc,
                                 unsigned char* found, bool at_time, double t1) {
  lock_guard<mutex> lg(target_lock_);
  const size_t nb = batches_.size();
  if (!at_time && smallBatchPath(ids, n)) {   // rows from the host-resident getter table (filled by the flush's own launch)
    long done = 0;
    for (long i = 0; i < n; ++i) {
      Loc loc;
      const bool ok = find(ids[i], loc);
      if (found) found[i] = ok ? 1 : 0;
      if (!ok) continue;
      batches_[(size_t)loc.batch]->outputs_one(loc.slot, pose ? pose + 7 * i : nullptr, twist ? twist + 6 * i : nullptr, acc ? acc + 6 * i : nullptr, false, 0.0);
      ++done;
    }
    return done;
  }
  for (size_t b = 0; b < nb; ++b) {   // same fast path as updateBatch
    Batch* bt = batches_[b].get();
    if (bt->size() == n && n > 0 && std::memcmp(ids, bt->slot_ids().data(), sizeof(unsigned) * (size_t)n) == 0) {
      bt->outputs(nullptr, n, pose, twist, acc, at_time, t1);
      if (found) std::memset(found, 1, (size_t)n);
      return n;
    }
  }
  if (n >= kDevResolveMin) {   // ids resolved on the device; rows come back in the caller's order
    ResolveCounters rc;
    if (resolveOnDevice(ids, n, rc)) {
      DevIds& d = dev_ids_;
      long total = 0;
      double* dp = pose ? d.out : nullptr;
      double* dtw = twist ? d.out + 7 * n : nullptr;
      double* da = acc ? d.out + 13 * n : nullptr;
      for (size_t b = 0; b < nb; ++b) {
        if (rc.found[b] <= 0) continue;
        total += rc.found[b];
        hipLaunchKernelGGL(id_select_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream_, d.loc, n, (int)b, d.idx,
                           (unsigned char*)nullptr);
        batches_[b]->outputs_indexed_dev(d.idx, n, dp, dtw, da, at_time, t1);
      }
      if (total == n) {   // every id known: straight into the caller's arrays
        if (pose) TE_HIP_CHECK(hipMemcpyAsync(pose, dp, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, stream_));
        if (twist) TE_HIP_CHECK(hipMemcpyAsync(twist, dtw, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, stream_));
        if (acc) TE_HIP_CHECK(hipMemcpyAsync(acc, da, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, stream_));
        TE_HIP_CHECK(hipStreamSynchronize(stream_));
        if (found) std::memset(found, 1, (size_t)n);
      } else {            // rows of unknown ids stay as the caller left them
        std::vector<double> hp(pose ? (size_t)n * 7 : 0), ht(twist ? (size_t)n * 6 : 0), ha(acc ? (size_t)n * 6 : 0);
        std::vector<int> hloc((size_t)n);
        if (pose) TE_HIP_CHECK(hipMemcpyAsync(hp.data(), dp, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, stream_));
        if (twist) TE_HIP_CHECK(hipMemcpyAsync(ht.data(), dtw, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, stream_));
        if (acc) TE_HIP_CHECK(hipMemcpyAsync(ha.data(), da, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, stream_));
        TE_HIP_CHECK(hipMemcpyAsync(hloc.data(), d.loc, sizeof(int) * n, hipMemcpyDeviceToHost, stream_));
        TE_HIP_CHECK(hipStreamSynchronize(stream_));
        for (long i = 0; i < n; ++i) {
          const bool ok = hloc[(size_t)i] >= 0;
          if (found) found[i] = ok ? 1 : 0;
          if (!ok) continue;
          if (pose) std::memcpy(pose + i * 7, &hp[(size_t)i * 7], sizeof(double) * 7);
          if (twist) std::memcpy(twist + i * 6, &ht[(size_t)i * 6], sizeof(double) * 6);
          if (acc) std::memcpy(acc + i * 6, &ha[(size_t)i * 6], sizeof(double) * 6);
        }
      }
      return total;
    }
  }
  std::vector<std::vector<int>> slots(nb);
  std::vector<std::vector<long>> src(nb);
  long done = 0;
  for (long i = 0; i < n; ++i) {
    Loc loc;
    const bool ok = find(ids[i], loc);
    if (found) found[i] = ok ? 1 : 0;
    if (!ok) continue;
    slots[(size_t)loc.batch].push_back(loc.slot);
    src[(size_t)loc.batch].push_back(i);
    ++done;
  }
  for (size_t b = 0; b < nb; ++b) {
    const long k = (long)slots[b].size();
    if (!k) continue;
    if (k == n) {
      batches_[b]->outputs(slots[b].data(), k, pose, twist, acc, at_time, t1);
      continue;
    }
    std::vector<double> p2(pose ? (size_t)k * 7 : 0), t2(twist ? (size_t)k * 6 : 0), a2(acc ? (size_t)k * 6 : 0);
    batches_[b]->outputs(slots[b].data(), k, pose ? p2.data() : nullptr, twist ? t2.data() : nullptr,
                         acc ? a2.data() : nullptr, at_time, t1);
    for (long j = 0; j < k; ++j) {
      const long i = src[b][(size_t)j];
      if (pose) std::memcpy(pose + i * 7, &p2[(size_t)j * 7], sizeof(double) * 7);
      if (twist) std::memcpy(twist + i * 6, &t2[(size_t)j * 6], sizeof(double) * 6);
      if (acc) std::memcpy(acc + i * 6, &a2[(size_t)j * 6], sizeof(double) * 6);
    }
  }
  return done;
}

long TargetManager::getStateBatch(const unsigned* ids, long n, double* x, double* P) {
  lock_guard<mutex> lg(target_lock_);
  if (n <= 0) return 0;
  std::vector<int> slots((size_t)n);
  int b0 = -1;
  for (long i = 0; i < n; ++i) {
    Loc loc;
    if (!find(ids[i], loc)) return -1;
    if (b0 < 0) b0 = loc.batch;
    if (loc.batch != b0) return -2;  // all ids must belong to one batch (one state size)
    slots[(size_t)i] = loc.slot;
  }
  batches_[(size_t)b0]->get_state(slots.data(), n, x, P);
  return batches_[(size_t)b0]->n_state();
}

double TargetManager::getIntersectionTimeWithSphere(unsigned id, double t1, const double* origin, double radius) {
  lock_guard<mutex> lg(target_lock_);
  Loc loc;
  if (!find(id, loc)) return -1;
  double d = -1;
  batches_[(size_t)loc.batch]->intersect(&loc.slot, 1, t1, origin, radius, &d, nullptr);
  return d;
}

bool TargetManager::getIntersectionPoseWithSphere(unsigned id, double t1, const double* origin, double radius,
                                                  double* pose7, double* delta) {
  lock_guard<mutex> lg(target_lock_);
  pose7[0] = pose7[1] = pose7[2] = pose7[3] = pose7[4] = pose7[5] = 0.0;
  pose7[6] = 1.0;  // initPose, intersection_solver.cpp:99
  if (delta) *delta = -1;
  Loc loc;
  if (!find(id, loc)) return false;
  double d = -1;
  batches_[(size_t)loc.batch]->intersect(&loc.slot, 1, t1, origin, radius, &d, pose7);
  if (delta) *delta = d;
  return d > -1;
}

bool TargetManager::getIntersectionPoseWithSphere(unsigned id, double t1, double pos_th, double ang_th,
                                                  const double* origin, double radius, double* pose7) {
  unsigned char conv = 0, found = 0;
  double delta = -1;
  intersectGatedBatch(&id, 1, t1, pos_th, ang_th, origin, radius, &delta, pose7, &conv, &found);
  return conv != 0;
}

long TargetManager::intersectGatedBatch(const unsigned* ids, long n, double t1, double pos_th, double ang_th,
                                        const double* origin, double radius, double* delta, double* pose,
                                        unsigned char* converged, unsigned char* found, double* filt) {
  lock_guard<mutex> lg(target_lock_);
  const size_t nb = batches_.size();
  std::vector<std::vector<int>> slots(nb);
  std::vector<std::vector<long>> src(nb);
  long done = 0;
  for (long i = 0; i < n; ++i) {
    Loc loc;
    const bool ok = find(ids[i], loc);
    if (found) found[i] = ok ? 1 : 0;
    if (delta) delta[i] = -1;
    if (converged) converged[i] = 0;
    if (pose) { for (int c = 0; c < 6; ++c) pose[i * 7 + c] = 0.0; pose[i * 7 + 6] = 1.0; }
    if (filt) { filt[i * 2] = 0.0; filt[i * 2 + 1] = 0.0; }
    if (!ok) continue;
    slots[(size_t)loc.batch].push_back(loc.slot);
    src[(size_t)loc.batch].push_back(i);
    ++done;
  }
  for (size_t b = 0; b < nb; ++b) {
    const long k = (long)slots[b].size();
    if (!k) continue;
    std::vector<double> d2((size_t)k), p2((size_t)k * 7), f2(filt ? (size_t)k * 2 : 0);
    std::vector<unsigned char> c2((size_t)k);
    batches_[b]->intersect_gated(slots[b].data(), k, t1, origin, radius, pos_th, ang_th, filters_length_, d2.data(),
                                 p2.data(), c2.data(), filt ? f2.data() : nullptr);
    for (long j = 0; j < k; ++j) {
      const long i = src[b][(size_t)j];
      if (delta) delta[i] = d2[(size_t)j];
      if (converged) converged[i] = c2[(size_t)j];
      if (pose) std::memcpy(pose + i * 7, &p2[(size_t)j * 7], sizeof(double) * 7);
      if (filt) { filt[i * 2] = f2[(size_t)j * 2]; filt[i * 2 + 1] = f2[(size_t)j * 2 + 1]; }
    }
  }
  return done;
}

long TargetManager::intersectBatch(const unsigned* ids, long n, double t1, const double* origin, double radius,
                                   double* delta, double* pose, unsigned char* found) {
  lock_guard<mutex> lg(target_lock_);
  const size_t nb = batches_.size();
  std::vector<std::vector<int>> slots(nb);
  std::vector<std::vector<long>> src(nb);
  long done = 0;
  for (long i = 0; i < n; ++i) {
    Loc loc;
    const bool ok = find(ids[i], loc);
    if (found) found[i] = ok ? 1 : 0;
    delta[i] = -1;
    if (pose) { for (int c = 0; c < 6; ++c) pose[i * 7 + c] = 0.0; pose[i * 7 + 6] = 1.0; }
    if (!ok) continue;
    slots[(size_t)loc.batch].push_back(loc.slot);
    src[(size_t)loc.batch].push_back(i);
    ++done;
  }
  for (size_t b = 0; b < nb; ++b) {
    const long k = (long)slots[b].size();
    if (!k) continue;
    std::vector<double> d2((size_t)k), p2(pose ? (size_t)k * 7 : 0);
    batches_[b]->intersect(slots[b].data(), k, t1, origin, radius, d2.data(), pose ? p2.data() : nullptr);
    for (long j = 0; j < k; ++j) {
      const long i = src[b][(size_t)j];
      delta[i] = d2[(size_t)j];
      if (pose) std::memcpy(pose + i * 7, &p2[(size_t)j * 7], sizeof(double) * 7);
    }
  }
  return done;
}

Batch* TargetManager::batchOfType(int type) {
  for (auto& b : batches_)
    if (b->type() == type) return b.get();
  return nullptr;
}

long TargetManager::posesToDevice(double* out_dev, long capacity, hipStream_t st) {
  lock_guard<mutex> lg(target_lock_);
  long rows = 0;
  for (auto& b : batches_) rows += b->size();
  if (!out_dev) return rows;
  if (capacity < rows) throw std::invalid_argument("target_estimation_amd: posesToDevice: buffer too small");
  if (st != stream_) throw std::invalid_argument("target_estimation_amd: posesToDevice runs on the manager's stream");
  long off = 0;
  for (auto& b : batches_) {
    if (!b->size()) continue;
    b->outputs_dev(out_dev + off * 7, nullptr, nullptr, false, 0.0);
    off += b->size();
  }
  return rows;
}

long TargetManager::posesForGather(long expect_rows, const std::function<double*(long, hipStream_t)>& prepare) {
  lock_guard<mutex> lg(target_lock_);   // count, stream and the outputs launches in ONE critical section
  long rows = 0;
  for (auto& b : batches_) rows += b->size();
  if (rows != expect_rows) throw std::invalid_argument("target_estimation_amd: gather: counts[rank] differs from the manager's size");
  double* out_dev = prepare(rows, stream_);
  if (!out_dev && rows > 0) throw std::invalid_argument("target_estimation_amd: gather: no destination for the pose rows");
  long off = 0;
  for (auto& b : batches_) {
    if (!b->size()) continue;
    b->outputs_dev(out_dev + off * 7, nullptr, nullptr, false, 0.0);
    off += b->size();
  }
  return rows;
}

void TargetManager::setStream(hipStream_t s) {
  lock_guard<mutex> lg(target_lock_);
  for (auto& b : batches_) { b->synchronize(); b->set_stream(s); }
  stream_ = s;
}

void TargetManager::dropSeqGraphs() {
  for (auto& g : seq_graphs_) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
  seq_graphs_.clear();
}

// Can the tick of all batches be ONE launch?  At least two non-empty batches, every one of them a one-class batch in the
// separable layout with packed groups (the automatic choice for the shipped models) -- then there is at most one batch per motion
// model.  TE_POPULATION_TICK=0 keeps the launch per batch (experiments, and the comparison in profiles/).
bool TargetManager::populationTick() const {
  static const bool on = [] { const char* e = std::getenv("TE_POPULATION_TICK"); return !(e && e[0] == '0'); }();
  if (!on) return false;
  int present = 0;
  bool seen[4] = {false, false, false, false};
  for (const auto& b : batches_) {
    if (b->size() == 0) continue;
    if (!b->population_ready() || b->type() < 0 || b->type() > 3 || seen[b->type()]) return false;
    seen[b->type()] = true;
    ++present;
  }
  return present >= 2;
}

void TargetManager::enqueuePopulationTick(hipStream_t st, long s, double dt, const Batch::SeqSpec* specs, bool query, const double* origin,
                                          double radius, bool reverse, bool ab) {
  StepParams parts[4];
  for (auto& q : parts) { q = StepParams{}; q.n = 0; q.idx = nullptr; }
  Batch* swap[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ab_all = ab;
  for (int pass = 0; pass < 2; ++pass) {   // (a batch without room for its second record buffer puts the whole tick in place)
    for (size_t b = 0; b < batches_.size(); ++b) {
      if (batches_[b]->size() == 0) continue;
      const int t = batches_[b]->type();
      parts[t] = batches_[b]->tick_params(s, dt, specs[b], query, origin, radius, ab_all);
      if (ab_all && !parts[t].rec_out) { ab_all = false; break; }
      swap[t] = batches_[b].get();
    }
    if (ab_all == ab || pass == 1) break;
  }
  if (!ab_all) for (auto& q : parts) q.rec_out = nullptr;
  launch_population_step(dtype_, parts, query, ab_all, reverse, st);
  if (ab_all) for (auto* b : swap) if (b) b->swap_records();
}

void TargetManager::stepSequenceAll(long n_ticks, double dt, const Batch::SeqSpec* specs, long n_specs, bool query,
                                    const double* origin, double radius, int use_graph) {
  lock_guard<mutex> lg(target_lock_);
  const size_t nb = batches_.size();
  if ((size_t)n_specs != nb) throw std::runtime_error("target_estimation_amd: stepSequenceAll needs one spec per batch");
  if (n_ticks <= 0 || nb == 0) return;
  if (query && !origin) throw std::runtime_error("target_estimation_amd: stepSequenceAll: query without an origin");
  for (size_t b = 0; b < nb; ++b) {
    if (query && batches_[b]->size() > 0 && !specs[b].delta_dev)
      throw std::runtime_error("target_estimation_amd: stepSequenceAll: query without a delta output");
    batches_[b]->prepare();   // queued one-target steps run first
  }
  const double zero3[3] = {0, 0, 0};
  const double* org = origin ? origin : zero3;
  if (!use_graph) {
    // Zig-zag over the WHOLE tick: tick s walks batch 0 .. nb-1, tiles forwards; tick s+1 walks batch nb-1 .. 0, tiles
    // backwards, so that what the Infinity Cache still holds at the end of a tick is what the next tick reads first.
    long state = 0;
    for (size_t b = 0; b < nb; ++b) state += batches_[b]->state_bytes();
    const bool zz = state >= Batch::zigzag_min_bytes();   // L2-resident populations keep their tile -> XCD affinity
    // A -> B ticks (Batch::pingpong_min_bytes) by the size of the WHOLE population: what decides is how much is streamed
    // between two uses of a record, not which batch it belongs to
    const bool ab = Batch::pingpong_min_bytes() >= 0 && state >= Batch::pingpong_min_bytes();
    const bool pop = populationTick();
    for (long s = 0; s < n_ticks; ++s) {
      const bool rev = zz && seq_flip_;
      if (pop) {
        enqueuePopulationTick(stream_, s, dt, specs, query, org, radius, rev, ab && !query);
      } else {
        for (size_t k = 0; k < nb; ++k) {
          const size_t b = rev ? nb - 1 - k : k;
          batches_[b]->enqueue_tick(stream_, s, dt, specs[b], query, org, radius, rev, ab);
        }
      }
      seq_flip_ = !seq_flip_;
    }
    TE_HIP_CHECK(hipGetLastError());
  } else {
    auto same_spec = [](const Batch::SeqSpec& x, const Batch::SeqSpec& y) {
      return x.meas_base == y.meas_base && x.tick_stride == y.tick_stride && x.ld == y.ld && x.has_base == y.has_base &&
             x.has_stride == y.has_stride && x.delta_dev == y.delta_dev && x.pose_dev == y.pose_dev && x.ring_ticks == y.ring_ticks;
    };
    auto same_id = [](const Batch::DevIdentity& x, const Batch::DevIdentity& y) {
      return x.rec == y.rec && x.qr == y.qr && x.tbase == y.tbase && x.nmbase == y.nmbase && x.n == y.n;
    };
    SeqGraph* hit = nullptr;
    for (auto& g : seq_graphs_) {
      if (g.n_ticks != n_ticks || g.dt != dt || g.query != query || g.specs.size() != nb) continue;
      if (query && (g.origin[0] != org[0] || g.origin[1] != org[1] || g.origin[2] != org[2] || g.radius != radius)) continue;
      bool ok = true;
      for (size_t b = 0; b < nb && ok; ++b) ok = same_spec(g.specs[b], specs[b]) && same_id(g.ident[b], batches_[b]->dev_identity());
      if (ok) { hit = &g; break; }
    }
    if (!hit) {
      if (seq_graphs_.size() >= 8) {   // evict the oldest recording (they are appended in order of creation)
        TE_HIP_CHECK(hipStreamSynchronize(stream_));
        (void)hipGraphExecDestroy(seq_graphs_.front().exec);
        (void)hipGraphDestroy(seq_graphs_.front().graph);
        seq_graphs_.erase(seq_graphs_.begin());
      }
      if (branch_streams_.empty()) {
        hipStream_t st; TE_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        branch_streams_.push_back(st);
      }
      SeqGraph g;
      g.n_ticks = n_ticks; g.dt = dt; g.query = query; g.radius = radius;
      g.origin[0] = org[0]; g.origin[1] = org[1]; g.origin[2] = org[2];
      g.specs.assign(specs, specs + nb);
      for (size_t b = 0; b < nb; ++b) g.ident.push_back(batches_[b]->dev_identity());
      g.graph = nullptr; g.exec = nullptr;
      // The launches are captured on ONE stream whose dependency set is replaced at the head of every
      // batch's chain (hipStreamUpdateCaptureDependencies): the chains become the branches of the graph.
      hipStream_t cap = branch_streams_[0];
      using Nodes = std::vector<hipGraphNode_t>;
      auto set_deps = [&](Nodes& deps) {
        TE_HIP_CHECK(hipStreamUpdateCaptureDependencies(cap, deps.empty() ? nullptr : deps.data(), deps.size(), hipStreamSetCaptureDependencies));
      };
      auto captured = [&]() {   // the node(s) the next launch would depend on = what was just captured
        hipStreamCaptureStatus status; unsigned long long id = 0; hipGraph_t gr = nullptr;
        const hipGraphNode_t* deps = nullptr; size_t n = 0;
        TE_HIP_CHECK(hipStreamGetCaptureInfo_v2(cap, &status, &id, &gr, &deps, &n));
        return Nodes(deps, deps + n);
      };
      TE_HIP_CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
      try {
        Nodes leaves, none;
        if (populationTick()) {
          // one launch per tick for the whole population: a single chain, nothing left to the placement of branches on
          // hardware queues (kf_step_sep.hpp, kf_step_population_kernel)
          long state = 0;
          for (size_t b = 0; b < nb; ++b) state += batches_[b]->state_bytes();
          const bool zz = state >= Batch::zigzag_min_bytes();
          for (long s = 0; s < n_ticks; ++s) enqueuePopulationTick(cap, s, dt, specs, query, org, radius, zz && (s & 1) != 0, false);
        } else
        for (size_t b = 0; b < nb; ++b) {
          if (batches_[b]->size() == 0) continue;
          set_deps(none);                                  // a new chain: no predecessor
          const bool zz = batches_[b]->state_bytes() >= Batch::zigzag_min_bytes();
          for (long s = 0; s < n_ticks; ++s) batches_[b]->enqueue_tick(cap, s, dt, specs[b], query, org, radius, zz && (s & 1) != 0);
#ifdef TE_TEST_HOOKS   // only in libtarget_estimation_amd_testhooks.so (csrc/Makefile `testhooks`), never in the product library
          if (std::getenv("TE_TEST_FAIL_IN_CAPTURE")) throw std::runtime_error("target_estimation_amd: injected failure inside stream capture");
#endif
          const Nodes tail = captured();
          leaves.insert(leaves.end(), tail.begin(), tail.end());
        }
        TE_HIP_CHECK(hipGetLastError());
        if (!leaves.empty()) set_deps(leaves);
      } catch (...) {
        hipGraph_t broken = nullptr;
        (void)hipStreamEndCapture(cap, &broken);   // leave capture mode before reporting
        if (broken) (void)hipGraphDestroy(broken);
        throw;
      }
      TE_HIP_CHECK(hipStreamEndCapture(cap, &g.graph));
      TE_HIP_CHECK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
      seq_graphs_.push_back(std::move(g));
      hit = &seq_graphs_.back();
    }
    if (use_graph == 2) return;
    TE_HIP_CHECK(hipGraphLaunch(hit->exec, stream_));
  }
  for (size_t b = 0; b < nb; ++b)
    if (batches_[b]->size() > 0) batches_[b]->account_sequence(n_ticks, dt, specs[b].meas_base && !specs[b].has_base);
}

void TargetManager::liveStartAll(double dt, const Batch::SeqSpec* specs, long n_specs, long first_entry, long max_ticks, double idle_limit_s,
                                 bool query, const double* origin, double radius) {
  lock_guard<mutex> lg(target_lock_);
  const size_t nb = batches_.size();
  if ((size_t)n_specs != nb || nb == 0) throw std::runtime_error("target_estimation_amd: liveStartAll needs one spec per batch");
  if (query && !origin) throw std::runtime_error("target_estimation_amd: liveStartAll: query without an origin");
  for (size_t b = 0; b < nb; ++b)
    if (query && !specs[b].delta_dev) throw std::runtime_error("target_estimation_amd: liveStartAll: query without a delta output");
  double share = 0.0;
  for (size_t b = 0; b < nb; ++b) {
    if (batches_[b]->size() == 0) throw std::runtime_error("target_estimation_amd: liveStartAll: an empty batch");
    if (specs[b].ring_ticks <= 0) throw std::invalid_argument("target_estimation_amd: liveStartAll: every batch needs a measurement ring");
    const long cap = batches_[b]->live_capacity_targets(query || batches_[b]->live_pose_output_set());
    if (cap <= 0) throw std::runtime_error("target_estimation_amd: live mode needs the axis-separable layout with packed groups (batch " + std::to_string(b) + ")");
    share += (double)(batches_[b]->size() + batches_[b]->layout().tpw) / (double)cap;   // + one tile for the relay wavefront
  }
  if (share > 1.0)
    throw std::runtime_error("target_estimation_amd: liveStartAll: the batches' resident kernels do not fit the device together (" +
                             std::to_string(share) + " of its capacity)");
  size_t started = 0;
  try {
    for (; started < nb; ++started)
      batches_[started]->live_start(dt, specs[started].meas_base, specs[started].tick_stride, specs[started].ld, specs[started].has_base,
                                    specs[started].has_stride, specs[started].ring_ticks, first_entry, max_ticks, idle_limit_s,
                                    query ? origin : nullptr, radius, query ? specs[started].delta_dev : nullptr,
                                    query ? specs[started].pose_dev : nullptr);
    // side by side, or not at all: a kernel that could only start because an earlier one gave up (one hardware queue for all
    // of them and an idle limit shorter than the start timeout) is not a session
    for (size_t b = 0; b < nb; ++b)
      if (!batches_[b]->live_running())
        throw std::runtime_error("target_estimation_amd: liveStartAll: the batches' resident kernels do not run side by side (batch " + std::to_string(b) +
                                 " has ended already: they share a hardware queue -- more live batches than GPU_MAX_HW_QUEUES?)");
  } catch (...) {
    for (size_t b = 0; b < started; ++b) { try { batches_[b]->live_stop(); } catch (...) {} }
    throw;
  }
}

void TargetManager::livePostAll(long n_ticks, bool one_doorbell_per_tick) {
  lock_guard<mutex> lg(target_lock_);
  if (one_doorbell_per_tick) {
    for (long i = 0; i < n_ticks; ++i)
      for (auto& b : batches_) b->live_post(1);
  } else {
    for (auto& b : batches_) b->live_post(n_ticks);
  }
}

long TargetManager::liveDoneAll() {
  lock_guard<mutex> lg(target_lock_);
  long mn = -1;
  for (auto& b : batches_) {
    if (!b->live_active()) continue;
    const long d = b->live_done();
    mn = mn < 0 ? d : std::min(mn, d);
  }
  return mn < 0 ? 0 : mn;
}

bool TargetManager::liveWaitAll(long tick, double timeout_s) {
  std::vector<Batch*> open;   // the list under the lock, the spinning without it (posts come from other threads)
  {
    lock_guard<mutex> lg(target_lock_);
    for (auto& b : batches_) if (b->live_active()) open.push_back(b.get());
  }
  for (Batch* b : open)
    if (!b->live_wait(tick, timeout_s)) return false;
  return true;
}

long TargetManager::liveStopAll() {
  lock_guard<mutex> lg(target_lock_);
  long served = -1;
  std::string err;
  for (auto& b : batches_) {
    if (!b->live_active()) continue;
    try {
      const long k = b->live_stop();
      if (served >= 0 && k != served) err = "target_estimation_amd: liveStopAll: the batches served different numbers of ticks";
      served = k;
    } catch (const std::exception& e) {
      err = e.what();
    }
  }
  if (!err.empty()) throw std::runtime_error(err);
  return served < 0 ? 0 : served;
}

void TargetManager::synchronize() {
  lock_guard<mutex> lg(target_lock_);
  for (auto& b : batches_) b->synchronize();
}

}  // namespace te
