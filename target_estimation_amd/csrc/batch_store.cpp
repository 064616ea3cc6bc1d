// batch_store.cpp -- see batch_store.hpp.
#include "batch_store.hpp"
#include "measured_pose.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "hip_check.hpp"
#include "intersect_gate.hpp"

namespace te {

namespace {
std::mutex g_live_mu;
std::vector<void*> g_deferred_frees;
int g_live_sessions = 0;
double g_live_share = 0.0;
}  // namespace

void device_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lg(g_live_mu);
    if (g_live_sessions > 0) { g_deferred_frees.push_back(p); return; }
  }
  (void)hipFree(p);
}

bool live_session_begin(double share) {
  std::lock_guard<std::mutex> lg(g_live_mu);
  const char* e = std::getenv("TE_LIVE_IGNORE_OTHERS");   // tests: the start-word path of a grid that does not become resident
  if (!(e && e[0] == '1') && g_live_sessions > 0 && g_live_share + share > 1.0) return false;
  ++g_live_sessions;
  g_live_share += share;
  return true;
}

void live_session_ended(double share) {
  std::vector<void*> drain;
  {
    std::lock_guard<std::mutex> lg(g_live_mu);
    g_live_share -= share;
    if (--g_live_sessions > 0) return;
    g_live_sessions = 0; g_live_share = 0.0;
    drain.swap(g_deferred_frees);
  }
  for (void* p : drain) (void)hipFree(p);
}

double live_sessions_share() {
  std::lock_guard<std::mutex> lg(g_live_mu);
  return g_live_share;
}


const Ops* get_ops(int type, int dtype, int g) {
  switch (type) {
    case UNIFORM_VELOCITY: return get_ops_uv(dtype, g);
    case UNIFORM_ACCELERATION: return get_ops_ua(dtype, g);
    case ANGULAR_RATES: return get_ops_ar(dtype, g);
    case ANGULAR_VELOCITIES: return get_ops_av(dtype, g);
    default: return nullptr;
  }
}

Batch::Batch(int type, int dtype, int lanes, const double* Q, const double* R, hipStream_t stream, std::mutex* owner_lock)
    : type_(type), dtype_(dtype), lanes_code_(lanes), owner_lock_(owner_lock), ops_(get_ops(type, dtype, lanes)), stream_(stream) {
  if (!ops_) throw std::runtime_error("target_estimation_amd: unsupported (model, precision, lanes-per-target) combination");
  add_class(Q, R);
}

static std::string class_key(const double* Q, int nq, const double* R, int nr) {
  std::string k((size_t)(nq + nr) * sizeof(double), '\0');
  std::memcpy(&k[0], Q, (size_t)nq * sizeof(double));
  std::memcpy(&k[(size_t)nq * sizeof(double)], R, (size_t)nr * sizeof(double));
  return k;
}

int Batch::find_class(const double* Q, const double* R) const {
  const int n = ops_->L.n, m = ops_->L.m;
  auto it = class_index_.find(class_key(Q, n * n, R, m * m));
  return it == class_index_.end() ? -1 : it->second;
}

int Batch::add_class(const double* Q, const double* R) {
  const int n = ops_->L.n, m = ops_->L.m;
  const bool sep = ops_->L.layout == LAYOUT_SEPARABLE || ops_->L.layout == LAYOUT_SEPARABLE_PACKED;
  const int words = qr_words(type_, sep);   // te_layout.hpp: full [Q | R], or only the in-group entries (separable layouts)
  const size_t es = elem_size();
  if (n_classes_ == qr_cap_) {   // grow the device table (geometric); recorded graphs hold the old pointer
    const int want = qr_cap_ ? qr_cap_ * 2 : 1;
    void* nt = nullptr;
    TE_HIP_CHECK(hipMalloc(&nt, (size_t)want * words * es));
    if (n_classes_ > 0) {
      TE_HIP_CHECK(hipStreamSynchronize(stream_));
      TE_HIP_CHECK(hipMemcpy(nt, d_qr_, (size_t)n_classes_ * words * es, hipMemcpyDeviceToDevice));
    }
    device_free(d_qr_);
    d_qr_ = nt;
    qr_cap_ = want;
    drop_graphs();
  }
  std::vector<unsigned char> host((size_t)words * es);
  auto put = [&](int w, double v) {
    if (w < 0) return;   // an entry between different axis groups: zero by the separable layout's precondition
    if (dtype_ == F64) reinterpret_cast<double*>(host.data())[w] = v;
    else reinterpret_cast<float*>(host.data())[w] = (float)v;
  };
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) put(qr_q_word(type_, sep, r, c), Q[r * n + c]);
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < m; ++c) put(qr_r_word(type_, sep, r, c), R[r * m + c]);
  TE_HIP_CHECK(hipMemcpy(static_cast<char*>(d_qr_) + (size_t)n_classes_ * words * es, host.data(), host.size(), hipMemcpyHostToDevice));
  class_index_.emplace(class_key(Q, n * n, R, m * m), n_classes_);
  class_qr_.emplace_back((size_t)(n * n + m * m));
  std::copy(Q, Q + n * n, class_qr_.back().begin());
  std::copy(R, R + m * m, class_qr_.back().begin() + n * n);
  if (n_classes_ == 1) drop_graphs();   // the single-class kernels were recorded: from now on the per-class ones run
  return n_classes_++;
}

void Batch::launch_step(const StepParams& p, hipStream_t st, int meas_rows) {
  ops_->step(p, st);
  if (!keep_meas_ || !p.meas || p.n <= 0) return;
  const unsigned blocks = (unsigned)((p.n + 255) / 256);
  if (dtype_ == F64)
    keep_measurement_kernel<double><<<blocks, 256, 0, st>>>(static_cast<const double*>(p.meas), p.meas_ld, p.tick_stride, p.has_meas, p.has_stride,
                                                            p.n_ticks, p.idx, p.n, meas_rows, d_lastmeas_);
  else
    keep_measurement_kernel<float><<<blocks, 256, 0, st>>>(static_cast<const float*>(p.meas), p.meas_ld, p.tick_stride, p.has_meas, p.has_stride,
                                                           p.n_ticks, p.idx, p.n, meas_rows, d_lastmeas_);
}

void Batch::set_keep_measurement(bool on) {
  if (on == keep_meas_) return;
  touch();
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  drop_graphs();   // recorded sequences do (not) contain the row kernels
  if (on) {
    if (cap_ > 0) {
      TE_HIP_CHECK(hipMalloc((void**)&d_lastmeas_, sizeof(double) * 7 * (size_t)cap_));
      init_measured_rows_kernel<<<(unsigned)((cap_ * 7 + 255) / 256), 256, 0, stream_>>>(d_lastmeas_, 0, cap_);
      TE_HIP_CHECK(hipGetLastError());
    }
  } else {
    device_free(d_lastmeas_);
    d_lastmeas_ = nullptr;
  }
  keep_meas_ = on;
}

void Batch::measured_poses(const int* slots, long n, double* out) {
  if (!keep_meas_) throw std::runtime_error("target_estimation_amd: measured poses are not kept (target_manager_set_keep_measurement)");
  flush();
  if (n <= 0) return;
  if (!slots) {
    TE_HIP_CHECK(hipMemcpyAsync(out, d_lastmeas_, sizeof(double) * 7 * (size_t)n, hipMemcpyDeviceToHost, stream_));
  } else {
    for (long i = 0; i < n; ++i)
      TE_HIP_CHECK(hipMemcpyAsync(out + 7 * i, d_lastmeas_ + 7 * (long)slots[i], sizeof(double) * 7, hipMemcpyDeviceToHost, stream_));
  }
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Batch::class_matrices(long slot, double* Q, double* R) {
  const int n = ops_->L.n, m = ops_->L.m;
  int cls = 0;
  if (n_classes_ > 1) {
    flush();
    TE_HIP_CHECK(hipMemcpyAsync(&cls, d_cls_ + slot, sizeof(int), hipMemcpyDeviceToHost, stream_));
    TE_HIP_CHECK(hipStreamSynchronize(stream_));
  }
  const std::vector<double>& qr = class_qr_.at((size_t)cls);
  if (Q) std::copy(qr.begin(), qr.begin() + n * n, Q);
  if (R) std::copy(qr.begin() + n * n, qr.begin() + n * n + m * m, R);
}

int Batch::intern_p0(const double* P0) {
  if (!p0_kept_) return -1;
  const int n = ops_->L.n;
  std::string key(reinterpret_cast<const char*>(P0), sizeof(double) * (size_t)n * n);
  auto it = p0_index_.find(key);
  if (it != p0_index_.end()) return it->second;
  if (p0_tab_.size() >= kP0TableMax) {   // too many distinct initial covariances to mirror on the host: stop keeping them
    p0_kept_ = false;
    p0_tab_.clear(); p0_index_.clear(); slot_p0_.clear();
    return -1;
  }
  p0_tab_.emplace_back(P0, P0 + (size_t)n * n);
  p0_index_.emplace(std::move(key), (int)p0_tab_.size() - 1);
  return (int)p0_tab_.size() - 1;
}

bool Batch::initial_covariance(long slot, double* P0) {
  if (!p0_kept_ || slot < 0 || (size_t)slot >= slot_p0_.size()) return false;
  const std::vector<double>& v = p0_tab_[(size_t)slot_p0_[(size_t)slot]];
  std::copy(v.begin(), v.end(), P0);
  return true;
}

StepParams Batch::base_params() const {
  StepParams p;
  p.rec = d_rec_; p.qr = d_qr_; p.cls = n_classes_ > 1 ? d_cls_ : nullptr; p.n = n_; p.idx = nullptr;
  p.meas = nullptr; p.meas_ld = 0; p.has_meas = nullptr; p.dt_per = nullptr; p.dt = 0.0;
  p.t_base = d_tbase_; p.nm_base = d_nmbase_;
  return p;
}

Batch::~Batch() {
  if (live_.active) { try { live_stop(); } catch (...) {} }
  if (live_.zombie && live_.stream) (void)hipStreamSynchronize(live_.stream);   // (told to stop: it ends as soon as it starts)
  live_release();
  (void)hipStreamSynchronize(stream_);
  drop_graphs();
  if (cap_stream_) (void)hipStreamDestroy(cap_stream_);
  device_free(d_qr_); device_free(d_rec_); device_free(d_rec_alt_); device_free(d_tbase_); device_free(d_nmbase_); device_free(d_cls_);
  device_free(d_idx_); device_free(d_aos_); device_free(d_meas_); device_free(d_mask_); device_free(d_P0_);
  device_free(d_gate_ring_); device_free(d_gate_sum_); device_free(d_gate_state_); device_free(d_gate_prev_);
  device_free(d_dtper_); device_free(d_lastmeas_);
  if (h_pin_) (void)hipHostFree(h_pin_);
  device_free(bar_pin_);
  if (h_cache_) (void)hipHostFree(h_cache_);
  device_free(d_state_scratch_);
  if (h_done_) (void)hipHostFree(h_done_);
  if (d_done_count_) device_free(d_done_count_);
  if (live_.h_block) (void)hipHostFree(live_.h_block);
  device_free(live_.bar_bell);
  device_free(live_.d_block);
  if (live_.stream) (void)hipStreamDestroy(live_.stream);
  if (live_.ready) (void)hipEventDestroy(live_.ready);
}

long Batch::zigzag_min_bytes() {
  static const long v = [] { const char* e = std::getenv("TE_ZIGZAG_MIN_MB"); return (e ? std::atol(e) : 128L) << 20; }();
  return v;
}

long Batch::pingpong_min_bytes() {
  static const long v = [] { const char* e = std::getenv("TE_PINGPONG_MIN_MB"); const long mb = e ? std::atol(e) : 1536L; return mb < 0 ? -1L : mb << 20; }();
  return v;
}

char* Batch::alt_records() {
  if (!d_rec_alt_ && !alt_failed_) {   // same capacity as d_rec_; zero-filled so that the idle lanes of a ragged last tile hold defined words
    const size_t bytes = (size_t)(cap_ / ops_->L.tpw) * (size_t)ops_->L.tile_bytes;
    if (hipMalloc((void**)&d_rec_alt_, bytes) != hipSuccess) {   // no room for a second copy of the state: stay in place (same results)
      (void)hipGetLastError();
      d_rec_alt_ = nullptr;
      alt_failed_ = true;
      return nullptr;
    }
    TE_HIP_CHECK(hipMemsetAsync(d_rec_alt_, 0, bytes, stream_));
  }
  return d_rec_alt_;
}

void Batch::synchronize() {
  flush();
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
}

long Batch::algorithmic_bytes_per_cycle() const {
  const long n = ops_->L.n;
  const bool angular = (type_ == ANGULAR_RATES || type_ == ANGULAR_VELOCITIES);
  // full P: 2n + 2n^2 + 7 (+6); symmetric-packed P: 2n + n(n+1) + 7 (+6)
  long pwords = 2 * n * n;
  if (ops_->L.layout == LAYOUT_PACKED) pwords = n * (n + 1);
  if (ops_->L.layout == LAYOUT_SEPARABLE || ops_->L.layout == LAYOUT_SEPARABLE_PACKED) {
    // only the entries inside an axis group exist (the others are structural zeros): read + write
    pwords = 0;
    for (int r = 0; r < n; ++r)
      for (int c = (ops_->L.layout == LAYOUT_SEPARABLE_PACKED ? r : 0); c < n; ++c)
        pwords += group_of(type_, r) == group_of(type_, c) ? 2 : 0;
  }
  // measurement words the step kernel READS: [x y z] for the linear models, [x y z qx qy qz qw] for the angular
  // ones (kf_step_sep.hpp MW, kf_step.hpp ymeas_own/qmeas); SURVEY 8d's formula charges 7 for every model
  return (2 * n + pwords + (angular ? 7 + 6 : 3)) * (long)elem_size();
}


void Batch::reserve(long n) {
  if (n <= cap_) return;
  const long tpw = ops_->L.tpw;
  long want = std::max(n, cap_ * 2);
  want = (want + tpw - 1) / tpw * tpw;
  const size_t new_bytes = (size_t)(want / tpw) * (size_t)ops_->L.tile_bytes;
  char* rec = nullptr;
  double* tb = nullptr;
  int* nm = nullptr;
  int* cl = nullptr;
  TE_HIP_CHECK(hipMalloc((void**)&rec, new_bytes));
  TE_HIP_CHECK(hipMalloc((void**)&tb, sizeof(double) * want));
  TE_HIP_CHECK(hipMalloc((void**)&nm, sizeof(int) * want));
  TE_HIP_CHECK(hipMalloc((void**)&cl, sizeof(int) * want));
  TE_HIP_CHECK(hipMemsetAsync(rec, 0, new_bytes, stream_));
  TE_HIP_CHECK(hipMemsetAsync(tb, 0, sizeof(double) * want, stream_));
  TE_HIP_CHECK(hipMemsetAsync(nm, 0, sizeof(int) * want, stream_));
  TE_HIP_CHECK(hipMemsetAsync(cl, 0, sizeof(int) * want, stream_));
  if (cap_ > 0) {
    TE_HIP_CHECK(hipMemcpyAsync(rec, d_rec_, (size_t)(cap_ / tpw) * (size_t)ops_->L.tile_bytes, hipMemcpyDeviceToDevice, stream_));
    TE_HIP_CHECK(hipMemcpyAsync(tb, d_tbase_, sizeof(double) * cap_, hipMemcpyDeviceToDevice, stream_));
    TE_HIP_CHECK(hipMemcpyAsync(nm, d_nmbase_, sizeof(int) * cap_, hipMemcpyDeviceToDevice, stream_));
    TE_HIP_CHECK(hipMemcpyAsync(cl, d_cls_, sizeof(int) * cap_, hipMemcpyDeviceToDevice, stream_));
  }
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  device_free(d_rec_); device_free(d_tbase_); device_free(d_nmbase_); device_free(d_cls_);
  device_free(d_rec_alt_); d_rec_alt_ = nullptr; alt_failed_ = false;   // re-created at the new capacity by the next A -> B tick
  if (keep_meas_) {
    double* lm = nullptr;
    TE_HIP_CHECK(hipMalloc((void**)&lm, sizeof(double) * 7 * (size_t)want));
    if (d_lastmeas_ && cap_ > 0) TE_HIP_CHECK(hipMemcpy(lm, d_lastmeas_, sizeof(double) * 7 * (size_t)cap_, hipMemcpyDeviceToDevice));
    device_free(d_lastmeas_);
    d_lastmeas_ = lm;
  }
  d_rec_ = rec; d_tbase_ = tb; d_nmbase_ = nm; d_cls_ = cl; cap_ = want;
  drop_graphs();
}

void Batch::stage_reserve(long n) {
  if (n <= stage_cap_) return;
  const long want = std::max(n, stage_cap_ * 2);
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  device_free(d_idx_); device_free(d_aos_); device_free(d_meas_); device_free(d_mask_); device_free(d_dtper_);
  TE_HIP_CHECK(hipMalloc((void**)&d_dtper_, sizeof(double) * want));
  TE_HIP_CHECK(hipMalloc((void**)&d_idx_, sizeof(int) * want));
  TE_HIP_CHECK(hipMalloc((void**)&d_aos_, sizeof(double) * 19 * want));
  TE_HIP_CHECK(hipMalloc((void**)&d_meas_, elem_size() * 7 * want));
  TE_HIP_CHECK(hipMalloc((void**)&d_mask_, (size_t)want));
  stage_cap_ = want;
}

void Batch::upload_slots(const int* slots, long n) {
  TE_HIP_CHECK(hipMemcpyAsync(d_idx_, slots, sizeof(int) * n, hipMemcpyHostToDevice, stream_));
}

long Batch::append(long count, const unsigned* ids, double t0, const double* P0, bool per_target_P0,
                   const double* p0, const double* v0, const double* a0, int cls, const int* cls_of, const int* P0_index,
                   long P0_count) {
  const int N = ops_->L.n;
  // ONE target (the reference's TargetManager::init, called target by target): queued.  The slot, its id and the host
  // mirrors exist at once; the record is written by the next flush() -- one init launch for a whole run of creations with
  // the same (t0, P0, class) instead of three copies, a launch and a synchronisation each (28 us -> 0.3 us per target).
  if (count == 1 && !per_target_P0 && !cls_of && !P0_index && v0 && a0 && n_ < cap_ && !live_.active) {
    InitQueue& q = initq_;
    if (!q.ids.empty() && (t0 != q.t0 || cls != q.cls || std::memcmp(P0, q.P0.data(), sizeof(double) * (size_t)N * N) != 0)) flush_inits();
    if (q.ids.empty()) { q.t0 = t0; q.cls = cls; q.P0.assign(P0, P0 + (size_t)N * N); q.first = n_; }
    q.ids.push_back(ids[0]);
    q.p0.insert(q.p0.end(), p0, p0 + 7);
    q.v0.insert(q.v0.end(), v0, v0 + 6);
    q.a0.insert(q.a0.end(), a0, a0 + 6);
    cache_valid_ = false;   // the getter table is laid out by n_
    nm_valid_ = false; nm_reads_ = 0;
    if (p0_kept_) {
      slot_p0_.resize((size_t)n_, 0);
      const int r = intern_p0(P0);
      if (p0_kept_) slot_p0_.push_back(r);
    }
    slot_ids_.push_back(ids[0]);
    return n_++;
  }
  touch();
  if (count <= 0) return n_;
  return append_now(n_, count, ids, t0, P0, per_target_P0, p0, v0, a0, cls, cls_of, P0_index, P0_count, true);
}

void Batch::flush_inits() {
  InitQueue& q = initq_;
  if (q.ids.empty()) return;
  InitQueue w;
  w.ids.swap(q.ids); w.p0.swap(q.p0); w.v0.swap(q.v0); w.a0.swap(q.a0); w.P0.swap(q.P0);   // (append_now synchronises: nothing re-enters)
  append_now(q.first, (long)w.ids.size(), w.ids.data(), q.t0, w.P0.data(), false, w.p0.data(), w.v0.data(), w.a0.data(), q.cls, nullptr, nullptr, 0, false);
}

// The device side of append (and, with `bookkeeping`, the host side of a multi-target append): records first .. first + count.
long Batch::append_now(long first, long count, const unsigned* ids, double t0, const double* P0, bool per_target_P0,
                       const double* p0, const double* v0, const double* a0, int cls, const int* cls_of, const int* P0_index,
                       long P0_count, bool bookkeeping) {
  const int N = ops_->L.n;
  reserve(first + count);
  stage_reserve((cls_of || P0_index) ? 3 * count : count);   // slot list (+ per-entry class and P0 indices)
  std::vector<int> slots((size_t)count);
  for (long i = 0; i < count; ++i) slots[(size_t)i] = (int)(first + i);
  upload_slots(slots.data(), count);
  double* d_p0 = d_aos_;
  double* d_v0 = d_aos_ + 7 * count;
  double* d_a0 = d_aos_ + 13 * count;
  TE_HIP_CHECK(hipMemcpyAsync(d_p0, p0, sizeof(double) * 7 * count, hipMemcpyHostToDevice, stream_));
  if (v0) TE_HIP_CHECK(hipMemcpyAsync(d_v0, v0, sizeof(double) * 6 * count, hipMemcpyHostToDevice, stream_));
  if (a0) TE_HIP_CHECK(hipMemcpyAsync(d_a0, a0, sizeof(double) * 6 * count, hipMemcpyHostToDevice, stream_));
  const long p0_words = (P0_index ? P0_count : (per_target_P0 ? count : 1)) * (long)N * N;
  if (p0_words > P0_cap_) {
    TE_HIP_CHECK(hipStreamSynchronize(stream_));
    device_free(d_P0_);
    TE_HIP_CHECK(hipMalloc((void**)&d_P0_, sizeof(double) * p0_words));
    P0_cap_ = p0_words;
  }
  TE_HIP_CHECK(hipMemcpyAsync(d_P0_, P0, sizeof(double) * p0_words, hipMemcpyHostToDevice, stream_));
  InitArgs a;
  a.rec = d_rec_; a.idx = d_idx_; a.n = count; a.p0 = d_p0; a.v0 = v0 ? d_v0 : nullptr; a.a0 = a0 ? d_a0 : nullptr;
  a.P0 = d_P0_; a.per_target_P0 = per_target_P0 ? 1 : 0;
  a.cls = d_cls_; a.cls_value = cls;
  if (cls_of || P0_index) {   // per-entry class / initial-covariance indices: behind the slot list in the index staging
    if (cls_of) { TE_HIP_CHECK(hipMemcpyAsync(d_idx_ + count, cls_of, sizeof(int) * count, hipMemcpyHostToDevice, stream_)); a.cls_of = d_idx_ + count; }
    if (P0_index) { TE_HIP_CHECK(hipMemcpyAsync(d_idx_ + 2 * count, P0_index, sizeof(int) * count, hipMemcpyHostToDevice, stream_)); a.P0_index = d_idx_ + 2 * count; }
  }
  a.t_off = t0 - t_acc_; a.nm_off = (int)(-nm_acc_);
  a.t_base = d_tbase_; a.nm_base = d_nmbase_;
  ops_->init(a, stream_);
  TE_HIP_CHECK(hipGetLastError());
  if (d_gate_ring_) { if (gate_cap_ < cap_) gate_reserve(gate_window_); gate_reset(first, count); }
  if (keep_meas_) {
    init_measured_rows_kernel<<<(unsigned)((count * 7 + 255) / 256), 256, 0, stream_>>>(d_lastmeas_, first, count);
    TE_HIP_CHECK(hipGetLastError());
  }
  if (p0_kept_ && bookkeeping) {   // host mirror of the initial covariances (initial_covariance)
    slot_p0_.resize((size_t)first, 0);
    if (P0_index) {
      std::vector<int> row((size_t)P0_count);
      for (long k = 0; k < P0_count && p0_kept_; ++k) row[(size_t)k] = intern_p0(P0 + (size_t)k * N * N);
      for (long i = 0; i < count && p0_kept_; ++i) slot_p0_.push_back(row[(size_t)P0_index[i]]);
    } else if (per_target_P0) {
      for (long i = 0; i < count && p0_kept_; ++i) { const int r = intern_p0(P0 + (size_t)i * N * N); if (p0_kept_) slot_p0_.push_back(r); }
    } else {
      const int r = intern_p0(P0);
      if (p0_kept_) slot_p0_.insert(slot_p0_.end(), (size_t)count, r);
    }
  }
  TE_HIP_CHECK(hipStreamSynchronize(stream_));  // host staging arrays may be pageable
  if (bookkeeping) {
    slot_ids_.insert(slot_ids_.end(), ids, ids + count);
    n_ += count;
  }
  return first;
}

unsigned Batch::erase_slot(long slot) {
  touch();
  const long last = n_ - 1;
  unsigned moved = slot_ids_[(size_t)slot];
  if (slot != last) {
    ops_->move_record(d_rec_, last, slot, d_tbase_, d_nmbase_, d_cls_, stream_);
    TE_HIP_CHECK(hipGetLastError());
    gate_move(last, slot);
    if (d_lastmeas_) TE_HIP_CHECK(hipMemcpyAsync(d_lastmeas_ + 7 * slot, d_lastmeas_ + 7 * last, sizeof(double) * 7, hipMemcpyDeviceToDevice, stream_));
    if (p0_kept_) slot_p0_[(size_t)slot] = slot_p0_[(size_t)last];
    moved = slot_ids_[(size_t)last];
    slot_ids_[(size_t)slot] = moved;
  }
  if (p0_kept_) slot_p0_.pop_back();
  slot_ids_.pop_back();
  n_ = last;
  return moved;
}

void Batch::erase_slots(const int* slots, long k, std::vector<std::pair<unsigned, int>>& moves_out) {
  touch();
  moves_out.clear();
  if (k <= 0) return;
  const long new_n = n_ - k;
  std::vector<unsigned char> gone((size_t)n_, 0);
  for (long j = 0; j < k; ++j) {
    if (slots[j] < 0 || slots[j] >= n_ || gone[(size_t)slots[j]]) throw std::runtime_error("target_estimation_amd: erase_slots: bad or repeated slot");
    gone[(size_t)slots[j]] = 1;
  }
  // holes below the new size are filled, in order, by the survivors at or above it
  std::vector<int> src, dst;
  long tail = new_n;
  for (long hole = 0; hole < new_n; ++hole) {
    if (!gone[(size_t)hole]) continue;
    while (gone[(size_t)tail]) ++tail;
    src.push_back((int)tail); dst.push_back((int)hole);
    ++tail;
  }
  const long m = (long)src.size();
  if (m > 0) {
    stage_reserve(2 * m);
    TE_HIP_CHECK(hipMemcpyAsync(d_idx_, src.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, stream_));
    TE_HIP_CHECK(hipMemcpyAsync(d_idx_ + m, dst.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, stream_));
    ops_->move_records(d_rec_, d_idx_, d_idx_ + m, m, d_tbase_, d_nmbase_, d_cls_, stream_);
    TE_HIP_CHECK(hipGetLastError());
    for (long j = 0; j < m; ++j) {
      gate_move(src[(size_t)j], dst[(size_t)j]);
      if (d_lastmeas_) TE_HIP_CHECK(hipMemcpyAsync(d_lastmeas_ + 7 * (long)dst[(size_t)j], d_lastmeas_ + 7 * (long)src[(size_t)j], sizeof(double) * 7, hipMemcpyDeviceToDevice, stream_));
      if (p0_kept_) slot_p0_[(size_t)dst[(size_t)j]] = slot_p0_[(size_t)src[(size_t)j]];
      const unsigned id = slot_ids_[(size_t)src[(size_t)j]];
      slot_ids_[(size_t)dst[(size_t)j]] = id;
      moves_out.emplace_back(id, dst[(size_t)j]);
    }
    TE_HIP_CHECK(hipStreamSynchronize(stream_));   // src / dst go out of scope
  }
  slot_ids_.resize((size_t)new_n);
  if (p0_kept_) slot_p0_.resize((size_t)new_n);
  n_ = new_n;
}

void Batch::step_dense(double dt, const void* meas_dev, long ld, const unsigned char* has_dev) {
  touch();
  if (n_ == 0) return;
  StepParams p = base_params();
  p.meas = meas_dev; p.meas_ld = ld; p.has_meas = has_dev; p.dt = dt;
  p.reverse = (flip_ && zigzag()) ? 1 : 0;   // zig-zag: consecutive dense ticks walk the tiles in opposite directions
  flip_ = !flip_;
  if (pingpong()) p.rec_out = alt_records();
  launch_step(p, stream_, host_meas_rows_);
  if (p.rec_out) std::swap(d_rec_, d_rec_alt_);   // later launches on the stream see the finished tick in the new current buffer
  t_acc_ += dt;
  if (meas_dev && !has_dev) nm_acc_ += 1;
}

void Batch::drop_graphs() {
  for (auto& g : graphs_) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
  graphs_.clear();
}

void Batch::step_sequence(long n_ticks, double dt, const void* meas_base, long tick_stride, long ld,
                          const unsigned char* has_base, long has_stride, int use_graph, long ring_ticks) {
  touch();
  if (n_ == 0 || n_ticks <= 0) return;
  const size_t es = elem_size();
  auto params = [&](long s0) {
    const long s = ring_ticks > 0 ? s0 % ring_ticks : s0;   // the measurements form a ring of ring_ticks ticks
    StepParams p = base_params();
    p.meas = meas_base ? static_cast<const char*>(meas_base) + (size_t)(s * tick_stride) * es : nullptr;
    p.meas_ld = ld;
    p.has_meas = has_base ? has_base + s * has_stride : nullptr;
    p.dt = dt;
    p.reverse = zigzag() ? (int)((s0 + (use_graph ? 0 : (flip_ ? 1 : 0))) & 1) : 0;   // zig-zag; a recorded graph starts forwards
    return p;
  };
  if (!use_graph) {
    const bool ab = pingpong();
    for (long s = 0; s < n_ticks; ++s) {
      StepParams p = params(s);
      if (ab) p.rec_out = alt_records();
      launch_step(p, stream_);
      if (p.rec_out) std::swap(d_rec_, d_rec_alt_);
    }
    TE_HIP_CHECK(hipGetLastError());
    if (n_ticks & 1) flip_ = !flip_;
  } else {
    GraphEntry* hit = nullptr;
    for (auto& g : graphs_)
      if (g.n_ticks == n_ticks && g.tick_stride == tick_stride && g.ld == ld && g.has_stride == has_stride && g.n == n_ &&
          g.dt == dt && g.meas_base == meas_base && g.has_base == has_base && g.rec == d_rec_ && g.ring_ticks == ring_ticks) hit = &g;
    if (!hit) {
      if (graphs_.size() >= 64) {   // e.g. a ring of 4096 ticks replayed in 64-tick blocks: evict the least recently used one
        size_t victim = 0;
        for (size_t k = 1; k < graphs_.size(); ++k)
          if (graphs_[k].last_use < graphs_[victim].last_use) victim = k;
        TE_HIP_CHECK(hipStreamSynchronize(stream_));   // it may still be running
        (void)hipGraphExecDestroy(graphs_[victim].exec);
        (void)hipGraphDestroy(graphs_[victim].graph);
        graphs_.erase(graphs_.begin() + (long)victim);
      }
      if (!cap_stream_) TE_HIP_CHECK(hipStreamCreateWithFlags(&cap_stream_, hipStreamNonBlocking));
      GraphEntry e{n_ticks, tick_stride, ld, has_stride, n_, dt, meas_base, has_base, d_rec_, nullptr, nullptr, ring_ticks};
      TE_HIP_CHECK(hipStreamBeginCapture(cap_stream_, hipStreamCaptureModeThreadLocal));
      try {
        for (long s = 0; s < n_ticks; ++s) {
          launch_step(params(s), cap_stream_);
#ifdef TE_TEST_HOOKS   // only in libtarget_estimation_amd_testhooks.so (csrc/Makefile `testhooks`)
          if (s == 0 && std::getenv("TE_TEST_FAIL_IN_CAPTURE"))   // test hook: a launch that throws between Begin and EndCapture
            throw std::runtime_error("target_estimation_amd: injected failure inside stream capture");
#endif
        }
        TE_HIP_CHECK(hipGetLastError());
      } catch (...) {
        hipGraph_t broken = nullptr;
        (void)hipStreamEndCapture(cap_stream_, &broken);   // never leave the stream in capture mode
        if (broken) (void)hipGraphDestroy(broken);
        throw;
      }
      TE_HIP_CHECK(hipStreamEndCapture(cap_stream_, &e.graph));
      if (hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(e.graph);
        throw std::runtime_error("target_estimation_amd: hipGraphInstantiate failed for a step sequence");
      }
      graphs_.push_back(e);
      hit = &graphs_.back();
    }
    hit->last_use = ++graph_clock_;
    if (use_graph == 2) return;  // record only
    TE_HIP_CHECK(hipGraphLaunch(hit->exec, stream_));
    flip_ = (n_ticks & 1) != 0;   // the graph's last tick ran forwards (odd count) or backwards
  }
  t_acc_ += dt * (double)n_ticks;
  if (meas_base && !has_base) nm_acc_ += n_ticks;
}

void Batch::enqueue_tick(hipStream_t st, long s, double dt, const SeqSpec& q, bool query, const double* origin, double radius,
                         bool reverse, bool ab) {
  if (n_ == 0) return;
  if (q.ring_ticks > 0) s %= q.ring_ticks;
  const size_t es = elem_size();
  const bool fused_q = ops_->fused_query && n_classes_ == 1;
  StepParams p = base_params();
  p.meas = q.meas_base ? static_cast<const char*>(q.meas_base) + (size_t)(s * q.tick_stride) * es : nullptr;
  p.meas_ld = q.ld;
  p.has_meas = q.has_base ? q.has_base + s * q.has_stride : nullptr;
  p.dt = dt;
  p.reverse = reverse ? 1 : 0;
  if (query && fused_q) {
    p.q_origin[0] = origin[0]; p.q_origin[1] = origin[1]; p.q_origin[2] = origin[2];
    p.q_radius = radius; p.q_delta = q.delta_dev; p.q_pose = q.pose_dev;
  }
  if (ab && !p.q_delta) p.rec_out = alt_records();   // (the fused query's kernels have no A -> B form: in place)
  launch_step(p, st);
  if (p.rec_out) std::swap(d_rec_, d_rec_alt_);
  if (query && !fused_q) {
    IntersectArgs a;
    a.rec = d_rec_; a.idx = nullptr; a.n = n_; a.t1 = std::numeric_limits<double>::quiet_NaN();
    a.origin[0] = origin[0]; a.origin[1] = origin[1]; a.origin[2] = origin[2]; a.radius = radius;
    a.t_acc = 0.0; a.t_base = d_tbase_; a.delta = q.delta_dev; a.pose = q.pose_dev;
    ops_->intersect(a, st);
  }
}

bool Batch::population_ready() const {
  return ops_->L.layout == LAYOUT_SEPARABLE_PACKED && n_classes_ == 1 && !keep_meas_ && ops_->fused_query;
}

StepParams Batch::tick_params(long s, double dt, const SeqSpec& q, bool query, const double* origin, double radius, bool ab) {
  if (q.ring_ticks > 0) s %= q.ring_ticks;
  const size_t es = elem_size();
  StepParams p = base_params();
  p.meas = q.meas_base ? static_cast<const char*>(q.meas_base) + (size_t)(s * q.tick_stride) * es : nullptr;
  p.meas_ld = q.ld;
  p.has_meas = q.has_base ? q.has_base + s * q.has_stride : nullptr;
  p.dt = dt;
  if (query) {
    p.q_origin[0] = origin[0]; p.q_origin[1] = origin[1]; p.q_origin[2] = origin[2];
    p.q_radius = radius; p.q_delta = q.delta_dev; p.q_pose = q.pose_dev;
  }
  if (ab && !query) p.rec_out = alt_records();
  return p;
}

void Batch::account_sequence(long n_ticks, double dt, bool all_measured) {
  t_acc_ += dt * (double)n_ticks;
  if (all_measured) nm_acc_ += n_ticks;
}

void Batch::step_fused(long n_ticks, double dt, const void* meas_base, long tick_stride, long ld,
                       const unsigned char* has_base, long has_stride) {
  touch();
  if (n_ == 0 || n_ticks <= 0) return;
  StepParams p = base_params();
  p.meas = meas_base; p.meas_ld = ld; p.has_meas = has_base; p.dt = dt;
  p.n_ticks = (int)n_ticks; p.tick_stride = tick_stride; p.has_stride = has_stride;
  launch_step(p, stream_);
  TE_HIP_CHECK(hipGetLastError());
  t_acc_ += dt * (double)n_ticks;
  if (meas_base && !has_base) nm_acc_ += n_ticks;
}

// device block of a live session: [one mirror word per kLiveGroup wavefronts, 128 bytes apart][progress words]
static size_t live_mirror_bytes(long waves) { return (size_t)((waves + kLiveGroup - 1) / kLiveGroup) * kLiveMirrorStride * sizeof(long long); }
static long live_progress_words(long waves) { return (waves + kLiveScan - 1) / kLiveScan * kLiveScan; }   // whole relay scans
static size_t live_block_bytes(long waves) { return (live_mirror_bytes(waves) + sizeof(int) * (size_t)live_progress_words(waves) + 15) / 16 * 16; }

void Batch::live_start(double dt, const void* meas_ring, long tick_stride, long ld, const unsigned char* has_ring, long has_stride,
                       long ring_ticks, long first_entry, long max_ticks, double idle_limit_s, const double* q_origin, double q_radius,
                       double* q_delta_dev, double* q_pose_dev) {
  touch();   // ends a previous session, runs queued one-target steps
  if (q_delta_dev && !q_origin) throw std::invalid_argument("target_estimation_amd: live_start: query without an origin");
  if (n_ == 0) throw std::runtime_error("target_estimation_amd: live mode on an empty batch");
  if (!meas_ring || ring_ticks <= 0 || max_ticks <= 0 || first_entry < 0 || ld < n_ || tick_stride < 7 * ld || (has_ring && has_stride < n_))
    throw std::invalid_argument("target_estimation_amd: live_start: bad measurement ring");
  if (max_ticks > 0x7fffffffL) throw std::invalid_argument("target_estimation_amd: live_start: at most 2^31 - 1 ticks per session");
  if (n_classes_ > 1) throw std::runtime_error("target_estimation_amd: live mode serves batches with one (Q, R) class");
  long cap = ops_->live_capacity ? ops_->live_capacity((q_delta_dev || live_.pose_out) ? 1 : 0) : 0;
  if (const char* e = std::getenv("TE_LIVE_CAPACITY_WAVES")) { if (cap > 0 && std::atol(e) > 0) cap = std::atol(e); }   // experiments only (tools/live_capacity.py --probe)
  if (cap <= 0) throw std::runtime_error("target_estimation_amd: live mode needs the axis-separable layout with packed groups");
  const long waves = (n_ + ops_->L.tpw - 1) / ops_->L.tpw;
  if (waves > cap)
    throw std::runtime_error("target_estimation_amd: live mode: " + std::to_string(n_) + " targets need " + std::to_string(waves) +
                             " resident wavefronts, the device holds " + std::to_string(cap) + " of this kernel");
  if (waves + 1 > cap) throw std::runtime_error("target_estimation_amd: live mode: no room for the relay wavefront");
  // ... and next to the sessions that are resident already (other batches, other managers of this process): their grids hold
  // their wave slots until they end, so a grid that only fits an empty device would start in part and never get its relay
  const double share = (double)(waves + 1) / (double)cap;
  if (!live_session_begin(share))
    throw std::runtime_error("target_estimation_amd: live mode: " + std::to_string(n_) + " targets take " + std::to_string((int)(share * 100.0 + 0.5)) +
                             " % of the device's resident wavefronts and the sessions already resident in this process take " +
                             std::to_string((int)(live_sessions_share() * 100.0 + 0.5)) + " %: they do not fit the device together");
  live_.share = share;   // counted from here on; every way out of the session gives it back (live_release)
  try {
  if (!live_.h_block) {
    char* h = nullptr;
    TE_HIP_CHECK(hipHostMalloc((void**)&h, 128, hipHostMallocMapped | hipHostMallocCoherent));
    char* d = nullptr;
    TE_HIP_CHECK(hipHostGetDevicePointer((void**)&d, h, 0));
    live_.h_block = h;
    live_.h_posted = reinterpret_cast<long long*>(h); live_.d_posted = reinterpret_cast<long long*>(d);
    live_.h_done = reinterpret_cast<int*>(h + 64); live_.d_done = reinterpret_cast<int*>(d + 64);
    // The doorbell itself lives in DEVICE memory where the host can write it through the PCIe BAR (large-BAR systems, fine-grained
    // device memory: the device pointer is valid on the host): the relay then polls a local word instead of reading host memory
    // over PCIe every round -- the direction that costs a round trip (tools/bar_doorbell.hip: 2.66 -> 2.21 us per echo).  The
    // completion word stays in host memory: the GPU writes it, the host polls its own memory.  TE_LIVE_DOORBELL=host keeps both there.
    const char* e = std::getenv("TE_LIVE_DOORBELL");
    int dev = 0, large_bar = 0;
    if (!(e && e[0] == 'h') && hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && large_bar) {
      void* bell = nullptr;
      if (hipExtMallocWithFlags(&bell, 128, hipDeviceMallocFinegrained) == hipSuccess) {
        live_.bar_bell = static_cast<long long*>(bell);
        live_.h_posted = live_.d_posted = live_.bar_bell;
      } else {
        (void)hipGetLastError();
      }
    }
  }
  if (waves > live_.cap_waves) {
    TE_HIP_CHECK(hipStreamSynchronize(stream_));
    device_free(live_.d_block);
    live_.d_block = nullptr;
    TE_HIP_CHECK(hipMalloc((void**)&live_.d_block, live_block_bytes(waves)));
    live_.cap_waves = waves;
  }
  __atomic_store_n(live_.h_posted, 0LL, __ATOMIC_RELAXED);
  __atomic_store_n(live_.h_done, -1, __ATOMIC_RELAXED);   // the relay's first store makes it 0: "running"
  __atomic_store_n(live_.h_done + 2, 0, __ATOMIC_RELAXED);    // the relay's last store makes it 1: "ended"
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  // The resident kernel gets a stream of its own (non-blocking), ordered behind everything already queued on the batch's
  // stream: nothing the caller queues later on that stream -- for this batch's siblings in the manager, say -- waits for the
  // session, and two batches of one manager can be resident together.
  if (!live_.stream) {
    // A stream of its own PRIORITY class: the runtime multiplexes the streams of one priority over a few hardware queues, and
    // a stream that lands in the resident kernel's queue waits until the session ends (a ring refill on "another stream" took
    // 1.6 - 17 s, i.e. the idle limit, whenever that happened; with GPU_MAX_HW_QUEUES=1 always).  Queues are not shared across
    // priorities, so the resident kernel gets the high-priority class to itself (callers' streams are normal priority unless
    // they ask otherwise).
    int least = 0, greatest = 0;
    TE_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (const char* e = std::getenv("TE_LIVE_STREAM_PRIORITY")) {   // experiments only (tools/r4_regress.sh): 0 = normal, low = least
      if (e[0] == 'l') greatest = least; else if (std::atoi(e) == 0) greatest = 0;
    }
    TE_HIP_CHECK(hipStreamCreateWithPriority(&live_.stream, hipStreamNonBlocking, greatest));
  }
  if (!live_.ready) TE_HIP_CHECK(hipEventCreateWithFlags(&live_.ready, hipEventDisableTiming));
  TE_HIP_CHECK(hipEventRecord(live_.ready, stream_));
  TE_HIP_CHECK(hipStreamWaitEvent(live_.stream, live_.ready, 0));
  TE_HIP_CHECK(hipMemsetAsync(live_.d_block, 0, live_block_bytes(waves), live_.stream));   // mirrors + progress: zero before EVERY launch
  if (live_progress_words(waves) > waves)   // the padding of the progress words: INT_MAX, neutral in the relay's minimum
    TE_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(live_.d_block + live_mirror_bytes(waves) + sizeof(int) * (size_t)waves), 0x7fffffff,
                                   (size_t)(live_progress_words(waves) - waves), live_.stream));
  StepParams p = base_params();
  p.meas = meas_ring; p.meas_ld = ld; p.has_meas = has_ring; p.dt = dt;
  p.n_ticks = (int)max_ticks; p.tick_stride = tick_stride; p.has_stride = has_stride;
  p.live_posted = live_.d_posted; p.live_done = live_.d_done;
  p.live_mirror = reinterpret_cast<long long*>(live_.d_block); p.live_progress = reinterpret_cast<int*>(live_.d_block + live_mirror_bytes(waves));
  p.live_ring = ring_ticks; p.live_first = first_entry % ring_ticks;
  // a poll is a PCIe round trip plus s_sleep: ~1-2 us; the limit is a count of polls
  const double polls = idle_limit_s > 0 ? idle_limit_s * 5e5 : 5e6;
  p.live_spin_limit = (unsigned)std::min(polls, 4.0e9);
  {   // the relay measures the idle time on the device's constant-rate clock; the count above bounds the workers' own spins
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
      p.live_idle_ticks = (unsigned long long)((idle_limit_s > 0 ? idle_limit_s : 10.0) * 1e3 * (double)khz);
  }
  { const char* e = std::getenv("TE_LIVE_FLAGS"); p.live_flags = e ? std::atoi(e) : 0; }
  if (live_.pose_out) {
    if (live_.pose_ld < n_) throw std::invalid_argument("target_estimation_amd: live pose output: rows shorter than the batch (it grew since live_set_pose_output)");
    p.live_pose = live_.pose_out; p.live_pose_ld = live_.pose_ld;
  }
  if (q_delta_dev) {
    p.q_origin[0] = q_origin[0]; p.q_origin[1] = q_origin[1]; p.q_origin[2] = q_origin[2];
    p.q_radius = q_radius; p.q_delta = q_delta_dev; p.q_pose = q_pose_dev;
  }
  ops_->step(p, live_.stream);   // (the measured-pose rows are not kept during a live session: see measured_pose.hpp)
  TE_HIP_CHECK(hipGetLastError());
  live_.waves = waves; live_.posted = 0; live_.max_ticks = max_ticks; live_.dt = dt;
  live_.all_measured = has_ring == nullptr;
  // The session exists once its LAST workgroup (the relay) says so: then every worker holds its wave slot.  A kernel that is
  // still queued after two seconds is not going to run next to whatever is ahead of it (its stream shares a hardware queue with
  // another endless kernel -- more live batches than GPU_MAX_HW_QUEUES, or a caller's own high-priority stream -- or the device
  // cannot hold the whole grid): told to stop as soon as it starts, and reported, instead of a session that never serves a tick.
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(kLiveStartTimeoutS);
  for (unsigned spins = 0; __atomic_load_n(live_.h_done, __ATOMIC_ACQUIRE) < 0; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 1023u) != 1023u) continue;
    const hipError_t q = hipStreamQuery(live_.stream);
    if (q != hipErrorNotReady && q != hipSuccess) TE_HIP_CHECK(q);
    if (std::chrono::steady_clock::now() < t_end) continue;
    __atomic_store_n(live_.h_posted, kLiveStop, __ATOMIC_RELEASE);
    if (live_.bar_bell) __builtin_ia32_sfence();
    // (workers that did get a wave slot look at the host's word themselves every millisecond or two: they leave, the rest of the
    // grid gets their slots, sees the stop and leaves too; the records are back as they were)
    const auto t_gone = std::chrono::steady_clock::now() + std::chrono::duration<double>(1.0);
    while (hipStreamQuery(live_.stream) == hipErrorNotReady && std::chrono::steady_clock::now() < t_gone) __builtin_ia32_pause();
    live_.zombie = hipStreamQuery(live_.stream) == hipErrorNotReady;   // still queued behind somebody else's kernel: flush() waits for it before anything touches the records
    if (!live_.zombie) live_release();
    throw std::runtime_error("target_estimation_amd: live mode: the resident kernel did not start within " + std::to_string(kLiveStartTimeoutS) +
                             " s (its stream shares a hardware queue with another endless kernel -- more live batches than GPU_MAX_HW_QUEUES, or a "
                             "high-priority stream of the caller -- or the device is busy)");
  }
  } catch (...) {
    if (!live_.zombie) live_release();   // (a zombie still holds its place: flush() releases it once the kernel is gone)
    throw;
  }
  live_.active = true;
}

void Batch::live_release() {
  if (live_.share > 0.0) { const double s = live_.share; live_.share = 0.0; live_session_ended(s); }
}

void Batch::live_set_pose_output(double* pose_soa_dev, long ld) {
  if (pose_soa_dev && ld < n_) throw std::invalid_argument("target_estimation_amd: live pose output: rows shorter than the batch");
  live_.pose_out = pose_soa_dev;
  live_.pose_ld = pose_soa_dev ? ld : 0;
}

void Batch::live_post(long n_ticks) {
  if (!live_.active) throw std::runtime_error("target_estimation_amd: live_post without a live session");
  if (n_ticks <= 0) return;
  if (live_.posted + n_ticks > live_.max_ticks) throw std::runtime_error("target_estimation_amd: live_post beyond the session's max_ticks");
  live_.posted += n_ticks;
  __atomic_store_n(live_.h_posted, (long long)live_.posted, __ATOMIC_RELEASE);   // the ring entries were written before this call
  if (live_.bar_bell) __builtin_ia32_sfence();   // (device memory behind the BAR is write-combined on the host: push the store out now)
}

long Batch::live_done() const {
  if (!live_.active) return 0;
  return std::max(0L, (long)__atomic_load_n(live_.h_done, __ATOMIC_ACQUIRE));
}

bool Batch::live_wait(long tick, double timeout_s) const {
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
  for (unsigned spins = 0;; ++spins) {
    if (live_done() >= tick) return true;
    __builtin_ia32_pause();
    if ((spins & 255u) == 255u && std::chrono::steady_clock::now() >= t_end) return false;
  }
}

long Batch::live_stop() {
  if (!live_.active) return 0;
  __atomic_store_n(live_.h_posted, (long long)live_.posted | kLiveStop, __ATOMIC_RELEASE);
  if (live_.bar_bell) __builtin_ia32_sfence();
  live_.active = false;     // whatever happens below, the session is over (flush() must not come back here)
  {
    const hipError_t e = hipStreamSynchronize(live_.stream);   // bounded: every wavefront drains the posted ticks, then sees the stop bit
    live_release();                                          // the kernel is gone: its wave slots, and the frees that waited for it
    TE_HIP_CHECK(e);
  }
  // the relay's last word: the ticks EVERY wavefront served.  The host's stop, or the relay's own after a silent host, reaches
  // all workers through one device word, so they all stop at the same tick.
  const long mn = std::max(0L, (long)__atomic_load_n(live_.h_done, __ATOMIC_ACQUIRE));
  t_acc_ += live_.dt * (double)mn;
  if (live_.all_measured) nm_acc_ += mn;
  if (mn != live_.posted)
    throw std::runtime_error("target_estimation_amd: live session ended after " + std::to_string(mn) + " of " + std::to_string(live_.posted) +
                             " posted ticks (the idle limit stopped it before the last post?)");
  return (long)mn;
}

void Batch::step_indexed(const int* slots, long n, double dt, const double* meas_aos, const unsigned char* has) {
  touch();
  if (n <= 0) return;
  stage_reserve(n);
  upload_slots(slots, n);
  if (meas_aos) {
    TE_HIP_CHECK(hipMemcpyAsync(d_aos_, meas_aos, sizeof(double) * 7 * n, hipMemcpyHostToDevice, stream_));
    ops_->pack_meas(d_aos_, n, d_meas_, n, stream_);
  }
  if (has) TE_HIP_CHECK(hipMemcpyAsync(d_mask_, has, (size_t)n, hipMemcpyHostToDevice, stream_));
  StepParams p = base_params();
  p.n = n; p.idx = d_idx_; p.meas = meas_aos ? d_meas_ : nullptr; p.meas_ld = n;
  p.has_meas = (meas_aos && has) ? d_mask_ : nullptr; p.dt = dt;
  launch_step(p, stream_);
  TE_HIP_CHECK(hipGetLastError());
  TE_HIP_CHECK(hipStreamSynchronize(stream_));  // caller's host arrays may be reused after return
}

void Batch::step_indexed_dev(const int* idx_dev, long n, double dt, const void* meas_soa_dev, long ld, const unsigned char* has_dev) {
  touch();
  if (n <= 0) return;
  StepParams p = base_params();
  p.n = n; p.idx = idx_dev; p.meas = meas_soa_dev; p.meas_ld = ld;
  p.has_meas = meas_soa_dev ? has_dev : nullptr; p.dt = dt;
  launch_step(p, stream_);
  TE_HIP_CHECK(hipGetLastError());
}

void Batch::outputs_indexed_dev(const int* idx_dev, long n, double* pose_dev, double* twist_dev, double* acc_dev, bool at_time, double t1) {
  flush();
  if (n <= 0) return;
  OutArgs a;
  a.rec = d_rec_; a.idx = idx_dev; a.n = n; a.pose = pose_dev; a.twist = twist_dev; a.acc = acc_dev;
  a.at_time = at_time ? 1 : 0; a.t1 = t1; a.t_acc = t_acc_; a.t_base = d_tbase_;
  ops_->outputs(a, stream_);
  TE_HIP_CHECK(hipGetLastError());
}

void Batch::step_dense_host(double dt, const double* meas_aos, const unsigned char* has) {
  touch();   // before the staging buffers are filled: a pending flush uses them too
  if (n_ == 0) return;
  stage_reserve(n_);
  if (meas_aos) {
    TE_HIP_CHECK(hipMemcpyAsync(d_aos_, meas_aos, sizeof(double) * 7 * n_, hipMemcpyHostToDevice, stream_));
    ops_->pack_meas(d_aos_, n_, d_meas_, n_, stream_);
  }
  if (has) TE_HIP_CHECK(hipMemcpyAsync(d_mask_, has, (size_t)n_, hipMemcpyHostToDevice, stream_));
  step_dense(dt, meas_aos ? d_meas_ : nullptr, n_, (meas_aos && has) ? d_mask_ : nullptr);
  TE_HIP_CHECK(hipGetLastError());
  TE_HIP_CHECK(hipStreamSynchronize(stream_));  // caller's host arrays may be reused after return
}

void Batch::step_dense_host_soa(double dt, const void* meas_soa, long ld_host, const unsigned char* has) {
  touch();
  if (n_ == 0) return;
  if (meas_soa && ld_host < n_) throw std::invalid_argument("target_estimation_amd: step_host: the row stride of the host measurements is smaller than the batch");
  stage_reserve(n_);
  if (meas_soa) {
    const bool angular = (type_ == ANGULAR_RATES || type_ == ANGULAR_VELOCITIES);
    const size_t es = elem_size();
    // rows of the device staging block are stage_cap_ elements apart
    TE_HIP_CHECK(hipMemcpy2DAsync(d_meas_, (size_t)stage_cap_ * es, meas_soa, (size_t)ld_host * es, (size_t)n_ * es,
                                  angular ? 7 : 3, hipMemcpyHostToDevice, stream_));
  }
  if (has) TE_HIP_CHECK(hipMemcpyAsync(d_mask_, has, (size_t)n_, hipMemcpyHostToDevice, stream_));
  host_meas_rows_ = (type_ == ANGULAR_RATES || type_ == ANGULAR_VELOCITIES) ? 7 : 3;   // what was transported (measured_pose.hpp)
  step_dense(dt, meas_soa ? d_meas_ : nullptr, stage_cap_, (meas_soa && has) ? d_mask_ : nullptr);
  host_meas_rows_ = 7;
  TE_HIP_CHECK(hipGetLastError());
  TE_HIP_CHECK(hipStreamSynchronize(stream_));  // caller's host arrays may be reused after return
}

void Batch::step_one(long slot, double dt, const double* meas7) {
  if (pending_mark_.size() < (size_t)n_) pending_mark_.resize((size_t)n_, 0);
  if (pending_mark_[(size_t)slot]) flush();   // second step of the same target: keep the caller's order
  Pending p;
  p.slot = (int)slot; p.has = meas7 ? 1 : 0; p.dt = dt;
  for (int c = 0; c < 7; ++c) p.meas[c] = meas7 ? meas7[c] : 0.0;
  pending_.push_back(p);
  pending_mark_[(size_t)slot] = 1;
}

void Batch::pin_reserve(long k) {
  if (k <= pin_cap_) return;
  const long want = (std::max<long>(std::max<long>(k, 64), pin_cap_ * 2) + 15) / 16 * 16;   // keeps the sections aligned
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  if (h_pin_) (void)hipHostFree(h_pin_);
  h_pin_ = nullptr; d_pin_ = nullptr; pin_cap_ = 0;
  const size_t bytes = (size_t)want * (sizeof(int) + sizeof(double) + 7 * elem_size() + 1) + 64;
  TE_HIP_CHECK(hipHostMalloc((void**)&h_pin_, bytes, hipHostMallocMapped | hipHostMallocCoherent));
  TE_HIP_CHECK(hipHostGetDevicePointer((void**)&d_pin_, h_pin_, 0));
  pin_cap_ = want;
}

void Batch::cache_reserve(long n) {
  if (n <= cache_cap_) return;
  const long want = std::max<long>(std::max<long>(n, 64), std::min<long>(cache_cap_ * 2, n > kCacheMax ? kCacheBigMax : kCacheMax));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  if (h_cache_) (void)hipHostFree(h_cache_);
  h_cache_ = nullptr; d_cache_ = nullptr; cache_cap_ = 0;
  TE_HIP_CHECK(hipHostMalloc((void**)&h_cache_, sizeof(double) * 19 * (size_t)want, hipHostMallocMapped | hipHostMallocCoherent));
  TE_HIP_CHECK(hipHostGetDevicePointer((void**)&d_cache_, h_cache_, 0));
  cache_cap_ = want;
  cache_valid_ = false;
}

// TE_SPIN_WAIT=0 keeps the stream synchronisation (e.g. to leave the core to other threads)
static bool spin_wait_enabled() {
  static const bool on = [] { const char* e = std::getenv("TE_SPIN_WAIT"); return !(e && e[0] == '0'); }();
  return on;
}

void Batch::flush() {
  if (live_.zombie) {              // a resident kernel that never started in time: told to stop, it leaves the records as they are
    live_.zombie = false;
    const hipError_t e = hipStreamSynchronize(live_.stream);
    live_release();
    TE_HIP_CHECK(e);
  }
  if (live_.active) live_stop();   // the records in HBM are stale while a live kernel holds the state
  flush_inits();                   // queued creations first: a queued step may be for one of them
  const long k = (long)pending_.size();
  if (!k) return;
  nm_valid_ = false; nm_reads_ = 0;   // the stepped targets' counters change
  // The queue is written by the host and read by the kernel.  A flush of up to one wavefront of targets -- the latency path: the
  // reference's own loop steps one target and reads it back -- goes through a small block in fine-grained DEVICE memory that the host
  // writes through the PCIe BAR (large-BAR systems): posted writes, and the kernel's reads are local instead of four PCIe read round
  // trips at its head (the reference's loop 10.3 -> 8.5 us per cycle for uniform_velocity, 14.3 -> 12.3 for the EKF,
  // profiles/r04_bar_queue.txt; 90 tests and the one-target fuzz run through it: the block is reused by every flush and never read
  // stale).  Larger flushes keep the host-mapped block: the kernel's bulk reads pipeline over PCIe and cached host stores are
  // faster than write-combined ones (400 targets: 41 against 45 us).  TE_QUEUE_BAR=0: always host-mapped.
  constexpr long kBarQueue = 64;
  static const bool bar_wanted = [] { const char* e = std::getenv("TE_QUEUE_BAR"); return !(e && e[0] == '0'); }();
  const size_t es = elem_size();
  if (bar_wanted && !bar_pin_ && !bar_pin_failed_ && k <= kBarQueue) {
    int dev = 0, large_bar = 0;
    void* q = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && large_bar &&
        hipExtMallocWithFlags(&q, (size_t)kBarQueue * (sizeof(int) + sizeof(double) + 7 * sizeof(double) + 1) + 64, hipDeviceMallocFinegrained) == hipSuccess) {
      bar_pin_ = static_cast<char*>(q);
    } else {
      (void)hipGetLastError();
      bar_pin_failed_ = true;
    }
  }
  const bool use_bar = bar_wanted && bar_pin_ && k <= kBarQueue;
  if (!use_bar) pin_reserve(k);
  char* const hq = use_bar ? bar_pin_ : h_pin_;   // as the host writes it (never reads it: the BAR mapping is write-combined)
  char* const dq = use_bar ? bar_pin_ : d_pin_;   // as the kernel reads it
  const long qcap = use_bar ? kBarQueue : pin_cap_;
  // sections of the block (sized by its capacity, so that the offsets are 8-byte aligned)
  const size_t off_dt = (size_t)qcap * sizeof(int), off_meas = off_dt + (size_t)qcap * sizeof(double),
               off_has = off_meas + (size_t)qcap * 7 * es;
  int* h_idx = reinterpret_cast<int*>(hq);
  double* h_dt = reinterpret_cast<double*>(hq + off_dt);
  unsigned char* h_has = reinterpret_cast<unsigned char*>(hq + off_has);
  bool all_has = true, any_has = false;
  for (long j = 0; j < k; ++j) {
    const Pending& p = pending_[(size_t)j];
    h_idx[j] = p.slot; h_dt[j] = p.dt; h_has[j] = p.has;
    // SoA [7][k] in the batch precision (what pack_meas produces on the device)
    if (dtype_ == F64) { double* m = reinterpret_cast<double*>(hq + off_meas); for (int c = 0; c < 7; ++c) m[(size_t)c * k + j] = p.meas[c]; }
    else { float* m = reinterpret_cast<float*>(hq + off_meas); for (int c = 0; c < 7; ++c) m[(size_t)c * k + j] = (float)p.meas[c]; }
    all_has = all_has && p.has;
    any_has = any_has || p.has;
    pending_mark_[(size_t)p.slot] = 0;
  }
  pending_.clear();
  if (use_bar) __builtin_ia32_sfence();   // write-combined on the host: out before the launch
  StepParams p = base_params();
  p.n = k; p.idx = reinterpret_cast<const int*>(dq);
  p.meas = any_has ? dq + off_meas : nullptr; p.meas_ld = k;
  p.has_meas = (any_has && !all_has) ? reinterpret_cast<const unsigned char*>(dq + off_has) : nullptr;
  p.dt_per = reinterpret_cast<const double*>(dq + off_dt); p.dt = 0.0;
  // With a current getter table the flush reports its own completion through a flag in host-mapped memory and the host
  // spins on it instead of synchronising the stream (tools/launch_latency.hip: the runtime's completion path costs 4 us more
  // than a PCIe write).  Up to one wavefront of queued targets the step kernel itself writes the table rows and the flag --
  // ONE launch per flush; up to one workgroup of the outputs kernel that kernel does.
  // Round 4: beyond one wavefront of targets (up to kFusedFlushMax) the step kernel still does it all -- its wavefronts count
  // themselves in and the last one writes the flag (signal_done) -- instead of a second launch and, beyond one workgroup of
  // the outputs kernel, the runtime's wait (from C, 65 targets: 16.4 -> 12.x us per update + read-back tick; TE_FUSED_FLUSH_MAX).
  static const long fused_max = [] { const char* e = std::getenv("TE_FUSED_FLUSH_MAX"); return e && *e ? std::atol(e) : kFusedFlushMax; }();
  int seq = 0;
  if (cache_valid_ && spin_wait_enabled() && k <= std::max<long>(std::max<long>(ops_->L.tpw, kOutputsBlock), fused_max)) {
    if (!h_done_) {
      TE_HIP_CHECK(hipHostMalloc((void**)&h_done_, 64, hipHostMallocMapped | hipHostMallocCoherent));
      TE_HIP_CHECK(hipHostGetDevicePointer((void**)&d_done_, h_done_, 0));
      *h_done_ = 0;
      TE_HIP_CHECK(hipMalloc((void**)&d_done_count_, 64));   // the wavefront counter of multi-wavefront flushes: zero between launches
      TE_HIP_CHECK(hipMemsetAsync(d_done_count_, 0, 64, stream_));
    }
    seq = (int)(++done_seq_ & 0x7fffffffu);   // never 0 (= "no flag"), wraps without overflow
    if (seq == 0) seq = (int)(++done_seq_ & 0x7fffffffu);
  }
  const bool fused = seq != 0 && k <= std::max<long>(ops_->L.tpw, fused_max);
  if (fused) {
    p.o_pose = d_cache_; p.o_twist = d_cache_ + 7 * n_; p.o_acc = d_cache_ + 13 * n_;
    p.done_flag = d_done_; p.done_seq = seq; p.done_count = d_done_count_;
  }
  launch_step(p, stream_);
  if (cache_valid_ && !fused) {   // keep the getter table current: only the stepped slots change
    OutArgs a;
    a.rec = d_rec_; a.idx = p.idx; a.n = k; a.by_slot = 1;
    a.pose = d_cache_; a.twist = d_cache_ + 7 * n_; a.acc = d_cache_ + 13 * n_;
    a.at_time = 0; a.t1 = 0.0; a.t_acc = t_acc_; a.t_base = d_tbase_;
    if (seq != 0 && k <= kOutputsBlock) { a.done_flag = d_done_; a.done_seq = seq; }
    else seq = 0;
    ops_->outputs(a, stream_);
  }
  TE_HIP_CHECK(hipGetLastError());
  // the pinned block is rewritten by the next flush, and the getters read the table right after this call
  if (seq != 0) wait_done(seq);
  else TE_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Batch::wait_done(int seq) {
  // The signalling kernel is the last launch of the flush (in-order stream): once its flag is here, the step kernel has
  // consumed the pinned inputs and the table rows are in host memory.  A round trip takes 10-20 us; after 2 ms of spinning
  // something else is going on (a busy device, a debugger) and the runtime's own wait takes over.
  const volatile int* flag = h_done_;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return;
    __builtin_ia32_pause();
    if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Batch::outputs(const int* slots, long n, double* pose, double* twist, double* acc, bool at_time, double t1) {
  flush();
  if (n <= 0) return;
  stage_reserve(n);
  if (slots) upload_slots(slots, n);
  OutArgs a;
  a.rec = d_rec_; a.idx = slots ? d_idx_ : nullptr; a.n = n;
  a.pose = pose ? d_aos_ : nullptr; a.twist = twist ? d_aos_ + 7 * n : nullptr; a.acc = acc ? d_aos_ + 13 * n : nullptr;
  a.at_time = at_time ? 1 : 0; a.t1 = t1; a.t_acc = t_acc_; a.t_base = d_tbase_;
  ops_->outputs(a, stream_);
  TE_HIP_CHECK(hipGetLastError());
  if (pose) TE_HIP_CHECK(hipMemcpyAsync(pose, a.pose, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, stream_));
  if (twist) TE_HIP_CHECK(hipMemcpyAsync(twist, a.twist, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, stream_));
  if (acc) TE_HIP_CHECK(hipMemcpyAsync(acc, a.acc, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, stream_));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Batch::outputs_dev(double* pose_dev, double* twist_dev, double* acc_dev, bool at_time, double t1) {
  flush();
  if (n_ == 0) return;
  OutArgs a;
  a.rec = d_rec_; a.idx = nullptr; a.n = n_; a.pose = pose_dev; a.twist = twist_dev; a.acc = acc_dev;
  a.at_time = at_time ? 1 : 0; a.t1 = t1; a.t_acc = t_acc_; a.t_base = d_tbase_;
  ops_->outputs(a, stream_);
  TE_HIP_CHECK(hipGetLastError());
}

void Batch::outputs_one(long slot, double* pose, double* twist, double* acc, bool at_time, double t1) {
  flush();
  const int one = (int)slot;
  ++epoch_getters_;
  const bool big = n_ > kCacheMax;
  if (at_time || n_ > kCacheBigMax || (big && !cache_valid_ && !big_sweeps_ && epoch_getters_ <= kBigDirect)) {
    outputs(&one, 1, pose, twist, acc, at_time, t1);
    return;
  }
  // a host-resident table of every slot's outputs, written by the kernels themselves; filled once, then kept current by
  // flush().  Small batches (the reference's scale) always; large ones when the caller sweeps them (batch_store.hpp)
  if (!cache_valid_) {
    cache_reserve(n_);
    OutArgs a;
    a.rec = d_rec_; a.idx = nullptr; a.n = n_;
    a.pose = d_cache_; a.twist = d_cache_ + 7 * n_; a.acc = d_cache_ + 13 * n_;
    a.at_time = 0; a.t1 = 0.0; a.t_acc = t_acc_; a.t_base = d_tbase_;
    ops_->outputs(a, stream_);
    TE_HIP_CHECK(hipGetLastError());
    TE_HIP_CHECK(hipStreamSynchronize(stream_));
    cache_valid_ = true;
  }
  if (pose) std::memcpy(pose, h_cache_ + (size_t)slot * 7, sizeof(double) * 7);
  if (twist) std::memcpy(twist, h_cache_ + (size_t)n_ * 7 + (size_t)slot * 6, sizeof(double) * 6);
  if (acc) std::memcpy(acc, h_cache_ + (size_t)n_ * 13 + (size_t)slot * 6, sizeof(double) * 6);
}

void Batch::intersect(const int* slots, long n, double t1, const double* origin, double radius, double* delta, double* pose) {
  flush();
  if (n <= 0) return;
  stage_reserve(n);
  if (slots) upload_slots(slots, n);
  IntersectArgs a;
  a.rec = d_rec_; a.idx = slots ? d_idx_ : nullptr; a.n = n; a.t1 = t1;
  a.origin[0] = origin[0]; a.origin[1] = origin[1]; a.origin[2] = origin[2]; a.radius = radius;
  a.t_acc = t_acc_; a.t_base = d_tbase_;
  a.delta = d_aos_; a.pose = pose ? d_aos_ + n : nullptr;
  ops_->intersect(a, stream_);
  TE_HIP_CHECK(hipGetLastError());
  TE_HIP_CHECK(hipMemcpyAsync(delta, a.delta, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
  if (pose) TE_HIP_CHECK(hipMemcpyAsync(pose, a.pose, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, stream_));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Batch::intersect_dev(double t1, const double* origin, double radius, double* delta_dev, double* pose_dev) {
  flush();
  if (n_ == 0) return;
  IntersectArgs a;
  a.rec = d_rec_; a.idx = nullptr; a.n = n_; a.t1 = t1;
  a.origin[0] = origin[0]; a.origin[1] = origin[1]; a.origin[2] = origin[2]; a.radius = radius;
  a.t_acc = t_acc_; a.t_base = d_tbase_; a.delta = delta_dev; a.pose = pose_dev;
  ops_->intersect(a, stream_);
  TE_HIP_CHECK(hipGetLastError());
}

void Batch::gate_reset(long first, long count) {
  if (count <= 0 || !d_gate_ring_) return;
  const int W = gate_window_;
  TE_HIP_CHECK(hipMemsetAsync(d_gate_ring_ + first * 2 * W, 0, sizeof(double) * 2 * W * count, stream_));
  TE_HIP_CHECK(hipMemsetAsync(d_gate_sum_ + first * 2, 0, sizeof(double) * 2 * count, stream_));
  TE_HIP_CHECK(hipMemsetAsync(d_gate_state_ + first * 2, 0, sizeof(int) * 2 * count, stream_));
  hipLaunchKernelGGL(gate_reset_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream_, d_gate_prev_, first, count);
  TE_HIP_CHECK(hipGetLastError());
}

void Batch::gate_reserve(int window) {
  if (window <= 0 || window >= (1 << 30)) throw std::runtime_error("target_estimation_amd: bad filters_length");
  if (window == gate_window_ && gate_cap_ >= cap_) return;
  const bool keep = (window == gate_window_) && d_gate_ring_;
  const long old_cap = gate_cap_;
  double* ring = nullptr; double* sum = nullptr; int* st = nullptr; double* prev = nullptr;
  TE_HIP_CHECK(hipMalloc((void**)&ring, sizeof(double) * 2 * window * cap_));
  TE_HIP_CHECK(hipMalloc((void**)&sum, sizeof(double) * 2 * cap_));
  TE_HIP_CHECK(hipMalloc((void**)&st, sizeof(int) * 2 * cap_));
  TE_HIP_CHECK(hipMalloc((void**)&prev, sizeof(double) * 7 * cap_));
  if (keep) {
    TE_HIP_CHECK(hipMemcpyAsync(ring, d_gate_ring_, sizeof(double) * 2 * window * old_cap, hipMemcpyDeviceToDevice, stream_));
    TE_HIP_CHECK(hipMemcpyAsync(sum, d_gate_sum_, sizeof(double) * 2 * old_cap, hipMemcpyDeviceToDevice, stream_));
    TE_HIP_CHECK(hipMemcpyAsync(st, d_gate_state_, sizeof(int) * 2 * old_cap, hipMemcpyDeviceToDevice, stream_));
    TE_HIP_CHECK(hipMemcpyAsync(prev, d_gate_prev_, sizeof(double) * 7 * old_cap, hipMemcpyDeviceToDevice, stream_));
  }
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  device_free(d_gate_ring_); device_free(d_gate_sum_); device_free(d_gate_state_); device_free(d_gate_prev_);
  d_gate_ring_ = ring; d_gate_sum_ = sum; d_gate_state_ = st; d_gate_prev_ = prev;
  gate_window_ = window;
  gate_cap_ = cap_;
  gate_reset(keep ? old_cap : 0, keep ? cap_ - old_cap : cap_);
}

void Batch::gate_move(long src, long dst) {
  if (!d_gate_ring_ || src >= gate_cap_ || dst >= gate_cap_) return;
  const int W = gate_window_;
  TE_HIP_CHECK(hipMemcpyAsync(d_gate_ring_ + dst * 2 * W, d_gate_ring_ + src * 2 * W, sizeof(double) * 2 * W, hipMemcpyDeviceToDevice, stream_));
  TE_HIP_CHECK(hipMemcpyAsync(d_gate_sum_ + dst * 2, d_gate_sum_ + src * 2, sizeof(double) * 2, hipMemcpyDeviceToDevice, stream_));
  TE_HIP_CHECK(hipMemcpyAsync(d_gate_state_ + dst * 2, d_gate_state_ + src * 2, sizeof(int) * 2, hipMemcpyDeviceToDevice, stream_));
  TE_HIP_CHECK(hipMemcpyAsync(d_gate_prev_ + dst * 7, d_gate_prev_ + src * 7, sizeof(double) * 7, hipMemcpyDeviceToDevice, stream_));
}

void Batch::intersect_gated(const int* slots, long n, double t1, const double* origin, double radius, double pos_th,
                            double ang_th, int window, double* delta, double* pose, unsigned char* converged, double* filt) {
  flush();
  if (n <= 0) return;
  gate_reserve(window);
  stage_reserve(n);
  if (slots) upload_slots(slots, n);
  IntersectArgs a;
  a.rec = d_rec_; a.idx = slots ? d_idx_ : nullptr; a.n = n; a.t1 = t1;
  a.origin[0] = origin[0]; a.origin[1] = origin[1]; a.origin[2] = origin[2]; a.radius = radius;
  a.t_acc = t_acc_; a.t_base = d_tbase_;
  a.delta = d_aos_; a.pose = d_aos_ + n;                 // [n] + [n][7]
  ops_->intersect(a, stream_);
  GateArgs g;
  g.idx = a.idx; g.n = n; g.window = gate_window_; g.delta = a.delta; g.pose = a.pose; g.pos_th = pos_th; g.ang_th = ang_th;
  g.ring = d_gate_ring_; g.sum = d_gate_sum_; g.state = d_gate_state_; g.prev = d_gate_prev_;
  g.converged = d_mask_; g.filt = filt ? d_aos_ + 8 * n : nullptr;   // [n][2] after delta + pose
  g.var = nullptr;
  hipLaunchKernelGGL(gate_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream_, g);
  TE_HIP_CHECK(hipGetLastError());
  if (delta) TE_HIP_CHECK(hipMemcpyAsync(delta, a.delta, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
  if (pose) TE_HIP_CHECK(hipMemcpyAsync(pose, a.pose, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, stream_));
  if (converged) TE_HIP_CHECK(hipMemcpyAsync(converged, d_mask_, (size_t)n, hipMemcpyDeviceToHost, stream_));
  if (filt) TE_HIP_CHECK(hipMemcpyAsync(filt, g.filt, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, stream_));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Batch::intersect_gated_dev(double t1, const double* origin, double radius, double pos_th, double ang_th, int window,
                                double* delta_dev, double* pose_dev, unsigned char* converged_dev) {
  if (n_ == 0) return;
  intersect_dev(t1, origin, radius, delta_dev, pose_dev);
  gate_update_dev(delta_dev, pose_dev, pos_th, ang_th, window, converged_dev, nullptr, nullptr);
}

void Batch::gate_update_dev(const double* delta_dev, const double* pose_dev, double pos_th, double ang_th, int window,
                            unsigned char* converged_dev, double* filt_dev, double* var_dev) {
  flush();
  if (n_ == 0) return;
  gate_reserve(window);
  GateArgs g;
  g.idx = nullptr; g.n = n_; g.window = gate_window_; g.delta = delta_dev; g.pose = pose_dev; g.pos_th = pos_th; g.ang_th = ang_th;
  g.ring = d_gate_ring_; g.sum = d_gate_sum_; g.state = d_gate_state_; g.prev = d_gate_prev_;
  g.converged = converged_dev; g.filt = filt_dev; g.var = var_dev;
  hipLaunchKernelGGL(gate_kernel, dim3((unsigned)((n_ + 127) / 128)), dim3(128), 0, stream_, g);
  TE_HIP_CHECK(hipGetLastError());
}

void Batch::pack_meas_dev(const double* aos_dev, long n, void* soa_dev, long ld) {
  ops_->pack_meas(aos_dev, n, soa_dev, ld, stream_);
  TE_HIP_CHECK(hipGetLastError());
}

void Batch::get_state(const int* slots, long n, double* x, double* P) {
  flush();
  if (n <= 0) return;
  const int N = ops_->L.n;
  stage_reserve(n);
  if (slots) upload_slots(slots, n);
  // scratch: up to kStateScratchKeep bytes are kept between calls (a caller reading one target's covariance at a time -- the
  // reference test's getEstimator()->getP() -- pays no allocation, and no hipFree: that synchronises the whole device, a
  // resident kernel of another batch included); larger requests are freed again
  const size_t bx = x ? sizeof(double) * (size_t)N * n : 0, bP = P ? sizeof(double) * (size_t)N * N * n : 0;
  char* big = nullptr;
  char* buf = nullptr;
  if (bx + bP <= kStateScratchKeep) {
    if (bx + bP > state_scratch_bytes_) {
      TE_HIP_CHECK(hipStreamSynchronize(stream_));
      device_free(d_state_scratch_); d_state_scratch_ = nullptr; state_scratch_bytes_ = 0;
      const size_t want = std::max<size_t>(bx + bP, std::min<size_t>(2 * state_scratch_bytes_ + 4096, kStateScratchKeep));
      TE_HIP_CHECK(hipMalloc((void**)&d_state_scratch_, want));
      state_scratch_bytes_ = want;
    }
    buf = d_state_scratch_;
  } else {
    TE_HIP_CHECK(hipMalloc((void**)&big, bx + bP));
    buf = big;
  }
  double* dx = x ? reinterpret_cast<double*>(buf) : nullptr;
  double* dP = P ? reinterpret_cast<double*>(buf + bx) : nullptr;
  ops_->get_state(d_rec_, slots ? d_idx_ : nullptr, n, dx, dP, stream_);
  TE_HIP_CHECK(hipGetLastError());
  if (x) TE_HIP_CHECK(hipMemcpyAsync(x, dx, bx, hipMemcpyDeviceToHost, stream_));
  if (P) TE_HIP_CHECK(hipMemcpyAsync(P, dP, bP, hipMemcpyDeviceToHost, stream_));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  if (big) device_free(big);
}

void Batch::set_state(const int* slots, long n, const double* x, const double* P, const double* unwrap) {
  touch();
  if (n <= 0) return;
  const int N = ops_->L.n;
  stage_reserve(n);
  if (slots) upload_slots(slots, n);
  double *dx = nullptr, *dP = nullptr, *du = nullptr;
  if (x) { TE_HIP_CHECK(hipMalloc((void**)&dx, sizeof(double) * N * n)); TE_HIP_CHECK(hipMemcpyAsync(dx, x, sizeof(double) * N * n, hipMemcpyHostToDevice, stream_)); }
  if (P) { TE_HIP_CHECK(hipMalloc((void**)&dP, sizeof(double) * N * N * n)); TE_HIP_CHECK(hipMemcpyAsync(dP, P, sizeof(double) * N * N * n, hipMemcpyHostToDevice, stream_)); }
  if (unwrap) { TE_HIP_CHECK(hipMalloc((void**)&du, sizeof(double) * 3 * n)); TE_HIP_CHECK(hipMemcpyAsync(du, unwrap, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream_)); }
  ops_->set_state(d_rec_, slots ? d_idx_ : nullptr, n, dx, dP, du, stream_);
  TE_HIP_CHECK(hipGetLastError());
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  device_free(dx); device_free(dP); device_free(du);
}

long long Batch::n_measurements(long slot) {
  flush();
  // a caller that asks target after target (get_n_measurements is one of the reference's ten symbols) gets a host copy of all
  // the counters after a few single reads; any step drops it (flush() with queued steps, touch())
  if (!nm_valid_ && ++nm_reads_ > kCounterDirect) {
    h_nm_.resize((size_t)n_);
    TE_HIP_CHECK(hipMemcpyAsync(h_nm_.data(), d_nmbase_, sizeof(int) * (size_t)n_, hipMemcpyDeviceToHost, stream_));
    TE_HIP_CHECK(hipStreamSynchronize(stream_));
    nm_valid_ = true;
  }
  if (nm_valid_) return (long long)h_nm_[(size_t)slot] + nm_acc_;
  int v = 0;
  TE_HIP_CHECK(hipMemcpyAsync(&v, d_nmbase_ + slot, sizeof(int), hipMemcpyDeviceToHost, stream_));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  return (long long)v + nm_acc_;
}

void Batch::times(double* out) {
  flush();
  if (n_ == 0) return;
  TE_HIP_CHECK(hipMemcpyAsync(out, d_tbase_, sizeof(double) * n_, hipMemcpyDeviceToHost, stream_));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  for (long i = 0; i < n_; ++i) out[i] += t_acc_;
}

double Batch::time(long slot) {
  flush();
  double v = 0;
  TE_HIP_CHECK(hipMemcpyAsync(&v, d_tbase_ + slot, sizeof(double), hipMemcpyDeviceToHost, stream_));
  TE_HIP_CHECK(hipStreamSynchronize(stream_));
  return v + t_acc_;
}

}  // namespace te
