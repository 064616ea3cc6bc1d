// kf_ops.hpp -- type-erased launch table: one Ops per (model, precision, lanes-per-target).
// The four kf_model_*.hip translation units instantiate the kernels; the host side
// (batch_store.cpp) only sees this table.
#pragma once
#include <hip/hip_runtime.h>

#include "kf_aux.hpp"
#include "te_layout.hpp"

namespace te {

struct StepParams {
  char* rec;
  char* rec_out = nullptr;        // dense single-tick launches: write the records here instead of in place (StepArgs::rec_out)
  const void* qr;                 // one [Q | R] block, or (cls != null) a table of them
  const int* cls = nullptr;       // per-slot parameter class: selects the per-class kernels
  long n;
  const int* idx;                 // non-null selects the indexed kernel
  const void* meas;               // SoA [7][meas_ld] in the batch precision, or null (predict only)
  long meas_ld;
  const unsigned char* has_meas;
  const double* dt_per;
  double dt;
  double* t_base;
  int* nm_base;
  int n_ticks = 1;        // > 1: temporally fused launch (state stays in registers for n_ticks ticks)
  long tick_stride = 0;   // elements between the measurement blocks of consecutive ticks
  long has_stride = 0;
  // fused own-time sphere query (dense single-tick launches): q_delta != null selects it
  double q_origin[3] = {0, 0, 0};
  double q_radius = 0;
  double* q_delta = nullptr;
  double* q_pose = nullptr;
  int nt_meas = 0;        // nontemporal measurement loads (kf_step.hpp StepArgs::nt_meas)
  int reverse = 0;        // walk the tiles last-to-first (zig-zag between consecutive ticks: kf_step.hpp StepArgs)
  // indexed launches: also write the per-slot getter table and a completion flag (StepArgs::o_pose); more than L.tpw entries
  // need done_count (a device word, zero between launches)
  double* o_pose = nullptr;
  double* o_twist = nullptr;
  double* o_acc = nullptr;
  int* done_flag = nullptr;
  int done_seq = 0;
  int* done_count = nullptr;
  // resident ("live") launch: live_posted != null (kf_step.hpp StepArgs::live_*); n_ticks = the most ticks it will serve
  const long long* live_posted = nullptr;
  long long* live_mirror = nullptr;
  int* live_progress = nullptr;
  int* live_done = nullptr;
  long live_ring = 0, live_first = 0;
  unsigned live_spin_limit = 0;
  unsigned long long live_idle_ticks = 0;
  int live_flags = 0;
  double* live_pose = nullptr;   // SoA [7][live_pose_ld] per-tick pose output of a live launch, or null
  long live_pose_ld = 0;
};

struct Ops {
  LayoutInfo L;
  int wpb;  // wavefronts per workgroup of the step kernel
  bool fused_query;  // step() honours StepParams::q_delta
  void (*step)(const StepParams&, hipStream_t);
  // wavefronts of the live kernel the device can hold at once (0: this (model, precision, layout) has no live kernel)
  long (*live_capacity)(int with_outputs);   // resident wavefronts of the plain / the query-and-pose-output variant
  void (*init)(const InitArgs&, hipStream_t);
  void (*get_state)(char* rec, const int* idx, long n, double* x, double* P, hipStream_t);
  void (*set_state)(char* rec, const int* idx, long n, const double* x, const double* P, const double* uw, hipStream_t);
  void (*move_record)(char* rec, long src, long dst, double* t_base, int* nm_base, int* cls, hipStream_t);
  void (*move_records)(char* rec, const int* src_dev, const int* dst_dev, long m, double* t_base, int* nm_base, int* cls, hipStream_t);
  void (*outputs)(const OutArgs&, hipStream_t);
  void (*pack_meas)(const double* aos, long n, void* soa, long ld, hipStream_t);
  void (*intersect)(const IntersectArgs&, hipStream_t);
};

// g == 0 selects the default lanes-per-target of the (model, precision); nullptr if unsupported
const Ops* get_ops(int type, int dtype, int g);
const Ops* get_ops_uv(int dtype, int g);
const Ops* get_ops_ua(int dtype, int g);
const Ops* get_ops_ar(int dtype, int g);
const Ops* get_ops_av(int dtype, int g);

}  // namespace te
