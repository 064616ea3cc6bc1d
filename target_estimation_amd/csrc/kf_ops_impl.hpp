// kf_ops_impl.hpp -- builds the Ops table entry of one (model, precision, G); included only by
// the kf_model_*.hip translation units.
#pragma once
#include "kf_ops.hpp"
#include "kf_step.hpp"
#include "kf_step_sep.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

namespace te {

// host-side launch parameters -> the kernels' argument block
template <typename T>
inline StepArgs<T> make_step_args(const StepParams& p) {
  StepArgs<T> a;
  a.rec = p.rec; a.rec_out = p.rec_out; a.qr = static_cast<const T*>(p.qr); a.cls = p.cls; a.n = p.n; a.idx = p.idx;
  a.meas = static_cast<const T*>(p.meas); a.meas_ld = p.meas_ld; a.has_meas = p.has_meas;
  a.dt_per = p.dt_per; a.dt = p.dt; a.t_base = p.t_base; a.nm_base = p.nm_base;
  a.n_ticks = p.n_ticks; a.tick_stride = p.tick_stride; a.has_stride = p.has_stride;
  a.q_origin[0] = p.q_origin[0]; a.q_origin[1] = p.q_origin[1]; a.q_origin[2] = p.q_origin[2];
  a.q_radius = p.q_radius; a.q_delta = p.q_delta; a.q_pose = p.q_pose;
  a.reverse = p.reverse; a.nt_meas = p.nt_meas;
  a.o_pose = p.o_pose; a.o_twist = p.o_twist; a.o_acc = p.o_acc; a.done_flag = p.done_flag; a.done_seq = p.done_seq; a.done_count = p.done_count;
  a.live_posted = p.live_posted; a.live_mirror = p.live_mirror; a.live_progress = p.live_progress; a.live_done = p.live_done;
  a.live_ring = p.live_ring; a.live_first = p.live_first;
  a.live_spin_limit = p.live_spin_limit; a.live_idle_ticks = p.live_idle_ticks; a.live_flags = p.live_flags; a.live_pose = p.live_pose; a.live_pose_ld = p.live_pose_ld;
  return a;
}

template <class M, typename T, int G, int LAYOUT = LAYOUT_FULL>
struct OpsImpl {
  using C = Cfg<M, T, G, LAYOUT>;

  static constexpr bool kHasLive = LAYOUT == LAYOUT_SEPARABLE_PACKED;
  static long live_capacity(int with_outputs) {   // with_outputs: the variant with the per-tick query / pose output (LIVE == 2)
    if constexpr (kHasLive) {
      int per_cu = 0, dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
      const void* kernel = with_outputs ? (const void*)kf_step_sep_kernel<M, T, LAYOUT, false, true, false, false, 2>
                                        : (const void*)kf_step_sep_kernel<M, T, LAYOUT, false, true, false, false, 1>;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess) return 0;
      // Measured on an MI355X with the relay's start word (tools/live_capacity.py --probe, profiles/r03_live_capacity.txt): the
      // largest grid that becomes resident is the query's figure for every kernel at up to 6 wavefronts per SIMD, but 28 per CU
      // where the query says 32 (8 per SIMD).  And a device that full starves everybody else: with 22 or more resident
      // wavefronts per CU (of 24 / 28) a device-to-device copy on another stream -- the caller's ring refill -- waited for the
      // session to end, with 18.4 of 20 it took its usual 0.4 ms.  So: at most 5 per SIMD, and one block per CU in hand.
      if (per_cu > 20) per_cu = 20;
      return (long)(per_cu > 1 ? per_cu - 1 : 1) * (long)prop.multiProcessorCount;
    } else {
      return 0;
    }
  }
  static void step(const StepParams& p, hipStream_t s) {
    if (p.n <= 0) return;
    StepArgs<T> a = make_step_args<T>(p);
    if (p.live_posted) {   // resident launch: one wavefront per workgroup, every workgroup resident (Batch::live_start checked the capacity)
      if constexpr (kHasLive) {
        if (p.idx || p.cls || p.rec_out || !p.live_progress || !p.live_mirror || !p.live_done || p.live_ring <= 0 || p.n_ticks < 1)
          throw std::runtime_error("target_estimation_amd: a live launch is a dense launch of a one-class batch over a measurement ring");
        const long waves_live = (p.n + C::TPW - 1) / C::TPW;
        // + 1: the relay wavefront (kf_step.hpp live_relay)
        if (p.q_delta || p.live_pose)
          hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, false, true, false, false, 2>), dim3((unsigned)waves_live + 1), dim3(64), 0, s, a);
        else
          hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, false, true, false, false, 1>), dim3((unsigned)waves_live + 1), dim3(64), 0, s, a);
        return;
      } else {
        throw std::runtime_error("target_estimation_amd: live mode needs the axis-separable layout with packed groups (the automatic choice for the shipped models)");
      }
    }
    if (p.o_pose && (!p.idx || (p.n > C::TPW && !p.done_count) || !p.o_twist || !p.o_acc || !p.done_flag))
      throw std::runtime_error("target_estimation_amd: the fused getter table needs an indexed launch (and a wavefront counter beyond one wavefront of entries)");
    static const int nt_env = [] { const char* e = std::getenv("TE_NT_MEAS"); return e ? std::atoi(e) : -1; }();
    a.nt_meas = nt_env >= 0 ? nt_env : p.nt_meas;
    if (p.q_delta && (p.idx || p.n_ticks > 1))
      throw std::runtime_error("target_estimation_amd: the fused query needs a dense single-tick launch");
    const long waves = (p.n + C::TPW - 1) / C::TPW;
    static const long small_grid = [] { const char* e = std::getenv("TE_SMALL_GRID_WAVES"); return e ? std::atol(e) : 1024L; }();
    // small (latency-bound) grids: one wavefront per workgroup spreads the waves over more CUs
    const int wpb_dense = waves <= small_grid ? 1 : C::WPB;
    const unsigned blocks = (unsigned)((waves + wpb_dense - 1) / wpb_dense);
    if (p.n_ticks > 1 && p.idx) throw std::runtime_error("target_estimation_amd: fused multi-tick launches are dense only");
    if (p.rec_out && (p.idx || p.n_ticks > 1 || p.q_delta))
      throw std::runtime_error("target_estimation_amd: A -> B ticks are dense single-tick launches without the fused query");
    // A few temporally fused instantiations do not fit the register file and would spill hundreds of bytes per lane to
    // scratch (228 / 116 / 340 / 352 B): for them a fused request is served tick by tick -- same results.
    constexpr bool kFusedSpills = (M::TYPE == ANGULAR_RATES && sizeof(T) == 8 && G == 3 && LAYOUT == LAYOUT_PACKED) ||
                                  (M::TYPE == ANGULAR_VELOCITIES && sizeof(T) == 4 && G == 1 && LAYOUT == LAYOUT_FULL) ||
                                  (M::TYPE == ANGULAR_VELOCITIES && sizeof(T) == 8 && G == 1 && LAYOUT == LAYOUT_PACKED) ||
                                  (M::TYPE == ANGULAR_RATES && sizeof(T) == 8 && G == 6 && LAYOUT == LAYOUT_PACKED);
    // A batch with several (Q, R) classes has no temporally fused kernel either: same tick-by-tick service (same results).
    if ((kFusedSpills || p.cls) && p.n_ticks > 1) {
      StepParams q = p;
      q.n_ticks = 1;
      for (int t = 0; t < p.n_ticks; ++t) {
        q.meas = p.meas ? static_cast<const char*>(p.meas) + (size_t)t * (size_t)p.tick_stride * sizeof(T) : nullptr;
        q.has_meas = p.has_meas ? p.has_meas + (long)t * p.has_stride : nullptr;
        step(q, s);
      }
      return;
    }
    if (p.cls && p.q_delta)
      throw std::runtime_error("target_estimation_amd: a batch with several (Q, R) classes has no fused sphere query (step, then target_batch_intersect_sphere_dev)");
    if constexpr (C::SEP) {
      // small (latency-bound) grids: one wavefront per workgroup spreads the waves over more CUs
      const int wpb = waves <= small_grid ? 1 : 4;
      const unsigned b4 = (unsigned)((waves + wpb - 1) / wpb);
      const dim3 blk(64 * wpb);
      if (p.cls && p.idx)
        hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, true, false, false, true>), dim3(b4), blk, 0, s, a);
      else if (p.cls && p.rec_out)
        hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, false, false, false, true, false, true>), dim3(b4), blk, 0, s, a);
      else if (p.cls)
        hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, false, false, false, true>), dim3(b4), blk, 0, s, a);
      else if (p.rec_out)
        hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, false, false, false, false, false, true>), dim3(b4), blk, 0, s, a);
      else if (p.n_ticks > 1)
        hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, false, true>), dim3(b4), blk, 0, s, a);
      else if (p.idx)
        hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, true>), dim3(b4), blk, 0, s, a);
      else if (p.q_delta)
        hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, false, false, true>), dim3(b4), blk, 0, s, a);
      else
        hipLaunchKernelGGL((kf_step_sep_kernel<M, T, LAYOUT, false>), dim3(b4), blk, 0, s, a);
    } else {
      if (p.cls && p.idx)
        hipLaunchKernelGGL((kf_step_kernel<M, T, G, LAYOUT, true, false, false, true>), dim3(blocks), dim3(wpb_dense * 64), 0, s, a);
      else if (p.cls && p.rec_out)
        hipLaunchKernelGGL((kf_step_kernel<M, T, G, LAYOUT, false, false, false, true, true>), dim3(blocks), dim3(wpb_dense * 64), 0, s, a);
      else if (p.cls)
        hipLaunchKernelGGL((kf_step_kernel<M, T, G, LAYOUT, false, false, false, true>), dim3(blocks), dim3(wpb_dense * 64), 0, s, a);
      else if (p.rec_out)
        hipLaunchKernelGGL((kf_step_kernel<M, T, G, LAYOUT, false, false, false, false, true>), dim3(blocks), dim3(wpb_dense * 64), 0, s, a);
      else if (p.n_ticks > 1) {
        if constexpr (!kFusedSpills)
          hipLaunchKernelGGL((kf_step_kernel<M, T, G, LAYOUT, false, true>), dim3(blocks), dim3(wpb_dense * 64), 0, s, a);
      }
      else if (p.idx)
        hipLaunchKernelGGL((kf_step_kernel<M, T, G, LAYOUT, true>), dim3(blocks), dim3(wpb_dense * 64), 0, s, a);
      else if (p.q_delta)
        hipLaunchKernelGGL((kf_step_kernel<M, T, G, LAYOUT, false, false, true>), dim3(blocks), dim3(wpb_dense * 64), 0, s, a);
      else
        hipLaunchKernelGGL((kf_step_kernel<M, T, G, LAYOUT, false>), dim3(blocks), dim3(wpb_dense * 64), 0, s, a);
    }
  }
  static void init(const InitArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL((init_kernel<M, T, G, LAYOUT>), dim3((unsigned)((a.n + 127) / 128)), dim3(128), 0, s, a);
  }
  static void get_state(char* rec, const int* idx, long n, double* x, double* P, hipStream_t s) {
    if (n <= 0) return;
    const long th = n * C::N;
    hipLaunchKernelGGL((get_state_kernel<M, T, G, LAYOUT>), dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, rec, idx, n, x, P);
  }
  static void set_state(char* rec, const int* idx, long n, const double* x, const double* P, const double* uw, hipStream_t s) {
    if (n <= 0) return;
    const long th = n * C::N;
    hipLaunchKernelGGL((set_state_kernel<M, T, G, LAYOUT>), dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, rec, idx, n, x, P, uw);
  }
  static void move_records(char* rec, const int* src, const int* dst, long m, double* t_base, int* nm_base, int* cls, hipStream_t s) {
    const int th = C::G * C::RW;
    for (long done = 0; done < m; done += 65535) {   // grid.y limit
      const long part = std::min<long>(65535, m - done);
      hipLaunchKernelGGL((move_records_kernel<M, T, G, LAYOUT>), dim3((th + 255) / 256, (unsigned)part), dim3(256), 0, s, rec, src + done,
                         dst + done, part, t_base, nm_base, cls);
    }
  }
  static void move_record(char* rec, long src, long dst, double* t_base, int* nm_base, int* cls, hipStream_t s) {
    const int th = C::G * C::RW;
    hipLaunchKernelGGL((move_record_kernel<M, T, G, LAYOUT>), dim3((th + 255) / 256), dim3(256), 0, s, rec, src, dst, t_base, nm_base, cls);
  }
  static void outputs(const OutArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL((outputs_kernel<M, T, G, LAYOUT>), dim3((unsigned)((a.n + kOutputsBlock - 1) / kOutputsBlock)), dim3(kOutputsBlock), 0, s, a);
  }
  static void pack_meas(const double* aos, long n, void* soa, long ld, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL((pack_meas_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, aos, n, static_cast<T*>(soa), ld);
  }
  static void intersect(const IntersectArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL((intersect_kernel<M, T, G, LAYOUT>), dim3((unsigned)((a.n + 63) / 64)), dim3(64), 0, s, a);
  }
  static const Ops* get() {
    static const Ops ops = {
        LayoutInfo{C::N, C::K, G, LAYOUT, C::TPW, C::LPT, C::RW, C::TILE_BYTES, C::TILE_PAYLOAD},
        C::WPB, true, &step, &live_capacity, &init, &get_state, &set_state, &move_record, &move_records, &outputs, &pack_meas, &intersect};
    return &ops;
  }
};

}  // namespace te
