// kf_population_impl.hpp -- included by kf_population_f64.hip / kf_population_f32.hip (one precision per translation unit: the
// kernel holds the step of all four motion models, a minute of compile time each).
#pragma once
#include "kf_ops_impl.hpp"
#include "kf_population.hpp"

namespace te {

template <typename T>
void launch_population_step_t(const StepParams parts[4], bool query, bool ab, bool reverse, hipStream_t s) {
  constexpr int TPW = 64;   // thread per target in every separable layout
  PopulationArgs<T> p;
  long waves_max = 0;
  for (int k = 0; k < 4; ++k) {
    const StepParams& q = parts[k];
    if (q.n > 0 && (q.idx || q.cls || q.n_ticks != 1 || q.live_posted || q.o_pose || (query && !q.q_delta) || (ab && !q.rec_out) || (query && ab)))
      throw std::runtime_error("target_estimation_amd: a population launch takes dense single ticks of one-class batches");
    waves_max = std::max(waves_max, (q.n + TPW - 1) / TPW);
  }
  static const long small_grid = [] { const char* e = std::getenv("TE_SMALL_GRID_WAVES"); return e ? std::atol(e) : 1024L; }();
  static const int nt_env = [] { const char* e = std::getenv("TE_NT_MEAS"); return e ? std::atoi(e) : -1; }();
  // (as for the single-batch launches: small, latency-bound grids spread one wavefront per workgroup over the CUs)
  const int wpb = waves_max <= small_grid ? 1 : 4;
  unsigned end = 0;
  for (int k = 0; k < 4; ++k) {
    StepParams q = parts[k];
    q.reverse = 0;   // the order is reversed for the whole grid (PopulationArgs::reverse_blocks)
    p.part[k] = make_step_args<T>(q);
    if (nt_env >= 0) p.part[k].nt_meas = nt_env;
    const long waves = (q.n + TPW - 1) / TPW;
    end += (unsigned)((waves + wpb - 1) / wpb);
    p.end[k] = end;
  }
  if (end == 0) return;
  p.reverse_blocks = reverse ? 1 : 0;
  const dim3 blk(64 * wpb);
  if (query) hipLaunchKernelGGL((kf_step_population_kernel<T, true, false>), dim3(end), blk, 0, s, p);
  else if (ab) hipLaunchKernelGGL((kf_step_population_kernel<T, false, true>), dim3(end), blk, 0, s, p);
  else hipLaunchKernelGGL((kf_step_population_kernel<T, false, false>), dim3(end), blk, 0, s, p);
}

}  // namespace te
