// kf_population_f32.hip -- the one-launch population tick in fp32 (kf_population_impl.hpp), and the precision switch.
#include "kf_population_impl.hpp"

namespace te {

void launch_population_step_f64(const StepParams parts[4], bool query, bool ab, bool reverse, hipStream_t s);

void launch_population_step(int dtype, const StepParams parts[4], bool query, bool ab, bool reverse, hipStream_t s) {
  if (dtype == F64) launch_population_step_f64(parts, query, ab, reverse, s);
  else launch_population_step_t<float>(parts, query, ab, reverse, s);
}

}  // namespace te
