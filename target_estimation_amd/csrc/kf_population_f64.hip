// kf_population_f64.hip -- the one-launch population tick in the reference's arithmetic (kf_population_impl.hpp).
#include "kf_population_impl.hpp"

namespace te {

void launch_population_step_f64(const StepParams parts[4], bool query, bool ab, bool reverse, hipStream_t s) {
  launch_population_step_t<double>(parts, query, ab, reverse, s);
}

}  // namespace te
