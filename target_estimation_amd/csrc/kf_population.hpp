// kf_population.hpp -- the tick of a whole manager (one batch per motion model) as ONE launch: host side of
// kf_step_population_kernel (kf_step_sep.hpp).  Reference semantics served: every target of every model advances each
// tick (src/target_manager.cpp:190-225); the models' filters are independent, so their order inside the tick is free.
#pragma once
#include <hip/hip_runtime.h>

#include "kf_ops.hpp"

namespace te {

// parts[k] = the dense single-tick launch parameters of the batch of model type k (ModelType order), n == 0 for an absent
// model.  Every present batch: axis-separable layout with packed groups, one (Q, R) class, no slot list, one tick, the same
// precision `dtype`.  query: the fused own-time sphere query (q_delta set in every present part).  ab: A -> B tick (rec_out set
// in every present part; not together with the query).  reverse: walk the whole population last to first (zig-zag).
void launch_population_step(int dtype, const StepParams parts[4], bool query, bool ab, bool reverse, hipStream_t s);

}  // namespace te
