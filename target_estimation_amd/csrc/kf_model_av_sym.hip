// kf_model_av_sym.hip -- the angular-velocities EKF on the upper triangle, thread per target (ekf_sym.hpp), in a
// translation unit of its own because it is built WITHOUT the SLP vectorizer (Makefile: -fno-slp-vectorize).
// The step keeps the 78-word triangle in registers and updates it in place, in an order chosen so that every read still
// sees the prior covariance and no more than 36 + 18 temporaries are live.  SLP pairs the fp32 products into v_pk_fma_f32,
// whose operands are aligned register pairs: it hoists whole blocks of the update across the scheduling barriers to form
// them, and the allocation goes from 160 to 280 registers -- one wavefront per SIMD instead of three (195 us instead of
// 133 us per 10^6-target tick, profiles/r02_slp_ab.txt).  The other kernels are neutral to the flag and keep the default.
#include "kf_ops_impl.hpp"

namespace te {

const Ops* get_ops_av_sym(int dtype) {
  if (dtype == F64) return OpsImpl<ModelAV, double, 1, LAYOUT_PACKED>::get();
  if (dtype == F32) return OpsImpl<ModelAV, float, 1, LAYOUT_PACKED>::get();
  return nullptr;
}

}  // namespace te
