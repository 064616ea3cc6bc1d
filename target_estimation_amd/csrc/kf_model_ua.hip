// kf_model_ua.hip -- kernel instantiations of one motion model (see kf_step.hpp).
#include "kf_ops_impl.hpp"

namespace te {

const Ops* get_ops_ua(int dtype, int g) {
  if (dtype == F64) {
    if (g == 0) g = 3;
    switch (g) {
      case 1: return OpsImpl<ModelUA, double, 1>::get();
      case 101: return OpsImpl<ModelUA, double, 1, LAYOUT_PACKED>::get();  // symmetric-packed P
      case 3: return OpsImpl<ModelUA, double, 3>::get();
      case 103: return OpsImpl<ModelUA, double, 3, LAYOUT_PACKED>::get();  // symmetric-packed P, 3 lanes per target
      case 201: return OpsImpl<ModelUA, double, 1, LAYOUT_SEPARABLE>::get();  // axis-separable
      case 301: return OpsImpl<ModelUA, double, 1, LAYOUT_SEPARABLE_PACKED>::get();  // + symmetric-packed groups
      default: return nullptr;
    }
  } else if (dtype == F32) {
    if (g == 0) g = 3;   // 90 VGPRs (5 waves per SIMD); G = 1 needs 210 and is no faster (profiles/r02_layout_sweep.txt)
    switch (g) {
      case 1: return OpsImpl<ModelUA, float, 1>::get();
      case 101: return OpsImpl<ModelUA, float, 1, LAYOUT_PACKED>::get();  // symmetric-packed P
      case 3: return OpsImpl<ModelUA, float, 3>::get();
      case 103: return OpsImpl<ModelUA, float, 3, LAYOUT_PACKED>::get();  // symmetric-packed P, 3 lanes per target
      case 201: return OpsImpl<ModelUA, float, 1, LAYOUT_SEPARABLE>::get();  // axis-separable
      case 301: return OpsImpl<ModelUA, float, 1, LAYOUT_SEPARABLE_PACKED>::get();  // + symmetric-packed groups
      default: return nullptr;
    }
  }
  return nullptr;
}

}  // namespace te
