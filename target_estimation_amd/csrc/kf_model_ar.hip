// kf_model_ar.hip -- kernel instantiations of one motion model (see kf_step.hpp).
#include "kf_ops_impl.hpp"

namespace te {

const Ops* get_ops_ar(int dtype, int g) {
  if (dtype == F64) {
    if (g == 0) g = 6;   // 184 VGPRs, 2 waves per SIMD (G = 3 needs 370: one); same speed (profiles/r02_layout_sweep.txt)
    switch (g) {
      case 3: return OpsImpl<ModelAR, double, 3>::get();
      case 6: return OpsImpl<ModelAR, double, 6>::get();
      case 103: return OpsImpl<ModelAR, double, 3, LAYOUT_PACKED>::get();  // symmetric-packed P, 3 lanes per target
      case 106: return OpsImpl<ModelAR, double, 6, LAYOUT_PACKED>::get();  // symmetric-packed P, 6 lanes per target
      case 201: return OpsImpl<ModelAR, double, 1, LAYOUT_SEPARABLE>::get();  // axis-separable
      case 301: return OpsImpl<ModelAR, double, 1, LAYOUT_SEPARABLE_PACKED>::get();  // + symmetric-packed groups
      default: return nullptr;
    }
  } else if (dtype == F32) {
    if (g == 0) g = 6;
    switch (g) {
      case 2: return OpsImpl<ModelAR, float, 2>::get();
      case 3: return OpsImpl<ModelAR, float, 3>::get();
      case 6: return OpsImpl<ModelAR, float, 6>::get();
      case 102: return OpsImpl<ModelAR, float, 2, LAYOUT_PACKED>::get();  // symmetric-packed P, 2 lanes per target
      case 103: return OpsImpl<ModelAR, float, 3, LAYOUT_PACKED>::get();  // symmetric-packed P, 3 lanes per target
      case 106: return OpsImpl<ModelAR, float, 6, LAYOUT_PACKED>::get();  // symmetric-packed P, 6 lanes per target
      case 201: return OpsImpl<ModelAR, float, 1, LAYOUT_SEPARABLE>::get();  // axis-separable
      case 301: return OpsImpl<ModelAR, float, 1, LAYOUT_SEPARABLE_PACKED>::get();  // + symmetric-packed groups
      default: return nullptr;
    }
  }
  return nullptr;
}

}  // namespace te
