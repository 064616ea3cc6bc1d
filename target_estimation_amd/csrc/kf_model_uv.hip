// kf_model_uv.hip -- kernel instantiations of one motion model (see kf_step.hpp).
#include "kf_ops_impl.hpp"

namespace te {

const Ops* get_ops_uv(int dtype, int g) {
  if (dtype == F64) {
    if (g == 0) g = 1;   // thread per target: 140 / 112 VGPRs (3-4 waves per SIMD), 3-9 % faster than G = 3 at 10^6 targets (profiles/r02_layout_sweep.txt)
    switch (g) {
      case 1: return OpsImpl<ModelUV, double, 1>::get();
      case 101: return OpsImpl<ModelUV, double, 1, LAYOUT_PACKED>::get();  // symmetric-packed P
      case 3: return OpsImpl<ModelUV, double, 3>::get();
      case 103: return OpsImpl<ModelUV, double, 3, LAYOUT_PACKED>::get();  // symmetric-packed P, 3 lanes per target
      case 201: return OpsImpl<ModelUV, double, 1, LAYOUT_SEPARABLE>::get();  // axis-separable
      case 301: return OpsImpl<ModelUV, double, 1, LAYOUT_SEPARABLE_PACKED>::get();  // + symmetric-packed groups
      default: return nullptr;
    }
  } else if (dtype == F32) {
    if (g == 0) g = 1;   // thread per target: 140 / 112 VGPRs (3-4 waves per SIMD), 3-9 % faster than G = 3 at 10^6 targets (profiles/r02_layout_sweep.txt)
    switch (g) {
      case 1: return OpsImpl<ModelUV, float, 1>::get();
      case 101: return OpsImpl<ModelUV, float, 1, LAYOUT_PACKED>::get();  // symmetric-packed P
      case 3: return OpsImpl<ModelUV, float, 3>::get();
      case 103: return OpsImpl<ModelUV, float, 3, LAYOUT_PACKED>::get();  // symmetric-packed P, 3 lanes per target
      case 201: return OpsImpl<ModelUV, float, 1, LAYOUT_SEPARABLE>::get();  // axis-separable
      case 301: return OpsImpl<ModelUV, float, 1, LAYOUT_SEPARABLE_PACKED>::get();  // + symmetric-packed groups
      default: return nullptr;
    }
  }
  return nullptr;
}

}  // namespace te
