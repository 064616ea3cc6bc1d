// kf_model_av.hip -- kernel instantiations of one motion model (see kf_step.hpp).
#include "kf_ops_impl.hpp"

namespace te {

const Ops* get_ops_av_sym(int dtype);   // kf_model_av_sym.hip

const Ops* get_ops_av(int dtype, int g) {
  if (dtype == F64) {
    if (g == 0) g = 6;   // 178 VGPRs, 2 waves per SIMD (G = 3 needs 270: one); profiles/r02_layout_sweep.txt
    switch (g) {
      case 3: return OpsImpl<ModelAV, double, 3>::get();
      case 6: return OpsImpl<ModelAV, double, 6>::get();
      case 101: return get_ops_av_sym(F64);  // symmetric-packed P, thread per target (ekf_sym.hpp)
      case 103: return OpsImpl<ModelAV, double, 3, LAYOUT_PACKED>::get();  // symmetric-packed P, 3 lanes per target
      case 106: return OpsImpl<ModelAV, double, 6, LAYOUT_PACKED>::get();  // symmetric-packed P, 6 lanes per target
      case 201: return OpsImpl<ModelAV, double, 1, LAYOUT_SEPARABLE>::get();  // axis-separable
      case 301: return OpsImpl<ModelAV, double, 1, LAYOUT_SEPARABLE_PACKED>::get();  // + symmetric-packed groups
      default: return nullptr;
    }
  } else if (dtype == F32) {
    if (g == 0) g = 3;
    switch (g) {
      case 1: return OpsImpl<ModelAV, float, 1>::get();
      case 101: return get_ops_av_sym(F32);  // symmetric-packed P, thread per target (ekf_sym.hpp)
      case 3: return OpsImpl<ModelAV, float, 3>::get();
      case 6: return OpsImpl<ModelAV, float, 6>::get();
      case 103: return OpsImpl<ModelAV, float, 3, LAYOUT_PACKED>::get();  // symmetric-packed P, 3 lanes per target
      case 106: return OpsImpl<ModelAV, float, 6, LAYOUT_PACKED>::get();  // symmetric-packed P, 6 lanes per target
      case 201: return OpsImpl<ModelAV, float, 1, LAYOUT_SEPARABLE>::get();  // axis-separable
      case 301: return OpsImpl<ModelAV, float, 1, LAYOUT_SEPARABLE_PACKED>::get();  // + symmetric-packed groups
      default: return nullptr;
    }
  }
  return nullptr;
}

}  // namespace te
