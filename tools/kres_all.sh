#!/bin/bash
# The compiler's resource table of every step kernel of the library, with the flags of csrc/Makefile (CPU only: hipcc cross-compiles).
#   bash tools/kres_all.sh > profiles/r04_kernel_resources.txt
cd "$(dirname "$0")/../target_estimation_amd/csrc"
echo "# hipcc -Rpass-analysis=kernel-resource-usage, every kf_step* instantiation of the library at round 4 (tools/kres_all.sh, tools/kres.py; flags of csrc/Makefile)."
echo "# kf_step_sep_kernel<model,T,layout,INDEXED,FUSED,QUERY,PERQR,LIVE (0 / 1 resident / 2 resident with per-tick query or pose output),AB>; kf_step_kernel<model,T,G,layout,INDEXED,FUSED,QUERY,PERQR,AB>;"
echo "# kf_step_population_kernel<T,QUERY,AB>  (0/1 = false/true; trailing defaults omitted; lds = static LDS per workgroup: resident kernels are one wavefront per workgroup)"
for f in kf_model_uv kf_model_ua kf_model_ar kf_model_av kf_model_av_sym kf_population_f64 kf_population_f32; do
  extra=""
  case $f in kf_model_ar|kf_model_av) extra="-mllvm -disable-machine-licm";; kf_model_av_sym) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c $f.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 ../../tools/kres.py kf_step
done
