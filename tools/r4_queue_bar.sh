#!/bin/bash
# Round 4: the one-target ABI's queue (host writes, kernel reads) in device memory behind the PCIe BAR (TE_QUEUE_BAR=1) against host-mapped memory:
# first the tests that hammer that path (a stale read of the reused queue block would show there), then the reference's call pattern from C.
set -o pipefail
OUT=$PWD/gpurun_out/r4queuebar
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_c_abi_program.py tests/test_gpu_edge_cases.py tests/test_gpu_by_id.py tests/test_gpu_getters_log.py tests/test_oracle_harness.py tests/test_highprec_kat.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests (default: small flushes behind the BAR) rc=$rc"; tail -3 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/extended/soak_one_target.py 60 > $OUT/one_target.txt 2>&1; echo "one-target fuzz rc=$?"; tail -1 $OUT/one_target.txt
gcc -O2 -I include/target_estimation_amd tools/scalar_abi_rate.c -o /tmp/scalar_abi_rate -L target_estimation_amd/lib -ltarget_estimation_amd -Wl,-rpath,$PWD/target_estimation_amd/lib -Wl,-rpath,/opt/rocm/lib
for mode in 1 0 1 0; do
  echo "== TE_QUEUE_BAR=$mode" | tee -a $OUT/summary.txt
  for model in uniform_velocity angular_velocities; do TE_QUEUE_BAR=$mode timeout -k 5 120 /tmp/scalar_abi_rate models/model_${model}_params.yaml 1 40 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt; done
done
