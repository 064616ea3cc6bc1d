#!/bin/bash
# Round 4: the resident mode's doorbell in device memory behind the PCIe BAR against the one in host memory (TE_LIVE_DOORBELL=host): live tests, then
# back-to-back / paced figures through Python and the paced round trip from C++ (tools/live_latency.cpp).
set -o pipefail
OUT=$PWD/gpurun_out/r4doorbell
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_live.py tests/test_gpu_c_abi_program.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 5 120 tools/_build/bar_doorbell | tee $OUT/echo.txt
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2 --reps 1"
for mode in bar host bar host; do
  TE_LIVE_DOORBELL=$mode timeout -k 10 300 python3 bench.py $COMMON --extra cfg2_live,cfg3_live,cfg4_live,cfg4_64_live,cfg5_live --side-file $OUT/$mode.json > $OUT/$mode.line 2> $OUT/$mode.err
  python3 - $OUT/$mode.json $mode <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = {e["name"]: e for e in json.load(open(sys.argv[1])).get("extra", [])}
print("%-5s" % sys.argv[2], "  ".join("%s %.2f / %.2f" % (n, d[n]["live"]["us_per_tick_back_to_back"], d[n]["live"]["us_per_tick_paced"]) for n in ("cfg2_live", "cfg3_live", "cfg4_live", "cfg4_64_live", "cfg5_live") if n in d and "live" in d[n]))
PY
done
if [ -f tools/live_latency.cpp ]; then
  hipcc --offload-arch=gfx950 -O2 -w -I include/target_estimation_amd tools/live_latency.cpp -o /tmp/live_latency -L target_estimation_amd/lib -ltarget_estimation_amd -Wl,-rpath,$PWD/target_estimation_amd/lib 2> $OUT/ll_build.txt
  for mode in bar host; do echo "== C++ paced round trip, doorbell: $mode" | tee -a $OUT/summary.txt; for args in "models/model_uniform_velocity_params.yaml 10000" "models/model_uniform_acceleration_params.yaml 100000 f32" "models/model_angular_rates_params.yaml 62500 f32"; do TE_LIVE_DOORBELL=$mode timeout -k 5 100 /tmp/live_latency $args 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt; done; true 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt; done
fi
