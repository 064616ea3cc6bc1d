#!/bin/bash
# live tests + the small-population bench rows: gpurun -- bash tools/r4_check2.sh <tag>
set -o pipefail
OUT=$PWD/gpurun_out/${1:-r4d}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_live.py tests/test_highprec_kat.py tests/test_gpu_mixed_configs.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -8 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/r4_pop_grid.sh
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2 --extra cfg2_live,cfg3_live,cfg4_live,cfg4_64_live,cfg5_live --side-file $OUT/live.json > $OUT/live.line 2> $OUT/live.err
python - $OUT/live.json <<'PY'
import json, sys
for e in json.load(open(sys.argv[1])).get("extra", []):
    if "error" in e: print("  %-14s ERROR %s" % (e["name"], e["error"][:180]))
    else: print("  %-14s %8.2f us/tick back to back   paced %s" % (e["name"], 1e3 * e["ms_per_step"], e.get("live", {}).get("us_per_tick_paced")))
PY
python tools/live_capacity.py 2>&1 | tail -12
