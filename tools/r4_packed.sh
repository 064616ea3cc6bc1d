#!/bin/bash
# the folded packed layout (angular_rates, 6 / 2 lanes per target): parity of everything that touches layouts, then its bench rows
set -o pipefail
OUT=$PWD/gpurun_out/r4packed
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_classes.py tests/test_gpu_by_id.py tests/test_gpu_intersection.py tests/test_gpu_mixed_configs.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2 --extra ar1m64_packed,ar1m_packed,av1m64_packed,ar1m64_full --side-file $OUT/side.json > $OUT/line.json 2> $OUT/err.txt
python - $OUT/side.json <<'PY'
import json, sys
for e in json.load(open(sys.argv[1])).get("extra", []):
    print("  %-16s %9.2f us/tick  frac %.3f  %d B/cycle" % (e["name"], 1e3 * e["ms_per_step"], e["roofline_frac"], e["algorithmic_bytes_per_cycle"]))
PY
