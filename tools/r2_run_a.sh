#!/bin/bash
# round-2 GPU call A: gpu tests, the driver-style bench line, SQ counter passes (fp64 angular kernels + fp32 twin), cache-policy experiment
set -o pipefail
OUT=$PWD/gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests" | tee $OUT/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/progress.txt
tail -5 $OUT/tests.log
echo "== bench" | tee -a $OUT/progress.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/progress.txt
tail -c 600 $OUT/bench.err
echo "== nt experiment" | tee -a $OUT/progress.txt
timeout -k 10 120 tools/_build/record_copy_nt > $OUT/record_copy_nt.txt 2>&1; echo "nt rc=$?" | tee -a $OUT/progress.txt
for WL in ar1m64 av1m64 ar1m; do
  echo "== pmc $WL" | tee -a $OUT/progress.txt
  timeout -k 10 300 bash tools/pmc.sh r2a_$WL $WL > $OUT/pmc_$WL.txt 2>&1; echo "pmc $WL rc=$?" | tee -a $OUT/progress.txt
done
cat $OUT/record_copy_nt.txt | head -40
for WL in ar1m64 av1m64 ar1m; do echo "-- $WL"; cat $OUT/pmc_$WL.txt; done
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2a/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["tick"]["frac"])
print(d["config"]["timing"])
for k in d["roofline"]["kernels"]: print(k)
print("cpu", d.get("cpu_baseline"))
for e in d.get("extra", []):
    print(e.get("name"), e.get("error") or "%.3g c/s  %.4f ms  frac %.3f  %s" % (e["cycles_per_s"], e["ms_per_step"], e["roofline_frac"], e["residency"][:12]))
PY
