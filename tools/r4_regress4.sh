#!/bin/bash
# Round 4, item 1, fourth pass: is it the NUMBER of hardware queues of the process (a 5th queue shares a compute pipe with the 1st)?
# No resident session at all; GPU_MAX_HW_QUEUES = 8 lets the runtime give every stream a hardware queue of its own.
set -o pipefail
OUT=$PWD/gpurun_out/r4regress4
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2 --reps 1"
for q in 8 4 2; do
  tag=q$q
  GPU_MAX_HW_QUEUES=$q AMD_LOG_LEVEL=4 AMD_LOG_MASK=24 timeout -k 10 300 python3 bench.py $COMMON --extra cfg4,cfg4_64,cfg5 --side-file $OUT/$tag.json > $OUT/$tag.line 2> /tmp/$tag.log
  echo "$tag rc=$? lines $(wc -l < /tmp/$tag.log)" | tee -a $OUT/progress.txt
  grep -n "Created SWq" /tmp/$tag.log | cut -d']' -f2 | cut -c1-120 > $OUT/$tag.queues.txt
  echo "queues created: $(wc -l < $OUT/$tag.queues.txt); dispatches per hardware queue over the last 1500 log lines:" | tee -a $OUT/progress.txt
  tail -n 1500 /tmp/$tag.log | grep HWq | sed 's/.*\(HWq=0x[0-9a-f]*\).*/\1/' | sort | uniq -c | tee -a $OUT/progress.txt
  python3 - $OUT/$tag.json <<'PY' | tee -a $OUT/progress.txt
import json, sys
for e in json.load(open(sys.argv[1])).get("extra", []):
    print("  %-12s %8.2f us/tick" % (e["name"], 1e3 * e["ms_per_step"]))
PY
done
