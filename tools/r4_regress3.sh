#!/bin/bash
# Round 4, item 1, third pass: which hardware queues do the two graph branches use, with and without a resident session before them?
# (runtime log instead of the profiler: under rocprofv3 the slow case does not reproduce)
set -o pipefail
OUT=$PWD/gpurun_out/r4regress3
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2 --reps 1"
for tag in bad good; do
  if [ $tag = bad ]; then EX=cfg2_live,cfg4_64; else EX=cfg4_64; fi
  AMD_LOG_LEVEL=4 AMD_LOG_MASK=24 timeout -k 10 300 python3 bench.py $COMMON --extra $EX --side-file $OUT/$tag.json > $OUT/$tag.line 2> /tmp/$tag.log
  echo "$tag rc=$? lines $(wc -l < /tmp/$tag.log)" | tee -a $OUT/progress.txt
  grep -n "hardware queues\|acquireQueue\|Created SWq\|created hardware queue\|hsa_queue_create\|priority" /tmp/$tag.log | tail -n 200 > $OUT/$tag.queues.txt
  tail -n 1500 /tmp/$tag.log | cut -c1-400 > $OUT/$tag.tail.txt
  python3 - $OUT/$tag.json <<'PY' | tee -a $OUT/progress.txt
import json, sys
for e in json.load(open(sys.argv[1])).get("extra", []):
    print("  %-12s %8.2f us/tick" % (e["name"], 1e3 * e["ms_per_step"]))
PY
done
