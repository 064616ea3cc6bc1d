#!/bin/bash
# Round 4: from how many wavefronts should a dense launch use 4 wavefronts per workgroup instead of 1?  (TE_SMALL_GRID_WAVES, default 1024)
# configs[2] = 100 000 UA fp32 = 1563 wavefronts sits just above the default: 391 workgroups of 4 over 256 CUs is a 2 : 1 imbalance.
set -o pipefail
OUT=$PWD/gpurun_out/r4smallgrid
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2 --reps 1"
printf "%-28s %8s %12s %14s %10s\n" "TE_SMALL_GRID_WAVES" cfg3 cfg3_stream ar100k64_1kcls cfg4_64 | tee $OUT/summary.txt
for g in 1024 2048 4096 8192 1024; do
  TE_SMALL_GRID_WAVES=$g timeout -k 10 300 python3 bench.py $COMMON --extra cfg3,cfg3_stream,ar100k64_1kcls,cfg4_64 --side-file $OUT/g$g.json > $OUT/g$g.line 2> $OUT/g$g.err
  python3 - $OUT/g$g.json $g <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = {e["name"]: e for e in json.load(open(sys.argv[1])).get("extra", [])}
def us(n): return ("%.2f" % (1e3 * d[n]["ms_per_step"])) if n in d and "ms_per_step" in d[n] else "error"
print("%-28s %8s %12s %14s %10s" % (sys.argv[2], us("cfg3"), us("cfg3_stream"), us("ar100k64_1kcls"), us("cfg4_64")))
PY
done
