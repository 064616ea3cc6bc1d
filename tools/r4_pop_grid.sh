#!/bin/bash
# Round 4: the population tick's workgroup shape at the configs[3] / configs[4] shares (launch-bound: 2 x 977 wavefronts).
set -o pipefail
OUT=$PWD/gpurun_out/r4popgrid
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2 --reps 1"
printf "%-44s %8s %8s %8s\n" "variant" cfg4 cfg4_64 cfg5 | tee $OUT/summary.txt
run() {
  env $2 timeout -k 10 300 python3 bench.py $COMMON --extra cfg4,cfg4_64,cfg5 --side-file $OUT/$1.json > $OUT/$1.line 2> $OUT/$1.err
  python3 - $OUT/$1.json "$3" <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = {e["name"]: e for e in json.load(open(sys.argv[1])).get("extra", [])}
def us(n): return ("%8.2f" % (1e3 * d[n]["ms_per_step"])) if n in d and "ms_per_step" in d[n] else "   error"
print("%-44s %s %s %s" % (sys.argv[2], us("cfg4"), us("cfg4_64"), us("cfg5")))
PY
}
run a "TE_X=0" "population tick, 1 wave per workgroup (default)"
run b "TE_SMALL_GRID_WAVES=512" "population tick, 4 waves per workgroup"
run c "TE_POPULATION_TICK=0" "one launch per batch, two graph branches"
run d "TE_X=0" "population tick, default (again)"
run e "TE_POP_EXPERIMENT=1" "population tick, light parts first"
run f "TE_POP_EXPERIMENT=2" "population tick, two parts interleaved"
run g "TE_X=0" "population tick, default (third time)"
