#!/bin/bash
# SQ counter passes for one bench workload: tools/pmc.sh <tag> <workload> [extra bench args]
# (program directly behind `--`, counters only together with --kernel-trace; two passes of <= 8 SQ counters)
set -e
TAG=$1; WL=$2; shift 2
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--workload $WL --steps 20 --warmup 4 --reps 1 --no-cpu --no-gather --extra '' --launch-mode sequence"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -o sq1 -- python3 bench.py --workload $WL --steps 20 --warmup 4 --reps 1 --no-cpu --no-gather --extra "" --launch-mode sequence "$@" > $OUT/b1.json 2> $OUT/e1.txt || { tail $OUT/e1.txt; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -o sq2 -- python3 bench.py --workload $WL --steps 20 --warmup 4 --reps 1 --no-cpu --no-gather --extra "" --launch-mode sequence "$@" > $OUT/b2.json 2> $OUT/e2.txt || { tail $OUT/e2.txt; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq1", "sq2"):
    f = glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    rows = []
    rd = csv.DictReader(open(f))
    for r in rd:
        if "kf_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            rows.append(r)
    for k, v in sorted(acc.items()):
        v = v[len(v) // 4:]
        print("%-24s avg/launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
    with open(out + "/" + sub + "_counter_collection.csv", "w", newline="") as g:   # the step kernels' rows only
        w = csv.DictWriter(g, fieldnames=rd.fieldnames)
        w.writeheader()
        w.writerows(rows)
PY
rm -rf $OUT/sq1 $OUT/sq2
