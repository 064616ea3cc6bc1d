#!/bin/bash
# Round 4, item 1: what slowed the two-branch graph ticks (cfg4 / cfg4_64 / cfg5) down 2-3x?  Discriminating runs in one gpurun call:
#   A  the three mixed shares alone                                  (no resident session before them in the process)
#   B  the same with cfg2_live ahead                                 (the resident kernel's high-priority stream exists)
#   C  B with TE_LIVE_STREAM_PRIORITY=0                              (the resident kernel's stream at normal priority)
#   D  rocprofv3 --kernel-trace of B for cfg4_64 only -> tools/overlap.py (queue ids, overlap of the two branches)
#   E  the same trace for A
set -o pipefail
OUT=$PWD/gpurun_out/r4regress
mkdir -p $OUT
export TMPDIR=/tmp
show() { python3 - "$1" <<'EOF'
import json, sys
d = json.load(open(sys.argv[1]))
for e in d.get("extra", []):
    if "error" in e: print("  %-12s ERROR %s" % (e["name"], e["error"][:200]))
    else: print("  %-12s %8.2f us/tick  frac %.3f  %s" % (e["name"], 1e3 * e["ms_per_step"], e["roofline_frac"], e["launch_mode"]))
EOF
}
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2"
echo "== A: mixed shares alone" | tee $OUT/progress.txt
timeout -k 10 300 python3 bench.py $COMMON --extra cfg4,cfg4_64,cfg5 --side-file $OUT/A.json > $OUT/A.line 2> $OUT/A.err; echo "rc=$?" | tee -a $OUT/progress.txt
show $OUT/A.json | tee -a $OUT/progress.txt
echo "== B: cfg2_live ahead" | tee -a $OUT/progress.txt
timeout -k 10 300 python3 bench.py $COMMON --extra cfg2_live,cfg4,cfg4_64,cfg5 --side-file $OUT/B.json > $OUT/B.line 2> $OUT/B.err; echo "rc=$?" | tee -a $OUT/progress.txt
show $OUT/B.json | tee -a $OUT/progress.txt
echo "== C: cfg2_live ahead, resident stream at normal priority" | tee -a $OUT/progress.txt
TE_LIVE_STREAM_PRIORITY=0 timeout -k 10 300 python3 bench.py $COMMON --extra cfg2_live,cfg4,cfg4_64,cfg5 --side-file $OUT/C.json > $OUT/C.line 2> $OUT/C.err; echo "rc=$?" | tee -a $OUT/progress.txt
show $OUT/C.json | tee -a $OUT/progress.txt
echo "== D: kernel trace of B (cfg2_live, cfg4_64)" | tee -a $OUT/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/traceD -o t -- python3 bench.py $COMMON --extra cfg2_live,cfg4_64 --side-file $OUT/D.json > $OUT/D.line 2> $OUT/D.err; echo "rc=$?" | tee -a $OUT/progress.txt
python3 tools/overlap.py $OUT/traceD 2>&1 | tee $OUT/D_overlap.txt | tee -a $OUT/progress.txt
echo "== E: kernel trace of A (cfg4_64)" | tee -a $OUT/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/traceE -o t -- python3 bench.py $COMMON --extra cfg4_64 --side-file $OUT/E.json > $OUT/E.line 2> $OUT/E.err; echo "rc=$?" | tee -a $OUT/progress.txt
python3 tools/overlap.py $OUT/traceE 2>&1 | tee $OUT/E_overlap.txt | tee -a $OUT/progress.txt
# keep only the step-kernel rows of the traces (the CSVs are large)
for t in D E; do
  f=$(find $OUT/trace$t -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep kf_step $f | tail -400) > $OUT/trace${t}_tail.csv
  rm -rf $OUT/trace$t
done
du -sh $OUT
