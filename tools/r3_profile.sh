#!/bin/bash
# Round-3 evidence for profiles/: (1) rocprofv3 --kernel-trace --stats of the driver's bench command; (2) ONE
# `rocprofv3 --kernel-trace --pmc FETCH_SIZE` pass over the WHOLE default bench process (round 2: exit 139 inside a torch
# elementwise launch of the stream set-up; the set-up no longer launches torch kernels).  bench.py keeps /proc/self/maps
# current on disk (TE_BENCH_MAPS) so that the PCs of an abort can be resolved afterwards.  The pass is taken once, not looped.
set -o pipefail
OUT=$PWD/gpurun_out/r3prof
mkdir -p $OUT
export TMPDIR=/tmp
echo "== kernel trace of the default bench command" | tee $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --side-file $OUT/bench_extra_trace.json > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace rc=$?" | tee -a $OUT/progress.txt
python3 tools/summarize_trace.py $OUT/trace $OUT/bench_trace.json > $OUT/bench_default_rocprofv3.txt 2>> $OUT/progress.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/bench_default_kernel_stats.csv \;
rm -rf $OUT/trace
head -40 $OUT/bench_default_rocprofv3.txt
echo "== all-in-one PMC pass (once)" | tee -a $OUT/progress.txt
TE_BENCH_MAPS=$OUT/allinone_maps.txt timeout -k 10 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/allinone -o allinone -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --side-file $OUT/bench_extra_allinone.json > $OUT/allinone.json 2> $OUT/allinone.err
echo "all-in-one rc=$?" | tee -a $OUT/progress.txt
cat $OUT/allinone_maps.txt.progress 2>/dev/null | tail -3
grep -v "^W20\|^I20" $OUT/allinone.err | tail -60 > $OUT/allinone_stderr_tail.txt
python3 - <<PY >> $OUT/progress.txt 2>&1
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("$OUT/allinone/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "kf_step" in r["Kernel_Name"]:
            rows[(r["Kernel_Name"].split("(")[0][:90], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
print("all-in-one FETCH_SIZE (KB, raw) per step kernel and grid: launches, mean")
for k, v in sorted(rows.items()):
    print("%-92s grid %-9s n=%5d mean=%12.1f" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
find $OUT/allinone -name "*counter_collection.csv" -size +8M -delete
du -sh $OUT
tail -30 $OUT/progress.txt
