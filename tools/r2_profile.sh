#!/bin/bash
# Evidence for profiles/: (1) rocprofv3 --kernel-trace --stats of the driver's bench command, (2) SQ counter passes on the
# fp64 angular kernels, (3) FETCH_SIZE / WRITE_SIZE passes for every workload the bench line reports.
set -o pipefail
OUT=$PWD/gpurun_out/r2prof
mkdir -p $OUT
export TMPDIR=/tmp
echo "== kernel trace of the default bench command" | tee $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace rc=$?" | tee -a $OUT/progress.txt
python3 tools/summarize_trace.py $OUT/trace $OUT/bench_trace.json > $OUT/bench_default_rocprofv3.txt 2>> $OUT/progress.txt
cp $OUT/trace/*kernel_stats.csv $OUT/bench_default_kernel_stats.csv 2>/dev/null || find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/bench_default_kernel_stats.csv \;
rm -rf $OUT/trace
head -60 $OUT/bench_default_rocprofv3.txt
for WL in ar4m64 av4m64 ar1m64; do
  echo "== sq counters $WL" | tee -a $OUT/progress.txt
  timeout -k 10 300 bash tools/pmc.sh r2prof_$WL $WL > $OUT/sq_$WL.txt 2>&1; echo "sq $WL rc=$?" | tee -a $OUT/progress.txt
  cat $OUT/sq_$WL.txt
done
echo "== traffic" | tee -a $OUT/progress.txt
timeout -k 10 1500 python3 tools/pmc_traffic.py --out $OUT/pmc > $OUT/pmc_traffic.txt 2>&1; echo "traffic rc=$?" | tee -a $OUT/progress.txt
tail -100 $OUT/pmc_traffic.txt
du -sh $OUT
