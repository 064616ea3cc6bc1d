#!/bin/bash
# Round-4 evidence at HEAD, one gpurun call: (1) the default bench line + side file, (2) rocprofv3 --kernel-trace --stats of the same command.
# (3) FETCH_SIZE / WRITE_SIZE passes for every workload of the line run as separate gpurun calls (a call is limited to 20 minutes):
#     python3 tools/pmc_traffic.py --out gpurun_out/r4final/pmcA --workloads ...     (tools/collect_r4.sh copies everything into profiles/)
set -o pipefail
OUT=$PWD/gpurun_out/r4final
mkdir -p $OUT
export TMPDIR=/tmp
echo "== default bench" | tee $OUT/progress.txt
timeout -k 10 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 --side-file $OUT/bench_extra.json > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/progress.txt
tail -c 1500 $OUT/bench_line.json
echo "== kernel trace" | tee -a $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --side-file $OUT/bench_extra_trace.json > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace rc=$?" | tee -a $OUT/progress.txt
python3 tools/summarize_trace.py $OUT/trace $OUT/bench_trace.json $OUT/bench_extra_trace.json > $OUT/bench_default_rocprofv3.txt 2>> $OUT/progress.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/bench_default_kernel_stats.csv \;
rm -rf $OUT/trace
echo "== resident-mode capacity" | tee -a $OUT/progress.txt
{ echo "# python3 tools/live_capacity.py   -- what the library allows at round 4 (occupancy query incl. static LDS, at most 5 resident wavefronts per SIMD, one per CU in hand)."
  echo "# Round 3 (profiles/r03_live_capacity.txt): angular_velocities / angular_rates f64 49152 / 49152 plain and with outputs; the other rows as here."
  python3 tools/live_capacity.py 2>/dev/null | grep -E "^(uniform|angular)"; } > $OUT/live_capacity.txt
cat $OUT/live_capacity.txt
head -45 $OUT/bench_default_rocprofv3.txt | cut -c1-200
du -sh $OUT
