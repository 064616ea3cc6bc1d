#!/bin/bash
# Copy what tools/r2_profile.sh left under gpurun_out/ into profiles/ (the tracked, judged copies): the kernel-trace summary
# of the default bench command, the FETCH_SIZE / WRITE_SIZE passes per workload (step-kernel rows + each pass's stderr) and the
# SQ counter passes.  Run here, after gpurun merged the outputs back.
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r2prof
rm -rf profiles/r02_pmc; mkdir -p profiles/r02_pmc
cp $S/pmc/hbm_traffic.json profiles/hbm_traffic.json
for d in $S/pmc/*/; do
  w=$(basename $d); mkdir -p profiles/r02_pmc/$w
  cp $d/*_counter_collection.csv profiles/r02_pmc/$w/ 2>/dev/null || true
  for c in FETCH_SIZE WRITE_SIZE; do grep -v "^W20\|^I20" $d/$c.stderr.txt > profiles/r02_pmc/$w/$c.stderr.txt 2>/dev/null || true; done
done
cp $S/pmc_traffic.txt profiles/r02_pmc_traffic_summary.txt
cp $S/bench_default_rocprofv3.txt profiles/r02_bench_default_rocprofv3.txt
cp $S/bench_default_kernel_stats.csv profiles/r02_bench_default_kernel_stats.csv
grep '^{' $S/bench_trace.json > profiles/r02_bench_line_under_kernel_trace.json
for w in ar4m64 av4m64 ar1m64; do
  mkdir -p profiles/r02_sq/$w
  cp gpurun_out/pmc_r2prof_$w/sq*_counter_collection.csv profiles/r02_sq/$w/
  cp $S/sq_$w.txt profiles/r02_sq/$w/summary.txt
done
ls profiles/r02_pmc | wc -l
