#!/bin/bash
# SQ counter passes (raw rows kept), the whole default bench under one --pmc process (round-1 SIGSEGV check), tests + bench
set -o pipefail
OUT=$PWD/gpurun_out/r2g
mkdir -p $OUT
export TMPDIR=/tmp
for WL in ar4m64 av4m64 ar1m64 ar8m; do
  timeout -k 10 300 bash tools/pmc.sh r2g_$WL $WL > $OUT/sq_$WL.txt 2>&1; echo "sq $WL rc=$?"
  cat $OUT/sq_$WL.txt
done
echo "== whole default bench under --pmc FETCH_SIZE (one process, every extra workload)"
timeout -k 10 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/allpmc -o all -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/allpmc_bench.json 2> $OUT/allpmc_stderr.txt
echo "all-in-one pmc rc=$?"
tail -5 $OUT/allpmc_stderr.txt
ls -la $OUT/allpmc 2>/dev/null | head; find $OUT/allpmc -name "*counter_collection.csv" -exec wc -l {} \;
rm -rf $OUT/allpmc
bash tools/r2_run_b.sh r2g
