#!/bin/bash
# nt-measurement experiment + dense layout sweep
set -o pipefail
OUT=$PWD/gpurun_out/r2f
mkdir -p $OUT
E="ar1m64,av1m64,ar4m64,uv10m,cfg4_4m,ar1m"
for NT in 0 1; do
  TE_NT_MEAS=$NT timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-gather --extra $E > $OUT/bench_nt$NT.json 2> $OUT/bench_nt$NT.err
  echo "== TE_NT_MEAS=$NT"; python tools/show_bench.py $OUT/bench_nt$NT.json | grep -v parity
done
timeout -k 10 900 python tools/sweep.py --steps 100 --sizes 1000000 > $OUT/sweep.txt 2>&1; echo "sweep rc=$?"
cat $OUT/sweep.txt
