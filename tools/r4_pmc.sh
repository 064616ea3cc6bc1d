#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes for a list of bench workloads: gpurun -- bash tools/r4_pmc.sh <part> <comma list>
set -o pipefail
OUT=$PWD/gpurun_out/r4final
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/pmc_traffic.py --out $OUT/$1 --workloads "$2" > $OUT/$1.txt 2>&1
echo "rc=$?"
tail -45 $OUT/$1.txt | cut -c1-190
