#!/bin/bash
# gpu tests + the driver-style bench line: gpurun -- bash tools/gpu_check.sh <tag>   (outputs under gpurun_out/<tag>)
set -o pipefail
OUT=$PWD/gpurun_out/${1:-r2b}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -15 $OUT/tests.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench.err
python tools/show_bench.py $OUT/bench.json
