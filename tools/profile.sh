#!/bin/bash
# rocprofv3 passes for the bench (run on the GPU box through gpurun):
#   1. --kernel-trace --stats : per-kernel durations
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (HBM traffic; MI355X_MICROARCH.md 'HBM')
# usage: tools/profile.sh <tag> <bench args...>      outputs under gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py "$@" > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 bench.py "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -20 $OUT/fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 bench.py "$@" > $OUT/bench_write.json 2> $OUT/write.err || { tail -20 $OUT/write.err; exit 1; }
find $OUT -name "*.csv" | head -20
