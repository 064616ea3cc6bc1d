#!/bin/bash
# rocprofv3 passes for the bench (run on the GPU box through gpurun):
#   1. --kernel-trace --stats : per-kernel durations
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, one workload per process (tools/pmc_traffic.sh)
# usage: tools/profile.sh <tag> <bench args...>      outputs under gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py "$@" > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
# The counter passes run one workload per process (tools/pmc_traffic.sh): with every extra workload in one
# process (graphs with concurrent branches, 1 GiB torch copies, ...) the rocprofv3 counter service of this image
# segfaults in one of its own threads.
bash tools/pmc_traffic.sh $TAG cfg2 cfg3 uv1m ua1m av1m ar1m cfg2_full uv1m_full uv1m_packed ar1m_full ar1m_packed > $OUT/pmc_traffic.txt 2>&1 || true
cat $OUT/pmc_traffic.txt
find $OUT -name "*.csv" | head -20
