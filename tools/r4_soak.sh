#!/bin/bash
# Round 4, long checks against the oracle at HEAD in one gpurun call: the new population tests, the headline population EVERY target over 200 ticks (fp64)
# and 500 ticks (fp32), extreme inputs, randomised schedules, one-target call sequences.
set -o pipefail
OUT=$PWD/gpurun_out/r4soak
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_mixed_configs.py tests/test_gpu_intersection.py -m gpu -x -q > $OUT/mixed.log 2>&1; echo "mixed rc=$?"; tail -3 $OUT/mixed.log
TE_SOAK_TICKS=200 TE_SOAK_OUT=$OUT/soak_cfg4_1gpu.json timeout -k 10 900 python -m pytest tests/test_gpu_soak.py -m gpu -q -s > $OUT/soak64.log 2>&1; echo "soak64 rc=$?"; tail -3 $OUT/soak64.log
TE_SOAK_TICKS=500 TE_SOAK_WL=cfg4_1gpu32 TE_SOAK_OUT=$OUT/soak_cfg4_1gpu32.json timeout -k 10 900 python -m pytest tests/test_gpu_soak.py -m gpu -q -s > $OUT/soak32.log 2>&1; echo "soak32 rc=$?"; tail -3 $OUT/soak32.log
timeout -k 10 600 python tests/extended/torture.py > $OUT/torture.txt 2>&1; echo "torture rc=$?"; tail -4 $OUT/torture.txt | cut -c1-200
timeout -k 10 900 python tests/extended/soak_schedules.py 60 > $OUT/schedules.txt 2>&1; echo "schedules rc=$?"; tail -2 $OUT/schedules.txt
timeout -k 10 900 python tests/extended/soak_one_target.py 150 > $OUT/one_target.txt 2>&1; echo "one-target rc=$?"; tail -2 $OUT/one_target.txt
