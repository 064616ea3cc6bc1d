#!/bin/bash
# Round 4, item 1, second pass: WHAT about the high-priority stream slows later two-branch graph ticks down?
#   F  no resident session at all, but a high-priority stream created (through torch) before the mixed shares
#   G  cfg2_live ahead, resident stream at the LOWEST priority
#   H  cfg2_live ahead, high priority, the stream (and its hardware queue) destroyed when the session stops
set -o pipefail
OUT=$PWD/gpurun_out/r4regress2
mkdir -p $OUT
export TMPDIR=/tmp
show() { python3 - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for e in d.get("extra", []):
    if "error" in e: print("  %-12s ERROR %s" % (e["name"], e["error"][:200]))
    else: print("  %-12s %8.2f us/tick  %s" % (e["name"], 1e3 * e["ms_per_step"], e["launch_mode"][:40]))
PY
}
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2"
echo "== F: a high-priority stream exists (never used), no resident session" | tee $OUT/progress.txt
timeout -k 10 300 python3 -c "
import sys, torch
s = torch.cuda.Stream(priority=-1)
print('priority stream', s, file=sys.stderr)
sys.argv = ['bench.py'] + '$COMMON --extra cfg4,cfg4_64,cfg5 --side-file $OUT/F.json'.split()
import bench
bench.main()
" > $OUT/F.line 2> $OUT/F.err; echo "rc=$?" | tee -a $OUT/progress.txt
show $OUT/F.json | tee -a $OUT/progress.txt
echo "== G: cfg2_live ahead, resident stream at the lowest priority" | tee -a $OUT/progress.txt
TE_LIVE_STREAM_PRIORITY=low timeout -k 10 300 python3 bench.py $COMMON --extra cfg2_live,cfg4,cfg4_64,cfg5 --side-file $OUT/G.json > $OUT/G.line 2> $OUT/G.err; echo "rc=$?" | tee -a $OUT/progress.txt
show $OUT/G.json | tee -a $OUT/progress.txt
echo "== H: cfg2_live ahead, high priority, stream destroyed at live_stop" | tee -a $OUT/progress.txt
TE_LIVE_STREAM_KEEP=0 timeout -k 10 300 python3 bench.py $COMMON --extra cfg2_live,cfg4,cfg4_64,cfg5 --side-file $OUT/H.json > $OUT/H.line 2> $OUT/H.err; echo "rc=$?" | tee -a $OUT/progress.txt
show $OUT/H.json | tee -a $OUT/progress.txt
