#!/bin/bash
# Round 4, VERDICT item 4: what do the release / acquire edges of the resident mode's hand-offs cost?  TE_LIVE_FLAGS switches single
# edges off (csrc/kf_step.hpp kLive*): 0 = all edges in force, 16 = all relaxed (round 3), the other masks = one edge at a time.
set -o pipefail
OUT=$PWD/gpurun_out/r4liveord
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-gather --workload cfg2 --reps 1"
printf "%-8s %-58s %10s %10s %10s\n" flags "edges in force" cfg2_live cfg3_live cfg4_live | tee $OUT/summary.txt
run() {
  TE_LIVE_FLAGS=$1 timeout -k 10 300 python3 bench.py $COMMON --extra cfg2_live,cfg3_live,cfg4_live --side-file $OUT/f$1.json > $OUT/f$1.line 2> $OUT/f$1.err
  python3 - $OUT/f$1.json "$1" "$2" <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = {e["name"]: e for e in json.load(open(sys.argv[1])).get("extra", [])}
def us(n): return ("%10.2f" % (1e3 * d[n]["ms_per_step"])) if n in d and "ms_per_step" in d[n] else "     error"
print("%-8s %-58s %s %s %s" % (sys.argv[2], sys.argv[3], us("cfg2_live"), us("cfg3_live"), us("cfg4_live")))
PY
}
run 16  "none (round 3: relaxed everywhere)"
run 0   "all"
run 992 "none, through the single switches (32+64+128+256+512)"
run 960 "worker acquire fence only"
run 928 "worker progress release only"
run 864 "relay acquire fence only"
run 736 "relay done release only"
run 480 "relay mirror release fence only"
