#!/bin/bash
# L2 memory-side (TCC / EA) counter passes for one bench workload: tools/tcc.sh <tag> <workload> [extra bench args]
# Three passes of <= 4 TCC counters (the block has 4 slots), the program directly behind `--`, counters only with
# --kernel-trace.  What they say about a streaming kernel:
#   TCC_EA0_RDREQ / WRREQ(_sum)            requests the L2 sent to the fabric (Infinity Cache / HBM side)
#   TCC_EA0_RDREQ_LEVEL / WRREQ_LEVEL      requests in flight, summed over cycles: LEVEL / REQ = average latency in L2 clocks
#   TCC_EA0_*_DRAM_CREDIT_STALL, WRREQ_STALL   cycles a request was ready but the memory side had no credit for it = back-pressure
#   TCC_CYCLE, TCC_BUSY, TCC_TAG_STALL     L2 clocks, clocks with work, clocks the tag pipeline was stalled
set -e
TAG=$1; WL=$2; shift 2
OUT=$PWD/gpurun_out/tcc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o $1 -- python3 bench.py --workload $WL --steps 20 --warmup 4 --reps 1 --no-cpu --no-gather --extra "" --launch-mode sequence "${@:3}" > $OUT/$1.json 2> $OUT/$1.err || { tail $OUT/$1.err; exit 1; }
}
run p1 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" "$@"
run p2 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_CYCLE_sum" "$@"
run p3 "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum" "$@"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("p1", "p2", "p3"):
    f = glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    rd = csv.DictReader(open(f))
    rows = []
    for r in rd:
        if "kf_step" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("void te::", "")[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
            rows.append(r)
    for k, v in sorted(acc.items()):
        v = v[len(v) // 4:]
        print("%-72s %-40s avg/launch %.5g  (n=%d)" % (k[0], k[1], sum(v) / len(v), len(v)))
    with open(out + "/" + sub + "_counter_collection.csv", "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=rd.fieldnames)
        w.writeheader()
        w.writerows(rows)
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
