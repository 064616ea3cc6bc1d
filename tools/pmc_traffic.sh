#!/bin/bash
# HBM traffic of the step kernel of single workloads: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
# (MI355X_MICROARCH.md 'HBM').  usage: tools/pmc_traffic.sh <tag> <workload> [<workload> ...]
# outputs gpurun_out/pmc_traffic_<tag>/<workload>/{fetch,write}/... and a summary on stdout
TAG=$1; shift
export TMPDIR=/tmp
for WL in "$@"; do
  OUT=$PWD/gpurun_out/pmc_traffic_$TAG/$WL
  mkdir -p $OUT
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o $C -- python3 bench.py --workload $WL --steps 40 --warmup 8 --no-cpu --extra "" --launch-mode sequence > $OUT/bench_$C.json 2> $OUT/$C.err || { echo "$WL $C FAILED"; tail -3 $OUT/$C.err; }
  done
  python3 - $OUT $WL <<'PY'
import csv, glob, sys
out, wl = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(out + "/" + c + "/*counter_collection.csv")
    if not fs:
        continue
    vals = {}
    for r in csv.DictReader(open(fs[0])):
        if "kf_step" in r["Kernel_Name"] and r["Counter_Name"] == c:
            vals.setdefault((r["Kernel_Name"].split("(")[0], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        res.setdefault(k, {})[c] = sum(v) / len(v)
for (k, g), d in res.items():
    rd = 2 * 1024 * d.get("FETCH_SIZE", float("nan")); wr = 1024 * d.get("WRITE_SIZE", float("nan"))
    print("%s %s grid %s: hbm_read %.3f MB (2 x FETCH_SIZE x 1024)  hbm_write %.3f MB" % (wl, k.replace("void te::", ""), g, rd / 1e6, wr / 1e6))
PY
done
