#!/bin/bash
# gpu tests + the driver-style bench line (everything else in the side file): gpurun -- bash tools/r4_check.sh <tag>
set -o pipefail
OUT=$PWD/gpurun_out/${1:-r4a}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -15 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --side-file $OUT/bench_extra.json > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 600 $OUT/bench.err
python - $OUT/bench_extra.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps(d["line"])[:700])
for e in d.get("extra", []):
    if e.get("name") == "gather_pose": print("  gather_pose: exposed %.3f ms, overlapped %.3f ms" % (e.get("gather_pose_ms_exposed", float("nan")), e.get("gather_pose_ms_overlapped", float("nan")))); continue
    if "error" in e: print("  %-18s ERROR %s" % (e["name"], e["error"][:160]))
    else: print("  %-18s %-3s %9d  %8.2f us/tick  frac %s  %s" % (e["name"], e["dtype"], e["targets_per_gpu"], 1e3 * e["ms_per_step"],
                ("%.3f" % e["roofline_frac"]) if e.get("roofline_frac") is not None else "  -  ", e["launch_mode"][:50]))
PY
