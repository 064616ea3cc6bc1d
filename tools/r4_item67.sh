#!/bin/bash
# Round 4, VERDICT items 6 and 7 in one gpurun call:
#  (7) north_star's literal form -- one wavefront per target, P in LDS (tools/wave_per_target.hip) -- timed at 10^6 angular-rates targets fp64, with
#      FETCH_SIZE / WRITE_SIZE passes, next to the shipped dense forms of the same case (lanes 6 full, 106 packed) and the default (301)
#  (6) SQ counters of the coupled-matrix fallbacks (ar1m64_packed, av1m64_packed) and of the default kernel for comparison
set -o pipefail
OUT=$PWD/gpurun_out/r4item67
mkdir -p $OUT
export TMPDIR=/tmp
[ -x tools/_build/wave_per_target ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wave_per_target.hip -o tools/_build/wave_per_target
echo "== wave per target" | tee $OUT/progress.txt
timeout -k 10 120 tools/_build/wave_per_target 1000000 20 > $OUT/wave_row.txt 2> $OUT/wave_check.txt; echo "rc=$?" | tee -a $OUT/progress.txt
cat $OUT/wave_check.txt $OUT/wave_row.txt | tee -a $OUT/progress.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/wave_$c -o w -- tools/_build/wave_per_target 1000000 10 > /dev/null 2> $OUT/wave_$c.err; echo "$c rc=$?" | tee -a $OUT/progress.txt
  python3 - $OUT/wave_$c $c <<'PY' | tee -a $OUT/progress.txt
import csv, glob, sys
v = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_wave_per_target" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
            v.append(float(r["Counter_Value"]))
v = v[len(v) // 2:]
if v:
    kb = sum(v) / len(v)
    print("%s avg/launch %.0f KB -> %.1f MB per tick (%s)" % (sys.argv[2], kb, kb * 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1) / 1e6,
          "x2: the guide's gfx950 correction for wide coalesced reads" if sys.argv[2] == "FETCH_SIZE" else "as counted"))
PY
  rm -rf $OUT/wave_$c
done
echo "== the shipped forms of the same case" | tee -a $OUT/progress.txt
python3 tools/sweep.py --steps 100 --sizes 1000000 --models angular_rates --dtypes f64 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_ar_f64.txt | tee -a $OUT/progress.txt
echo "== SQ counters" | tee -a $OUT/progress.txt
for wl in ar1m64_packed av1m64_packed ar1m64 av1m64; do
  echo "-- $wl" | tee -a $OUT/progress.txt
  bash tools/pmc.sh r4_$wl $wl 2>&1 | grep -v amdgpu.ids | tee $OUT/sq_$wl.txt | tee -a $OUT/progress.txt
done
