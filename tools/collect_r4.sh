#!/bin/bash
# Copy what tools/r4_final.sh and the two tools/r4_pmc.sh calls left under gpurun_out/r4final into profiles/ (tracked), and refresh DESIGN.md's blocks.
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r4final
python3 - <<PY
import json
a = json.load(open("$S/pmcA/hbm_traffic.json")); b = json.load(open("$S/pmcB/hbm_traffic.json"))
assert a["_meta"]["commit"] == b["_meta"]["commit"], (a["_meta"], b["_meta"])
a.update({k: v for k, v in b.items() if k != "_meta"})
json.dump(a, open("profiles/hbm_traffic.json", "w"), indent=1, sort_keys=True)
print("workloads:", len(a) - 1, "commit", a["_meta"]["commit"])
PY
rm -rf profiles/r04_pmc; mkdir -p profiles/r04_pmc
for part in pmcA pmcB; do
  for d in $S/$part/*/; do
    w=$(basename $d); mkdir -p profiles/r04_pmc/$w
    cp $d/*_counter_collection.csv profiles/r04_pmc/$w/ 2>/dev/null || true
    for c in FETCH_SIZE WRITE_SIZE; do grep -v "^W20\|^I20\|^E20" $d/$c.stderr.txt | tail -40 > profiles/r04_pmc/$w/$c.stderr.txt 2>/dev/null || true; done
  done
done
cat $S/pmcA.txt $S/pmcB.txt | grep -v "^$" > profiles/r04_pmc_traffic_summary.txt
cp $S/bench_default_rocprofv3.txt profiles/r04_bench_default_rocprofv3.txt
cp $S/bench_default_kernel_stats.csv profiles/r04_bench_default_kernel_stats.csv
grep '^{' $S/bench_trace.json > profiles/r04_bench_line_under_kernel_trace.json
grep '^{' $S/bench_line.json > profiles/r04_bench_line.json
cp $S/bench_extra.json profiles/r04_bench_extra.json
cp $S/live_capacity.txt profiles/r04_live_capacity.txt
python3 tools/design_table.py --write
du -sh profiles
