#!/bin/bash
# Copy what tools/r3_final.sh and the two tools/pmc_traffic.py calls left under gpurun_out/r3final into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r3final
python3 - <<PY
import json
a = json.load(open("$S/pmcA/hbm_traffic.json")); b = json.load(open("$S/pmcB/hbm_traffic.json"))
assert a["_meta"]["commit"] == b["_meta"]["commit"], (a["_meta"], b["_meta"])
a.update({k: v for k, v in b.items() if k != "_meta"})
json.dump(a, open("profiles/hbm_traffic.json", "w"), indent=1, sort_keys=True)
print("workloads:", len(a) - 1, "commit", a["_meta"]["commit"])
PY
rm -rf profiles/r03_pmc; mkdir -p profiles/r03_pmc
for part in pmcA pmcB; do
  for d in $S/$part/*/; do
    w=$(basename $d); mkdir -p profiles/r03_pmc/$w
    cp $d/*_counter_collection.csv profiles/r03_pmc/$w/ 2>/dev/null || true
    for c in FETCH_SIZE WRITE_SIZE; do grep -v "^W20\|^I20\|^E20" $d/$c.stderr.txt | tail -40 > profiles/r03_pmc/$w/$c.stderr.txt 2>/dev/null || true; done
  done
done
cat $S/pmcA.txt $S/pmcB.txt | grep -v "^$" > profiles/r03_pmc_traffic_summary.txt
cp $S/bench_default_rocprofv3.txt profiles/r03_bench_default_rocprofv3.txt
cp $S/bench_default_kernel_stats.csv profiles/r03_bench_default_kernel_stats.csv
grep '^{' $S/bench_trace.json > profiles/r03_bench_line_under_kernel_trace.json
grep '^{' $S/bench_line.json > profiles/r03_bench_line.json
cp $S/bench_extra.json profiles/r03_bench_extra.json
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --gpus 1 --steps 20 --warmup 5   (EVERY workload of the default bench, incl. the"
  echo "# resident-mode extras, in ONE process; taken ONCE at HEAD, tools/r3_final.sh).  Round 2: exit 139 inside a torch elementwise launch of the"
  echo "# stream set-up (profiles/r02_pmc_all_in_one_sigsegv.txt).  Round 3: the set-up launches no torch kernel (csrc/stream_gen.hip) and the pass exits 0;"
  echo "# the same create / step / graph-capture / live-session / destroy sequence of managers, batches, graph execs, pinned blocks and the"
  echo "# PoseComm ran to the end under the same counter collection, which is what excludes a stale mapping left by this library's teardown."
  cat $S/progress.txt
  echo "# the line that process printed (timings under counter collection: not performance numbers):"; grep '^{' $S/allinone.json; } > profiles/r03_pmc_all_in_one.txt
du -sh profiles
