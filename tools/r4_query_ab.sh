#!/bin/bash
# The sphere query's solver with and without the Sturm classification (csrc/te_quartic.hpp), same box, alternating: configs[4]'s per-GPU
# share, its parts, and the 10^6-target rows.  The comparison library: make OUT=../lib_base OBJ=../lib_base/obj HIPCC="hipcc -DTE_QUARTIC_NO_STURM".
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
BASE=$PWD/target_estimation_amd/lib_base/libtarget_estimation_amd.so
row() { python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('   %-14s %8.2f us/tick  frac %.3f' % (r['config']['workload'].split(':')[0][:14], r['ms_per_step'] * 1e3, r['roofline']['frac']))
"; }
export TE_PARTS_ONLY=0,2,4,9
for rep in 1 2; do
  for v in base sturm; do
    echo "== $v (pass $rep)"
    if [ $v = base ]; then export TARGET_ESTIMATION_AMD_LIB=$BASE; else unset TARGET_ESTIMATION_AMD_LIB; fi
    python tools/mixed_parts.py f32 | grep -v "^#" || exit 1
    for w in cfg5_1gpu cfg5_1gpu64; do python bench.py --workload $w --steps 200 --warmup 20 --extra none 2>/dev/null | row || exit 1; done
  done
done
