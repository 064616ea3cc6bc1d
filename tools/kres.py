#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) as one line per kernel:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c kf_model_ar.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | tools/kres.py [substring]
Template arguments: kf_step_sep_kernel<model, T, layout(2 separable, 3 separable+packed), INDEXED, FUSED, QUERY, PERQR>;
kf_step_kernel<model, T, lanes per target, layout(0 full, 1 packed), INDEXED, FUSED, QUERY, PERQR>."""
import re
import subprocess
import sys

cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    if " error" in line:
        print(line, end="")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if flt in r["name"]:
        nm = re.sub(r"te::|void |\(.*", "", r["name"]).replace("Model", "").replace("false", "0").replace("true", "1").replace(" ", "")
        tot = r.get("VGPRs", 0) + r.get("AGPRs", 0)
        print("%-52s vgpr %3d agpr %3d (%3d) sgpr %3d scratch %4d waves/SIMD %d lds %6d" % (
            nm[:52], r.get("VGPRs", 0), r.get("AGPRs", 0), tot, r.get("TotalSGPRs", 0), r.get("ScratchSize", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
