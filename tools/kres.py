#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) as one line per kernel."""
import re, subprocess, sys
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    if "error" in line: print(line, end="")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if flt in r["name"]:
        nm = re.sub(r"te::|\(.*", "", r["name"])
        print("%-60s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (nm[:60], r.get("VGPRs",0), r.get("AGPRs",0), r.get("SGPRs",0), r.get("ScratchSize",0), r.get("Occupancy",0), r.get("LDS Size",0)))
