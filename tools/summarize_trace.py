#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> one line per (kernel, grid): calls, avg / min / max duration.
usage: tools/summarize_trace.py <dir with *kernel_trace.csv> [bench line file [bench side file]]   (prints to stdout)"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("te::", "")


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    groups = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f, newline="")):
            g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            groups[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("# rocprofv3 --kernel-trace --stats summary; template arguments of kf_step_sep_kernel: <model, precision, layout (2 = separable,")
    print("# 3 = separable + packed), INDEXED, FUSED, QUERY, PERQR, LIVE (1 = resident, 2 = resident with per-tick query / pose output), AB>, of kf_step_kernel:")
    print("# <model, precision, lanes per target, layout (0 = full, 1 = packed), INDEXED, FUSED, QUERY, PERQR, AB>; kf_step_population_kernel<precision, QUERY, AB>:")
    print("# ONE launch for every batch of a manager (grid = the sum over the models of their targets, each rounded up to whole workgroups)")
    print("%-78s %9s %6s %10s %10s %10s" % ("kernel", "grid", "calls", "avg_us", "min_us", "max_us"))
    rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
    for (k, g), durs in rows:
        if "kf_step" not in k and len(durs) * (sum(durs) / len(durs)) < 2e6:
            continue
        print("%-78s %9d %6d %10.2f %10.2f %10.2f" % (k[:78], g, len(durs), sum(durs) / len(durs) / 1e3, min(durs) / 1e3, max(durs) / 1e3))
    if len(sys.argv) > 2:
        line = [ln for ln in open(sys.argv[2]) if ln.startswith("{")][-1]
        b = json.loads(line)
        print("# bench line of the same process: value %.4g %s, ms_per_step %.4f, roofline.kernel %s avg_launch_ms %.4f (HIP events) -- compare with the"
              % (b["value"], b["unit"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["avg_launch_ms"]))
        print("# rocprofv3 average of that kernel at the headline's grid above (a population launch: the sum of its parts' grids, each a multiple of 256)")
        kernels = b["roofline"].get("kernels")
        if kernels is None and len(sys.argv) > 3:          # the per-kernel table lives in the side file (bench.py --side-file)
            kernels = json.load(open(sys.argv[3]))["roofline"]["kernels"]
        for k in kernels or []:
            print("#   %-44s units %8d  avg_launch_ms (events) %.4f" % (k["kernel"], k["units_per_launch"], k["avg_launch_ms"]))


if __name__ == "__main__":
    main()
