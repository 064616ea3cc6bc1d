#!/usr/bin/env python3
"""Do the step kernels of the two batches of a mixed tick overlap on the device?  Reads a rocprofv3 --kernel-trace CSV
(*_kernel_trace.csv) and prints, for consecutive kf_step launches, the gap start(k+1) - end(k) (negative = overlap).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload cfg4_64 --extra "" --no-cpu --no-gather
    python tools/overlap.py DIR"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "kf_step" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[1].split(",")[0], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
print("%d step launches" % len(rows))
tail = rows[len(rows) // 2:]
gaps = [tail[i + 1][0] - tail[i][1] for i in range(len(tail) - 1)]
dur = [r[1] - r[0] for r in tail]
neg = sum(1 for g in gaps if g < 0)
print("second half: %d launches, mean duration %.2f us, overlapping pairs %d of %d, median gap %.2f us" % (
    len(tail), sum(dur) / len(dur) / 1e3, neg, len(gaps), sorted(gaps)[len(gaps) // 2] / 1e3))
span = (tail[-1][1] - tail[0][0]) / 1e3
print("span %.1f us for %d launches = %.2f us per launch; sum of durations %.1f us" % (span, len(tail), span / len(tail), sum(dur) / 1e3))
for r in tail[:12]:
    print("  start %12.2f us  dur %6.2f us  %s  queue %s stream %s" % ((r[0] - tail[0][0]) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[3], r[4]))
