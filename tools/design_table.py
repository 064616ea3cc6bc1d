#!/usr/bin/env python3
"""The per-workload table of DESIGN.md section 4 from a bench side file and the PMC traffic file:
    python tools/design_table.py profiles/r03_bench_extra.json profiles/hbm_traffic.json"""
import json
import sys

side = json.load(open(sys.argv[1]))
traffic = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}


def pmc(name):
    rows = traffic.get(name, {})
    tot = sum(r["hbm_read_bytes"] + r["hbm_write_bytes"] for r in rows.values()) if rows else None
    return "%.1f" % (tot / 1e6) if tot else "–"


def res(r):
    r = r or ""
    return "HBM" if r.startswith("HBM") else "L3" if r.startswith("L3") else "L2/L3"


line = side["line"]
print("| workload | dtype | targets | cycles/s | µs/tick | alg. B/cycle | alg. GB/s | frac | PMC MB/tick | residency |")
print("|---|---|---|---|---|---|---|---|---|---|")
t = side["roofline"]["tick"]
print("| **%s** (headline) | %s | %d | %.3g | %.1f | %d | %.0f | **%.3f** | %s | %s |" % (
    line["config"]["name"], line["dtype"], line["config"]["targets_per_gpu"], line["value"], line["ms_per_step"] * 1e3,
    t["algorithmic_bytes_per_step"] / line["config"]["targets_per_gpu"], t["achieved"], t["frac"], pmc(line["config"]["name"]), res(side["config"]["residency"])))
for e in side.get("extra", []):
    if "error" in e or e.get("name") == "gather_pose":
        continue
    if "live" in e:
        lv = e["live"]
        print("| %s | %s | %d | %.3g | %.2f (paced %.1f) | – | – | – | – | registers |" % (
            e["name"], e["dtype"], e["targets_per_gpu"], e["cycles_per_s"], lv["us_per_tick_back_to_back"], lv["us_per_tick_paced"]))
        continue
    print("| %s | %s | %d | %.3g | %.1f | %d | %.0f | %.3f | %s | %s |" % (
        e["name"], e["dtype"], e["targets_per_gpu"], e["cycles_per_s"], e["ms_per_step"] * 1e3, e["algorithmic_bytes_per_cycle"],
        e["achieved_gbs"], e["roofline_frac"], pmc(e["name"]), res(e.get("residency"))))
