#!/usr/bin/env python3
"""Print a bench.py JSON line as a table: python tools/show_bench.py <file>"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("value %.4g %s  ms/step %.4f  n_gpus %d  roofline.frac %.3f (dominant kernel)  tick frac %.3f" % (
    d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["roofline"]["frac"], d["roofline"]["tick"]["frac"]))
print(d["config"]["workload"])
print(d["config"].get("timing"))
for k in d["roofline"]["kernels"]:
    print("  %-44s %8d units x %4d B  %.4f ms  %.0f GB/s  frac %.3f  traffic %s" % (
        k["kernel"], k["units_per_launch"], k["algorithmic_bytes_per_unit"], k["avg_launch_ms"], k["achieved_gbs"], k["frac"], k.get("traffic")))
cb = d.get("cpu_baseline")
if cb:
    print("cpu_baseline:", {k: (("%.4g" % v) if isinstance(v, float) else v) for k, v in cb.items() if k != "per_model_f64"})
    print("  per model f64:", {k: "%.3g" % v for k, v in cb.get("per_model_f64", {}).items()})
if "parity" in d:
    print("parity:", json.dumps(d["parity"].get("models", d["parity"]))[:600])
for e in d.get("extra", []):
    if "error" in e:
        print("  %-16s ERROR %s" % (e["name"], e["error"]))
    elif e["name"] == "gather_pose":
        print("  gather_pose:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()})
    else:
        print("  %-16s %-3s %9d  %9.4g c/s  %8.4f ms  %6.0f GB/s  frac %.3f  %-12s %s" % (
            e["name"], e["dtype"], e["targets_per_gpu"], e["cycles_per_s"], e["ms_per_step"], e["achieved_gbs"], e["roofline_frac"],
            e["residency"][:12], e["layout"][:28]))
