#!/usr/bin/env python3
"""HBM-side traffic of the step kernels of bench.py workloads, from rocprofv3 PMC counters.

For every workload: two separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`; the TCC block cannot hold both, and
gpurun refuses counters together with the runtime traces), one workload per process, the program directly behind `--`.
Corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced stream -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16-byte-per-lane stores.  Both
counters sit on the L2's memory side: Infinity-Cache hits are included.

    python tools/pmc_traffic.py --out gpurun_out/pmc_r02 [--workloads a,b,...]   (on the GPU box)

Writes <out>/hbm_traffic.json ({workload: {kernel: {hbm_read_bytes, hbm_write_bytes, launches, grid}}}), keeps the raw
*_counter_collection.csv and every pass's stderr next to it.  Copy the JSON to profiles/hbm_traffic.json and the raw
directory to profiles/ to have bench.py report `roofline.traffic`.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short_kernel(name):
    """'void te::kf_step_sep_kernel<te::ModelAR, double, 3, false, ...>(...)' -> 'kf_step_sep_kernel<ModelAR,double,3>'
    (the form bench.py's kernel_name() prints): model, type, then LAYOUT (separable) or G, LAYOUT (dense)."""
    pm = re.search(r"kf_step_population_kernel<(double|float), (true|false), (true|false)>", name.replace("te::", ""))
    if pm:   # the whole population of a manager in one launch: <T, QUERY, AB>
        return "kf_step_population_kernel<%s>" % pm.group(1) + ("+query" if pm.group(2) == "true" else "")
    m = re.search(r"(kf_step(?:_sep)?_kernel)<([^>]*)>", name.replace("te::", ""))
    if not m:
        return None
    args = [a.strip() for a in m.group(2).split(",")]
    keep = 3 if m.group(1) == "kf_step_sep_kernel" else 4
    return "%s<%s>" % (m.group(1), ",".join(args[:keep]))


def one_pass(workload, counter, out, steps, extra_args):
    d = os.path.join(out, workload, counter)
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", counter, "--",
           "python3", os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", "2", "--reps", "1",
           "--no-cpu", "--no-gather", "--extra", ""] + extra_args
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(out, workload, counter + ".stderr.txt"), "w") as err, open(os.path.join(out, workload, counter + ".stdout.txt"), "w") as so:
        rc = subprocess.call(cmd, stdout=so, stderr=err, env=env, cwd="/tmp")
    rows = {}
    kept = []
    header = None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rd = csv.DictReader(open(f))
        header = rd.fieldnames
        for r in rd:
            k = short_kernel(r["Kernel_Name"])
            if k and r["Counter_Name"] == counter:
                rows.setdefault((k, r.get("Grid_Size", "")), []).append(float(r["Counter_Value"]))
                kept.append(r)
    # keep the raw rows of the step kernels only (the full CSVs of 80 passes do not fit gpurun's 64 MiB return limit)
    if header:
        with open(os.path.join(out, workload, counter + "_counter_collection.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=header)
            w.writeheader()
            w.writerows(kept)
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return rc, rows


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_traffic"))
    ap.add_argument("--workloads", default=",".join([bench.HEADLINE] + bench.DEFAULT_EXTRA.split(",")))
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    args.out = os.path.abspath(args.out)   # rocprofv3 runs with cwd = /tmp
    os.makedirs(args.out, exist_ok=True)
    result, failures = {}, []
    try:   # the commit the snapshot on this box was taken at (written before the gpurun call; there is no .git here)
        commit = open(os.path.join(ROOT, ".head_commit")).read().strip()
    except OSError:
        commit = "unknown"
    result["_meta"] = {"commit": commit, "tool": "tools/pmc_traffic.py", "counters": "FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM)"}
    # the resident (live) and free-running (replay) extras are ONE launch for many ticks: no per-tick launch to attribute counters to
    for wl in [w for w in args.workloads.split(",") if w and not w.endswith("_live") and not w.endswith("_replay")]:
        per = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rc, rows = one_pass(wl, counter, args.out, args.steps, [])
            if rc != 0 or not rows:
                failures.append("%s %s rc=%d rows=%d" % (wl, counter, rc, len(rows)))
                print("FAILED", failures[-1], flush=True)
                continue
            for (k, grid), v in sorted(rows.items(), key=lambda kv: int(kv[0][1] or 0)):   # one kernel at two grids: the larger one wins
                e = per.setdefault(k, {})
                e["grid"] = grid
                # skip the first launches (cold caches / first touch): average the second half
                tail = v[len(v) // 2:]
                avg = sum(tail) / len(tail)
                if counter == "FETCH_SIZE":
                    e["hbm_read_bytes"] = 2.0 * 1024.0 * avg
                    e["fetch_size_kb_raw"] = avg
                else:
                    e["hbm_write_bytes"] = 1024.0 * avg
                e["launches"] = len(v)
        per = {k: e for k, e in per.items() if "hbm_read_bytes" in e and "hbm_write_bytes" in e}
        if per:
            result[wl] = per
        for k, e in per.items():
            print("%-18s %-44s grid %-10s read %9.3f MB  write %9.3f MB  (%d launches)" % (
                wl, k, e["grid"], e["hbm_read_bytes"] / 1e6, e["hbm_write_bytes"] / 1e6, e["launches"]), flush=True)
        json.dump(result, open(os.path.join(args.out, "hbm_traffic.json"), "w"), indent=1, sort_keys=True)
    print("failures:", failures)


if __name__ == "__main__":
    main()
