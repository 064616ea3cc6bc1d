#!/usr/bin/env python3
"""Every figure DESIGN.md quotes from the bench records, generated from the records.

DESIGN.md holds blocks of the form

    <!-- GENERATED:<name> BEGIN (tools/design_table.py; do not edit) -->
    ...
    <!-- GENERATED:<name> END -->

and this tool fills them from the committed records of the round:
    profiles/r04_bench_extra.json      the side file of `python bench.py --gpus 1 --steps 20 --warmup 5` (every workload of the run)
    profiles/hbm_traffic.json          rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per workload (tools/pmc_traffic.py)
    profiles/r04_live_capacity.txt     tools/live_capacity.py
    profiles/r04_kernel_resources.txt  tools/kres_all.sh

    python tools/design_table.py --write     rewrite the blocks of DESIGN.md in place
    python tools/design_table.py --check     exit 1 (and print a diff) if DESIGN.md's blocks differ from what the records give
    python tools/design_table.py --print     print the blocks

tests/test_design_tables.py runs --check: a figure in DESIGN.md that no longer matches the records fails the CPU suite.  (Round 3's
DESIGN.md quoted 10.2 us for a row whose record said 20.9: the numbers were typed by hand.)"""
import difflib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIDE = os.path.join(ROOT, "profiles", "r04_bench_extra.json")
TRAFFIC = os.path.join(ROOT, "profiles", "hbm_traffic.json")
CAPACITY = os.path.join(ROOT, "profiles", "r04_live_capacity.txt")
KRES = os.path.join(ROOT, "profiles", "r04_kernel_resources.txt")
DESIGN = os.path.join(ROOT, "DESIGN.md")


def load():
    side = json.load(open(SIDE))
    traffic = json.load(open(TRAFFIC)) if os.path.exists(TRAFFIC) else {}
    return side, traffic


def pmc_mb(traffic, name):
    rows = {k: v for k, v in traffic.get(name, {}).items() if isinstance(v, dict)}
    tot = sum(r["hbm_read_bytes"] + r["hbm_write_bytes"] for r in rows.values()) if rows else None
    return tot / 1e6 if tot else None


def res(r):
    r = r or ""
    return "HBM" if r.startswith("HBM") else "L3" if r.startswith("L3") else "L2/L3"


def sig(x, n=3):
    return "%.*g" % (n, x)


def block_headline(side, traffic):
    line, roof = side["line"], side["roofline"]
    cfg, t = line["config"], roof["tick"]
    dom = max(roof["kernels"], key=lambda k: k["units_per_launch"] * k["algorithmic_bytes_per_unit"])
    out = ["| | value |", "|---|---|"]
    out.append("| workload | `%s`: %s |" % (cfg["name"], cfg["workload"]))
    out.append("| predict+update cycles/s (`value`) | **%s** (%.4f ms per tick of %d targets; %s) |" % (
        sig(line["value"], 4), line["ms_per_step"], cfg["targets_per_gpu"], side["config"]["timing"]))
    out.append("| launch mode | %s |" % cfg["launch_mode"])
    parts = dom.get("parts")
    if parts:
        out.append("| algorithmic bytes per tick | %.1f MB = %s |" % (
            t["algorithmic_bytes_per_step"] / 1e6, " + ".join("%d × %d B (%s)" % (p["units"], p["algorithmic_bytes_per_unit"], p["model"]) for p in parts)))
    else:
        out.append("| algorithmic bytes per tick | %.1f MB |" % (t["algorithmic_bytes_per_step"] / 1e6))
    mb = pmc_mb(traffic, cfg["name"])
    if mb:
        out.append("| HBM-side bytes per tick (PMC, `profiles/hbm_traffic.json`) | %.1f MB = %.3f × algorithmic |" % (mb, mb * 1e6 / t["algorithmic_bytes_per_step"]))
    out.append("| dominant kernel (`roofline`) | `%s`, %d units per launch × %s B: %.4f ms per launch (HIP events over the timed region) = %.0f GB/s = **%.3f** of 8 TB/s |" % (
        dom["kernel"], dom["units_per_launch"], sig(dom["algorithmic_bytes_per_unit"], 4), dom["avg_launch_ms"], dom["achieved_gbs"], dom["frac"]))
    out.append("| tick, all kernels over the timed region | %.0f GB/s = **%.3f** of 8 TB/s |" % (t["achieved"], t["frac"]))
    if "measured_copy_gbs" in roof:
        out.append("| box's own streaming rates (1 GiB arrays) | device copy %.2f TB/s, triad %.2f TB/s: the tick runs at %.2f × / %.2f × of them |" % (
            roof["measured_copy_gbs"] / 1e3, roof["measured_triad_gbs"] / 1e3, t.get("frac_of_measured_copy", 0), t.get("frac_of_measured_triad", 0)))
    cb = side.get("cpu_baseline")
    if cb and "value" in cb:
        pm = cb.get("per_model_f64", {})
        out.append("| CPU baseline (`kind: %s`, %d threads; %s) | %s cycles/s%s |" % (
            cb["kind"], cb["cores"], cb["sample"], sig(cb["value"]),
            ("; per model in fp64: " + ", ".join("%s %s" % (k, sig(v)) for k, v in pm.items())) if pm else ""))
    par = side.get("parity", {}).get("models")
    if par:
        out.append("| parity sample (%d targets per model after %s ticks, same-precision oracle, stream regenerated on the CPU) | %s |" % (
            side["parity"]["targets_per_model"], " / ".join(str(x) for x in side["parity"]["ticks"]),
            "; ".join("%s max abs dx %s, max rel dP %s, ids %s" % (m, sig(max(v["max_abs_x"]), 2), sig(max(v["max_rel_P"]), 2), "exact" if v["ids_exact"] else "DIFFER")
                      for m, v in par.items())))
    c0 = side.get("configs0", {}).get("rows")
    if c0:
        out.append("| configs[0] (1 target, the reference test's own stream and loop) | " + "; ".join(
            "%d steps: oracle %.2f µs per step, the ten C symbols on the GPU %.1f µs, max abs pose difference %s" % (
                r["steps"], r.get("cpu_us_per_step", float("nan")), r.get("gpu_us_per_step", float("nan")), sig(r["max_abs_pose_difference"], 2)) for r in c0) + " |")
    return out


def block_workloads(side, traffic):
    line = side["line"]
    t = side["roofline"]["tick"]
    out = ["| workload | dtype | targets | cycles/s | µs/tick | alg. B/cycle | alg. GB/s | frac | PMC MB/tick | residency | launches per tick |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]

    def mode(m):
        return "1 (population)" if "whole population" in m else "resident" if m.startswith("live") else "per batch" if m.startswith(("graph", "sequence", "python")) else m

    mb = pmc_mb(traffic, line["config"]["name"])
    out.append("| **%s** (headline) | %s | %d | %s | %.1f | %d | %.0f | **%.3f** | %s | %s | %s |" % (
        line["config"]["name"], line["dtype"], line["config"]["targets_per_gpu"], sig(line["value"]), line["ms_per_step"] * 1e3,
        round(t["algorithmic_bytes_per_step"] / line["config"]["targets_per_gpu"]), t["achieved"], t["frac"], ("%.1f" % mb) if mb else "–",
        res(side["config"]["residency"]), mode(line["config"]["launch_mode"])))
    for e in side.get("extra", []):
        if "error" in e or e.get("name") == "gather_pose":
            continue
        if "live" in e:
            lv = e["live"]
            out.append("| %s | %s | %d | %s | %.2f (paced %.1f) | – | – | – | – | registers | resident |" % (
                e["name"], e["dtype"], e["targets_per_gpu"], sig(e["cycles_per_s"]), lv["us_per_tick_back_to_back"], lv["us_per_tick_paced"]))
            continue
        mb = pmc_mb(traffic, e["name"])
        out.append("| %s | %s | %d | %s | %.1f | %d | %.0f | %.3f | %s | %s | %s |" % (
            e["name"], e["dtype"], e["targets_per_gpu"], sig(e["cycles_per_s"]), e["ms_per_step"] * 1e3, round(e["algorithmic_bytes_per_cycle"]),
            e["achieved_gbs"], e["roofline_frac"], ("%.1f" % mb) if mb else "–", res(e.get("residency")), mode(e.get("launch_mode", ""))))
    return out


def block_small(side, traffic):
    """launch per tick against the resident mode for the launch-bound configs"""
    ex = {e["name"]: e for e in side.get("extra", []) if "error" not in e}
    rows = [("cfg2", "configs[1]: 10 000 UV fp64"), ("cfg3", "configs[2]: 100 000 UA fp32"), ("cfg4", "configs[3], one GPU's share of 8: 62 500 AR + 62 500 AV, fp32"),
            ("cfg4_64", "the same in fp64 (the reference's arithmetic)"), ("cfg5", "configs[4], one GPU's share: 62 500 AR + 62 500 UA + sphere query every tick, fp32")]
    out = ["| workload | what | launch per tick: µs/tick (fraction of 8 TB/s) | resident, one doorbell per tick back to back: µs/tick | resident, paced (Python caller): µs/tick |",
           "|---|---|---|---|---|"]
    for name, what in rows:
        a, b = ex.get(name), ex.get(name + "_live")
        if not a:
            continue
        out.append("| %s | %s | %.2f (%.3f) | %s | %s |" % (
            name, what, a["ms_per_step"] * 1e3, a["roofline_frac"],
            ("%.2f" % b["live"]["us_per_tick_back_to_back"]) if b else "–", ("%.1f" % b["live"]["us_per_tick_paced"]) if b else "–"))
    return out


def block_bands(side, traffic):
    ex = [e for e in side.get("extra", []) if "error" not in e and "live" not in e and e.get("name") != "gather_pose"]
    default = [e for e in ex if not re.search(r"_(full|packed|s201|1kcls|1kcls_rand|a90|stream)$", e["name"])]
    hbm = [e for e in default if res(e.get("residency")) == "HBM"]
    l3 = [e for e in default if res(e.get("residency")) == "L3" and e["targets_per_gpu"] >= 1_000_000]
    dense = [e for e in ex if re.search(r"_(full|packed)$", e["name"]) and e["targets_per_gpu"] >= 1_000_000]
    out = []

    def band(rows):
        f = [e["roofline_frac"] for e in rows]
        return "%.3f–%.3f of 8 TB/s (%.2f–%.2f TB/s)" % (min(f), max(f), min(f) * 8, max(f) * 8)
    if hbm:
        out.append("* HBM-bound rows, default layout (state > 1 GB: %s): **%s**." % (", ".join(e["name"] for e in hbm), band(hbm)))
    if l3:
        out.append("* 10⁶-target rows, default layout, Infinity-Cache-assisted (%s): %s." % (", ".join(e["name"] for e in l3), band(l3)))
    if dense:
        out.append("* dense kernels for coupled matrices at 10⁶ targets (%s): %s." % (", ".join("%s %.3f" % (e["name"], e["roofline_frac"]) for e in dense), band(dense)))
    slow = side.get("slower_than_previous_record")
    if slow is not None:
        out.append("* rows more than 1.25 × slower than in the previous committed record: %s." % (
            ", ".join("%s %.2f ×" % (s["name"], s["ratio"]) for s in slow) if slow else "none"))
    return out


def block_capacity(side, traffic):
    out = ["```"]
    out += [ln.rstrip() for ln in open(CAPACITY) if ln.strip()]
    out.append("```")
    return out


def block_resources(side, traffic):
    want = [r"kf_step_population_kernel<", r"kf_step_sep_kernel<(UV|UA|AR|AV),(double|float),3,0,0,0,0,0,0>", r"kf_step_sep_kernel<(UV|UA|AR|AV),(double|float),3,0,1,0,0,[12],0>"]
    out = ["```"]
    for ln in open(KRES):
        if any(re.match(w, ln) for w in want):
            out.append(ln.rstrip())
    out.append("```")
    return out


BLOCKS = {"headline": block_headline, "workloads": block_workloads, "small_populations": block_small, "bands": block_bands,
          "resident_capacity": block_capacity, "kernel_resources": block_resources}


def render():
    side, traffic = load()
    return {name: "\n".join(fn(side, traffic)) for name, fn in BLOCKS.items()}


def apply(text, blocks):
    for name, body in blocks.items():
        pat = re.compile(r"(<!-- GENERATED:%s BEGIN[^\n]*-->\n).*?(<!-- GENERATED:%s END -->)" % (name, name), re.S)
        if not pat.search(text):
            raise SystemExit("DESIGN.md has no block GENERATED:%s" % name)
        text = pat.sub(lambda m: m.group(1) + body + "\n" + m.group(2), text)
    return text


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "--print"
    blocks = render()
    if mode == "--print":
        for name, body in blocks.items():
            print("<!-- GENERATED:%s BEGIN (tools/design_table.py; do not edit) -->\n%s\n<!-- GENERATED:%s END -->\n" % (name, body, name))
        return 0
    text = open(DESIGN).read()
    new = apply(text, blocks)
    if mode == "--write":
        open(DESIGN, "w").write(new)
        return 0
    if mode == "--check":
        if new != text:
            sys.stdout.writelines(difflib.unified_diff(text.splitlines(True), new.splitlines(True), "DESIGN.md", "DESIGN.md (from the records)", n=1))
            return 1
        return 0
    raise SystemExit(__doc__)


if __name__ == "__main__":
    sys.exit(main())
