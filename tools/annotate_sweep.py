#!/usr/bin/env python3
"""Join tools/sweep.py output (stdin) with the compiler's resource table (profiles/r02_kernel_resources.txt) and mark the
lanes codes the library picks by itself (csrc/target_manager.cpp chooseLayout, kf_model_*.hip defaults).
    python tools/sweep.py --steps 100 --sizes 1000000 > gpurun_out/sweep.txt     (GPU box)
    python tools/annotate_sweep.py < gpurun_out/sweep.txt > profiles/r02_layout_sweep.txt"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHORT = {"uniform_velocity": "UV", "uniform_acceleration": "UA", "angular_rates": "AR", "angular_velocities": "AV"}
res = {}
for ln in open(os.path.join(ROOT, "profiles", os.environ.get("TE_KRES_FILE", "r04_kernel_resources.txt"))):
    m = re.match(r"(kf_step\w*<[^>]*>)\s+vgpr\s+(\d+) agpr\s+(\d+) \(\s*(\d+)\) sgpr\s+\d+ scratch\s+(\d+) waves/SIMD (\d)", ln)
    if m:
        res[m.group(1)] = (int(m.group(4)), int(m.group(5)), int(m.group(6)))
# (model, precision) -> {lanes code: note}
AUTO = {}
for mdl in SHORT:
    for prec in ("f64", "f32"):
        a = {301: "shipped models (axis-separable, symmetric)"}
        if mdl == "uniform_velocity":
            a[101] = "coupled symmetric matrices"; a[1] = "non-symmetric matrices"
        elif mdl == "uniform_acceleration":
            a[101 if prec == "f64" else 103] = "coupled symmetric matrices"; a[3] = "non-symmetric matrices"
        elif mdl == "angular_rates":
            a[106 if prec == "f64" else 103] = "coupled symmetric matrices"; a[6] = "non-symmetric matrices"
        else:
            a[101] = "coupled symmetric matrices (csrc/ekf_sym.hpp)"; a[6 if prec == "f64" else 3] = "non-symmetric matrices"
        AUTO[(mdl, prec)] = a
print("# tools/sweep.py --steps 100 --sizes 1000000 on one MI355X (zig-zag traversal on), every (model, precision, lanes code) the library instantiates, joined by")
print("# tools/annotate_sweep.py with the register budget of the kernel that runs (profiles/" + os.environ.get("TE_KRES_FILE", "r04_kernel_resources.txt") + ").  lanes code G: dense kernel, full P, G lanes per")
print("# target; 100+G: dense kernel, symmetric-packed P; 201: axis-separable, full group blocks; 301: axis-separable, packed group blocks.")
print("# frac = algorithmic GB/s / 8000 with the bytes the kernel reads + writes.  '<- auto' marks what the library picks by itself.")
for ln in sys.stdin:
    p = ln.split()
    if len(p) >= 8 and p[0] in SHORT and not p[2].isdigit():   # a row that brings its own annotation (tools/wave_per_target.hip)
        print(ln.rstrip())
        continue
    if len(p) < 8 or p[0] not in SHORT:
        if ln.startswith("model"):
            print(ln.rstrip())
        continue
    mdl, prec, g = p[0], p[1], int(p[2])
    T = "double" if prec == "f64" else "float"
    if g >= 200:
        key = "kf_step_sep_kernel<%s,%s,%d,0,0,0,0,0,0>" % (SHORT[mdl], T, 2 if g == 201 else 3)
    else:
        key = "kf_step_kernel<%s,%s,%d,%d,0,0,0,0,0>" % (SHORT[mdl], T, g % 100, 1 if g >= 100 else 0)
    r = res.get(key)
    note = "regs %3d scratch %3d waves/SIMD %d" % r if r else "regs ?"
    auto = AUTO[(mdl, prec)].get(g)
    if (mdl, prec, g) == ("angular_rates", "f64", 106):
        note += " (held to two wavefronts, kf_step.hpp step_min_waves; unconstrained: 262 registers, one wavefront, 1034 us)"
    print("%s   %s%s" % (ln.rstrip(), note, ("  <- auto: " + auto) if auto else ""))
