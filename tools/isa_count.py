#!/usr/bin/env python3
"""Count instructions per kf_step_kernel instantiation in a hipcc -S --cuda-device-only listing."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_ZN2te14kf_step_kernel\w+):', txt)
for k in range(1, len(parts), 2):
    name, body = parts[k], parts[k + 1]
    body = body.split('.Lfunc_end')[0]
    lines = [l.strip() for l in body.split('\n')]
    lines = [l for l in lines if l and not l.startswith(('.', ';', '/')) and not l.split()[0].endswith(':')]
    cnt = lambda p: sum(1 for l in lines if re.match(p, l))
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'te::|void |\(.*', '', dem)
    if flt not in dem: continue
    print("%-46s total %5d valu %5d (f64 %4d trans %3d) salu %4d ds %4d vmem %3d wait %3d branch %3d" % (
        dem, len(lines), cnt(r'v_'), cnt(r'v_\w+_f64'), cnt(r'v_(rcp|rsq|sqrt|sin|cos|exp|log|div)'),
        cnt(r's_(?!waitcnt|nop|cbranch|branch)'), cnt(r'ds_'), cnt(r'(global|buffer|flat)_'), cnt(r's_waitcnt'), cnt(r's_c?branch')))
