#!/usr/bin/env python3
"""Live VGPR/AGPR count along the straight-line body of one kernel in a hipcc -S listing (backward liveness, branches
ignored: the step kernels are one long basic block with a few skipped tails).  Prints the pressure every `stride`
instructions and the source-level landmarks (sched_barrier comments, loads, stores) so that the peak can be placed.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only x.hip -o x.s && tools/vgpr_pressure.py x.s [kernel-substring] [stride]
"""
import re
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().splitlines()
# kernel bodies: from "<name>:" to "s_endpgm"
start = None
for i, ln in enumerate(lines):
    if re.match(r"^_Z\w+:", ln) and flt in ln:
        start = i
        break
if start is None:
    sys.exit("kernel not found")
body = []
for ln in lines[start + 1:]:
    t = ln.strip()
    if t.startswith(".Lfunc_end") or t.startswith(".section"):
        break
    if not t or t.startswith(";") and "sched_barrier" not in t:
        continue
    body.append(t)

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for k in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), k))
    return out


NODEF = ("global_store", "ds_write", "buffer_store", "s_", "v_cmp", "flat_store", "scratch_store", "v_cmpx", "global_atomic", "ds_add")
RMW = ("v_fmac", "v_mac", "v_pk_fmac", "v_dot", "v_accvgpr_write")
live = set()
press = [0] * len(body)
for i in range(len(body) - 1, -1, -1):
    t = body[i]
    if t.startswith(";") or t.endswith(":") or t.startswith("."):
        press[i] = len(live)
        continue
    op, _, rest = t.partition(" ")
    rest = rest.split(";")[0]
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith(NODEF):
        d, u = set(), set().union(*[regs(o) for o in ops]) if ops else set()
    else:
        d = regs(ops[0]) if ops else set()
        u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        if op.startswith(RMW):
            u |= d
    live -= d
    live |= u
    press[i] = len(live)
peak = max(press)
pi = press.index(peak)
print("instructions %d, peak live %d at #%d: %s" % (len(body), peak, pi, body[pi][:70]))
for i in range(0, len(body), stride):
    seg = press[i:i + stride]
    marks = [b.split()[0] for b in body[i:i + stride] if b.startswith(("global_load", "global_store", "ds_read", "ds_write", "; sched", ";sched"))]
    summ = {}
    for m in marks:
        summ[m] = summ.get(m, 0) + 1
    print("#%5d  live max %3d min %3d   %s" % (i, max(seg), min(seg), " ".join("%s x%d" % kv for kv in summ.items())))

if len(sys.argv) > 4:   # dump: where the registers live at instruction #at were defined
    at = int(sys.argv[4])
    # recompute the live set at `at`
    live = set()
    for i in range(len(body) - 1, at - 1, -1):
        t = body[i]
        if t.startswith(";") or t.endswith(":") or t.startswith("."):
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(";")[0].split(",")]
        if op.startswith(NODEF):
            d, u = set(), set().union(*[regs(o) for o in ops]) if ops else set()
        else:
            d = regs(ops[0]) if ops else set()
            u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if op.startswith(RMW):
                u |= d
        live -= d
        live |= u
    where = {}
    for r in live:
        for i in range(at - 1, -1, -1):
            t = body[i]
            if t.startswith(";") or t.endswith(":") or t.startswith("."):
                continue
            op, _, rest = t.partition(" ")
            if op.startswith(NODEF):
                continue
            ops = [o.strip() for o in rest.split(";")[0].split(",")]
            if ops and r in regs(ops[0]):
                where[r] = (i, op)
                break
    hist = {}
    for r, (i, op) in where.items():
        key = (i // 100 * 100, op)
        hist[key] = hist.get(key, 0) + 1
    for k in sorted(hist):
        print("  defined in #%5d.. by %-24s : %d registers" % (k[0], k[1], hist[k]))
    print("  live %d, with a definition found %d" % (len(live), len(where)))
