#!/usr/bin/env python3
"""Where is the vector-register peak of a gfx950 kernel?  Reads the ISA listing of ONE kernel (hipcc -S --cuda-device-only)
and runs a backward liveness pass over its VGPRs (block-level CFG from the labels and branches; a definition kills, which is
slightly optimistic inside divergent regions).  Prints the number of live VGPRs at every label and the peak with its line.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I target_estimation_amd/csrc -S --cuda-device-only one_kernel.hip -o k.s
    python tools/isa_liveness.py k.s [kernel-substring]
"""
import re
import sys

STORE_LIKE = ("global_store", "flat_store", "buffer_store", "scratch_store", "ds_write", "ds_store", "global_atomic", "s_", "v_cmp", "v_cmpx",
              "v_readlane", "v_readfirstlane", "global_wb", "buffer_wbl2", "buffer_inv")
RMW = ("v_fmac", "v_mac", "v_dot", "v_mfma", "v_cndmask")   # (cndmask: reads both sources, writes dst -- dst not read; kept out below)


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"(?<![\w\[])v(\d+)\b", tok):
        out.add(int(a))
    return out


def parse(path, want):
    lines = open(path).read().split("\n")
    body, inside = [], False
    for i, l in enumerate(lines):
        m = re.match(r"^(\S+):\s*(;.*)?$", l)
        if m and not l.startswith(".L") and not l.startswith("\t"):
            inside = (want in m.group(1)) if want else m.group(1).startswith("_Z")
            continue
        if inside:
            if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
                inside = False
                continue
            body.append((i + 1, l))
    return body


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    body = parse(path, want)
    blocks, cur = [], {"label": "entry", "ins": [], "line": body[0][0] if body else 0}
    for ln, l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ins": [], "line": ln}
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        cur["ins"].append((ln, t))
    blocks.append(cur)
    idx = {b["label"]: k for k, b in enumerate(blocks)}
    for k, b in enumerate(blocks):
        succ = []
        fall = True
        for ln, t in b["ins"]:
            op = t.split()[0]
            if op in ("s_branch",):
                succ.append(idx.get(t.split()[1], None))
                fall = False
            elif op.startswith("s_cbranch"):
                succ.append(idx.get(t.split()[1], None))
            elif op in ("s_endpgm",):
                fall = False
        if fall and k + 1 < len(blocks):
            succ.append(k + 1)
        b["succ"] = [s for s in succ if s is not None]
    # per instruction defs / uses
    for b in blocks:
        ops = []
        for ln, t in b["ins"]:
            parts = t.split(None, 1)
            op = parts[0]
            args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
            # re-join "v[1:2]" style tokens are intact (commas only separate operands)
            if op.startswith(STORE_LIKE) and not op.startswith(("v_cndmask",)):
                d, u = set(), set().union(*[regs(a) for a in args]) if args else set()
                if op.startswith(("v_readlane", "v_readfirstlane")):
                    u = set().union(*[regs(a) for a in args[1:]]) if len(args) > 1 else set()
                if op.startswith("global_atomic") and "sc0" in t and args:   # returning atomic: first operand is the result
                    d = regs(args[0]); u = set().union(*[regs(a) for a in args[1:]])
            else:
                d = regs(args[0]) if args else set()
                u = set().union(*[regs(a) for a in args[1:]]) if len(args) > 1 else set()
                if op.startswith(("v_fmac", "v_mac", "v_writelane", "v_accvgpr")) or "op_sel" in t and False:
                    u |= d
                if op.startswith(("v_div_scale", "v_add_co", "v_sub_co", "v_addc", "v_subb", "v_mad_u64", "v_mad_i64")) and len(args) > 1:
                    pass   # second operand is an SGPR/VCC def: regs() ignores it
            ops.append((ln, t, d, u))
        b["ops"] = ops
    live_in = [set() for _ in blocks]
    changed = True
    while changed:
        changed = False
        for k in range(len(blocks) - 1, -1, -1):
            b = blocks[k]
            live = set()
            for s in b["succ"]:
                live |= live_in[s]
            for ln, t, d, u in reversed(b["ops"]):
                live = (live - d) | u
            if live != live_in[k]:
                live_in[k] = live
                changed = True
    peak, where = 0, None
    rows = []
    for k, b in enumerate(blocks):
        live = set()
        for s in b["succ"]:
            live |= live_in[s]
        bmax, bline, btxt = len(live), b["line"], ""
        for ln, t, d, u in reversed(b["ops"]):
            here = len(live | d)
            if here > bmax:
                bmax, bline, btxt = here, ln, t
            live = (live - d) | u
        rows.append((b["label"], b["line"], len(b["ops"]), len(live_in[k]), bmax, bline, btxt))
        if bmax > peak:
            peak, where = bmax, (b["label"], bline, btxt)
    for r in rows:
        if r[2] >= 8:
            print("%-12s line %5d  %4d instr  live-in %3d  max %3d at line %5d  %s" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6][:60]))
    print("peak %d live VGPRs in %s at line %d: %s" % (peak, where[0], where[1], where[2]))


if __name__ == "__main__":
    main()
