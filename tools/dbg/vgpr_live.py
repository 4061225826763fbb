#!/usr/bin/env python3
"""Live-VGPR profile of one kernel in a `hipcc -S` listing (straight-line approximation: branches ignored).
usage: vgpr_live.py file.s kernel_name_substring [every]
Prints the number of live VGPRs every `every` instructions and the maximum with its position; used to see WHERE a fully
unrolled body (k_rbfull) reaches its register peak without a GPU."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
ins = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    ins.append(t.split(";")[0].strip())
rx = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in rx.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


defs, uses = [], []
for t in ins:
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    d, u = set(), set()
    is_store = "store" in op or op.startswith("ds_write") or op.startswith("s_") or op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane")
    for k, o in enumerate(ops):
        r = regs(o)
        if k == 0 and not is_store:
            d |= r
            if op.startswith("v_fmac") or op.startswith("v_mac") or op.startswith("v_writelane") or "dpp" in t:
                u |= r
        else:
            u |= r
    defs.append(d)
    uses.append(u)
live = set()
prof = [0] * len(ins)
for k in range(len(ins) - 1, -1, -1):
    live -= defs[k]
    live |= uses[k]
    prof[k] = len(live)
mx = max(prof)
at = prof.index(mx)
for k in range(0, len(ins), every):
    print("%5d live %3d  %s" % (k, prof[k], ins[k][:70]))
print("max live %d at instruction %d of %d: %s" % (mx, at, len(ins), ins[at]))
