#!/usr/bin/env python3
"""Static check of the explicit scalar prefetches (csrc/shared/llpf_rbfull.h: RBF_ROW_FETCH / RBF_ROW_READY) in a `hipcc -S` listing:
between an `s_load_dwordx16 s[a:b], ...` that came from inline asm and the next `s_waitcnt lgkmcnt(0)` NO instruction may read a
register of s[a:b] — the compiler does not know the load is still in flight, so a spill (v_writelane), a copy (s_mov) or any
other use it inserted there would read stale data.  usage: inflight_check.py file.s kernel_name_substring   (exit code 1 on a hit)"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
ins = []
in_asm = False
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    ins.append((t.split(";")[0].strip(), in_asm))
rx = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(tok):
    out = set()
    for m in rx.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


inflight = {}          # register -> index of the load
bad = 0
nload = 0
for k, (t, asm) in enumerate(ins):
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    if op == "s_waitcnt" and "lgkmcnt(0)" in t:
        inflight.clear()
        continue
    srcs = set()
    for i, o in enumerate(ops):
        if i == 0 and not (op.startswith("s_cmp") or op.startswith("s_store") or op.startswith("v_writelane") or op.startswith("s_cbranch") or "store" in op):
            if op.startswith("v_writelane"):
                pass
            continue                # destination
        srcs |= sregs(o)
    if op.startswith("v_writelane"):
        srcs |= sregs(ops[1]) if len(ops) > 1 else set()
    hit = srcs & set(inflight)
    if hit:
        bad += 1
        print("instruction %d reads s%s while the load at %d is in flight: %s" % (k, sorted(hit), inflight[min(hit)], t))
    if op.startswith("s_load_dwordx16") and asm:
        nload += 1
        for r in sregs(ops[0]):
            inflight[r] = k
print("%d explicit loads checked, %d hazards" % (nload, bad))
sys.exit(1 if bad else 0)
