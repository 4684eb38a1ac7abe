#!/usr/bin/env python3
"""Instruction mix of one kernel of a `hipcc -S` listing, segment by segment (segments end at labels and
s_barriers): how many MFMA / VALU / LDS / VMEM / wait instructions each part of the loop issues.
usage: tools/isa_mix.py FILE.s SUBSTRING_OF_MANGLED_NAME [min_instructions_per_segment]"""
import collections
import sys

path, key = sys.argv[1], sys.argv[2]
minseg = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if ".globl" in l and key in l)
end = next(i for i, l in enumerate(lines) if l.startswith(".Lfunc_end") and i > start)


def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr") or op.startswith("v_"): return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith(("buffer_load", "global_load")): return "vmem_ld"
    if op.startswith(("buffer_store", "global_store")): return "vmem_st"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"


segs, cur, total = [], collections.Counter(), collections.Counter()
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if t.startswith(".LBB") or t.startswith("s_barrier"):
        segs.append((dict(cur), t.split()[0]))
        cur = collections.Counter()
        if t.startswith("s_barrier"):
            total["barrier"] += 1
        continue
    if t.startswith("."):
        continue
    k = kind(t.split()[0])
    cur[k] += 1
    total[k] += 1
segs.append((dict(cur), "end"))
for c, why in segs:
    if sum(c.values()) >= minseg:
        print(f"{why:14s}", " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("TOTAL", dict(total))
