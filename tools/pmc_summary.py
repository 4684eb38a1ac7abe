#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per kernel (grouped by name + grid size), mean counter values."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "pass*", "*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        name = name.replace("esa::(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        key = (name, r.get("Grid_Size", "?"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "pass1" in f and r["Counter_Name"] == "SQ_WAVES":
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
keys = [k for k in agg if any(s in k[0] for s in ("conv_", "conv1x1", "fuse", "stem", "final", "keypoints", "head", "bblock", "cbam", "crops"))]
keys.sort(key=lambda k: -sum(dur.get(k, [0])))
for k in keys:
    d = dur.get(k, [0])
    print(f"\n== {k[0]} grid={k[1]} launches={len(d)} avg_us={sum(d)/max(1,len(d)):.1f}")
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    for n in sorted(c):
        print(f"   {n:28s} {c[n]:16.0f}")
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        wc = c["SQ_WAVE_CYCLES"]
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
            if n in c:
                print(f"   {n}/WAVE_CYCLES = {c[n]/wc:.3f}")
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   LDS conflict share = {c['SQ_LDS_BANK_CONFLICT']/c['SQ_LDS_IDX_ACTIVE']:.3f}")
    if "TCC_HIT_sum" in c:
        print(f"   L2 hit rate = {c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum']+1e-9):.3f}")
    if "FETCH_SIZE" in c:
        print(f"   FETCH bytes (x2 gfx950 corr) = {c['FETCH_SIZE']*1024*2/1e6:.1f} MB   WRITE bytes = {c.get('WRITE_SIZE',0)*1024/1e6:.1f} MB")
