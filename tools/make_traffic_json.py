#!/usr/bin/env python3
"""profiles/traffic.json from separate rocprofv3 --pmc passes (FETCH_SIZE in one pass, WRITE_SIZE in
another; MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads -> doubled).  Output: average HBM bytes per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if "esa::" not in n:
            continue
        if "conv_s2c32_kernel<1" in n:                # stream kernel template args are <stride, TH, MW>, always 3x3
            key = "conv_mfma<3,1>"
        elif "conv_s2c32_kernel<2" in n:
            key = "conv_mfma<3,2>"
        elif "conv1x1_kernel" in n:
            key = "conv_mfma<1,1>"
        elif "conv_mfma_ring_kernel<1" in n:          # ring kernel template args are <stride, TH, MT>, always 3x3
            key = "conv_mfma<3,1>"
        elif "conv_mfma_ring_kernel<2" in n:
            key = "conv_mfma<3,2>"
        elif "conv_mfma_kernel<3, 1" in n:
            key = "conv_mfma<3,1>"
        elif "conv_mfma_kernel<3, 2" in n:
            key = "conv_mfma<3,2>"
        elif "conv_mfma_kernel<1, 1" in n:
            key = "conv_mfma<1,1>"
        else:
            key = n.split("(")[0].split("::")[-1].split("<")[0].replace("_kernel", "").replace("void ", "").strip()
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 1024 * 2
        write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024
        res[k] = round(fetch + write)
        res[k + "#detail"] = {"fetch_bytes_x2corr": round(fetch), "write_bytes": round(write),
                              "launches_sampled": len(v["FETCH_SIZE"])}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in res.items() if "#" not in k}, indent=1))
