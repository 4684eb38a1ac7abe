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
        # key = the function name as the per-launch tables print it (esa::conv_kernel_name): no "void", no
        # namespaces, no argument list; the conv kernels keep their template arguments, the others do not
        fn = n.replace("void ", "").replace("esa::", "").strip()
        depth, cut = 0, len(fn)
        for i, ch in enumerate(fn):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        fn = fn[:cut]
        if fn.startswith(("conv_s2c32_kernel", "conv_s2c32_jobs_kernel", "conv_mfma_ring_kernel", "conv_mfma_kernel", "conv_x6_kernel", "conv_x6_jobs_kernel")):
            key = fn
        else:
            key = fn.split("<")[0].replace("_kernel", "")
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
