"""usage: x6_opbench_report.py <kernel_trace.csv>  — pairs the conv_x6 launches of tools/x6_opbench.py with its cases."""
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from x6_opbench import CASES  # noqa: E402  (importing runs nothing: guarded below)

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_x6_kernel" in r["Kernel_Name"]]
assert len(rows) == 3 * len(CASES), (len(rows), len(CASES))
for i, (n, cin, cout, h, w, k, s) in enumerate(CASES):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[3 * i:3 * i + 3]]
    oh, ow = ((h + 1) // 2, (w + 1) // 2) if s == 2 else (h, w)
    fl = 2.0 * n * oh * ow * cout * cin * k * k
    us = min(d) / 1e3
    name = rows[3 * i]["Kernel_Name"].split("conv_x6_kernel")[1].split(">")[0] + ">"
    print(f"n={n:4d} {cin:3d}->{cout:3d} {h:3d}x{w:<3d} k{k} s{s}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF  ({fl / us / 1e6 / 416.7:.2f})  {name}")
