#!/usr/bin/env python3
"""development aid: the idle gaps of the GPU (>= MIN us) inside the last step of a rocprofv3 kernel trace of bench.py,
with the kernels on both sides:  step_gaps.py <dir> [min_us=15]"""
import csv
import glob
import os
import re
import sys

f = max(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"<.*", "", re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("eigd::", ""))
# a step ends with the derivative callbacks (elem_bilinear); the last step starts behind the last but one group of them
eb = [i for i, r in enumerate(rows) if name(r) == "elem_bilinear_kernel"]
groups = [eb[0]]
for a, b in zip(eb, eb[1:]):
    if b - a > 50:
        groups.append(b)
start = groups[-2] + 2 if len(groups) >= 2 else 0
rows = rows[start:eb[-1] + 1]
end = int(rows[0]["End_Timestamp"])
tot = 0.0
for prev, r in zip(rows, rows[1:]):
    g = (int(r["Start_Timestamp"]) - end) / 1e3
    end = max(end, int(r["End_Timestamp"]))
    if g > 0:
        tot += g
    if g >= lim:
        print(f"{g:8.1f} us  at {(int(r['Start_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6:7.2f} ms   {name(prev)} -> {name(r)}")
print(f"{len(rows)} dispatches, wall {(end - int(rows[0]['Start_Timestamp'])) / 1e6:.2f} ms, idle {tot / 1e3:.2f} ms")
