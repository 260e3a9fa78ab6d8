#!/usr/bin/env python3
"""idle time of the GPU in the last FRACTION of a rocprofv3 kernel trace (development aid): gaps between consecutive
kernels by size class, and the kernels that follow the long ones:  trace_gaps.py <dir> [fraction=0.5]"""
import collections
import csv
import glob
import os
import re
import sys

f = max(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1.0 - frac)):]
name = lambda r: re.sub(r"<.*", "", re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("eigd::", ""))
classes = collections.Counter()
after = collections.Counter()
n_after = collections.Counter()
end = int(rows[0]["End_Timestamp"])
for prev, r in zip(rows, rows[1:]):
    g = (int(r["Start_Timestamp"]) - end) / 1e3
    end = max(end, int(r["End_Timestamp"]))
    if g <= 0:
        continue
    key = "<5us" if g < 5 else "5-20us" if g < 20 else "20-100us" if g < 100 else "0.1-1ms" if g < 1000 else ">1ms"
    classes[key] += g
    if g >= 20:
        after[(name(prev), name(r))] += g
        n_after[(name(prev), name(r))] += 1
wall = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{len(rows)} dispatches, wall {wall / 1e3:.1f} ms, idle {sum(classes.values()) / 1e3:.1f} ms")
for k in ("<5us", "5-20us", "20-100us", "0.1-1ms", ">1ms"):
    print(f"  gaps {k:9s} {classes[k] / 1e3:8.2f} ms")
print("long gaps (>= 20 us) by the kernels on both sides:")
for (a, b), t in after.most_common(15):
    print(f"  {t / 1e3:7.2f} ms in {n_after[(a, b)]:4d} gaps   {a} -> {b}")
