#!/usr/bin/env python3
"""development aid: the kernels of the first MS milliseconds (and the last MS) of the last step in a rocprofv3 kernel trace
of bench.py, with durations and the idle gap in front of each:  step_head.py <dir> [ms=4]"""
import csv
import glob
import os
import re
import sys

f = max(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("eigd::", "")[:60]
eb = [i for i, r in enumerate(rows) if "elem_bilinear_kernel" in r["Kernel_Name"]]
groups = [eb[0]]
for a, b in zip(eb, eb[1:]):
    if b - a > 50:
        groups.append(b)
rows = rows[groups[-2] + 2:eb[-1] + 1]
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    at = (s - t0) / 1e6
    if at <= ms or (t1 - s) / 1e6 <= ms:
        print(f"{at:7.3f} ms  gap {max(0, s - end) / 1e3:7.1f} us  dur {(e - s) / 1e3:7.1f} us  {name(r)}")
    elif abs(at - ms) < 0.05:
        print("   ...")
    end = max(end, e)
