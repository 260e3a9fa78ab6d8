#!/usr/bin/env python3
"""kernel time by family over the last FRACTION of a rocprofv3 kernel trace (development aid):
trace_families.py <dir> [fraction=0.5]"""
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
tot, cnt = collections.Counter(), collections.Counter()
for r in rows:
    fam = re.sub(r"<.*", "", re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("eigd::", ""))
    tot[fam] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[fam] += 1
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(tot.values())
print(f"{len(rows)} dispatches, wall {wall / 1e6:.1f} ms, kernel busy {busy / 1e6:.1f} ms")
for fam, t in tot.most_common(25):
    print(f"{fam:34s} {t / 1e6:8.3f} ms {100 * t / busy:5.1f} %  launches {cnt[fam]}")
