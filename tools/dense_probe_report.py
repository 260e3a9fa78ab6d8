#!/usr/bin/env python3
"""per-width kernel averages from a rocprofv3 kernel trace of tools/dense_probe.py (development aid)"""
import csv
import glob
import re
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
phase = None
acc = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    m = re.search(r"spmm_(?:rows|tiled)_kernel<(\d+)>", name)
    if m:
        phase = int(m.group(1))
    if phase is None:
        continue
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("eigd::", "")
    short = re.sub(r"<.*", "", short)
    acc[(phase, short)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (ph, short), v in sorted(acc.items()):
    v = v[len(v) // 4:]  # drop warm-up calls
    print(f"k<={ph:2d} {short:28s} calls {len(v):3d}  avg {sum(v) / len(v):8.1f} us  min {min(v):8.1f}")
