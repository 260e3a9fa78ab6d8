#!/usr/bin/env python3
"""development aid: per-launch times of 32-column sweeps of several factors of the same matrix in one process, from a
rocprofv3 kernel trace of tools/pre_ab_probe.py (THRS=512,512,512 WIDTHS=32): which launches carry the spread between
factors?   placement_probe.py <trace dir> <number of factors>"""
import csv
import glob
import os
import re
import sys

import numpy as np

f = max(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
nfac = int(sys.argv[2])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(t in r["Kernel_Name"] for t in ("level_kernel", "thin_kernel", "v1_assemble"))]
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void eigd::", "")
# a sweep starts with the forward leaf launch
starts = [i for i, r in enumerate(rows) if "fwd_thin_kernel<32" in r["Kernel_Name"] and ", 0, " in name(r)]
L = starts[1] - starts[0]
sweeps = [rows[a:a + L] for a in starts if len(rows[a:a + L]) == L]
# the probe runs 6 rounds x nfac factors x 22 sweeps each (2 warm-up + 20 timed), factor by factor
per = 22
times = {i: [] for i in range(nfac)}
for blk in range(len(sweeps) // per):
    fac = blk % nfac
    for sw in sweeps[blk * per + 2: (blk + 1) * per]:
        times[fac].append([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sw])
med = {i: np.median(np.array(times[i]), axis=0) for i in range(nfac)}
print(f"{'launch':52s} " + " ".join(f"fac{i:>2d}" for i in range(nfac)) + "   spread")
for j, r in enumerate(sweeps[0]):
    v = [med[i][j] for i in range(nfac)]
    print(f"{name(r):52s} " + " ".join(f"{x:6.1f}" for x in v) + f"   {max(v) - min(v):5.1f}")
tot = [med[i].sum() for i in range(nfac)]
print(f"{'sum':52s} " + " ".join(f"{x:6.0f}" for x in tot) + f"   {max(tot) - min(tot):5.1f}")
