#!/usr/bin/env python3
"""Kernel time of ONE bench step by kernel family, from a rocprofv3 kernel trace of bench.py (development aid).
usage: python tools/step_breakdown.py <dir with *_kernel_trace.csv>      (a step ends with the two elem_bilinear launches)"""
import collections
import csv
import glob
import re
import sys

import os

f = max(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)  # the newest run
rows = list(csv.DictReader(open(f)))
names = [r["Kernel_Name"] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
ends = [i for i, nm in enumerate(names) if "elem_bilinear" in nm][1::2]
a, b = ends[-2] + 1, ends[-1] + 1
tot, cnt = collections.Counter(), collections.Counter()
for i in range(a, b):
    fam = re.sub(r"<.*", "", re.sub(r"\(.*", "", names[i]).replace("void ", "").replace("eigd::", ""))
    tot[fam] += en[i] - st[i]
    cnt[fam] += 1
busy = sum(tot.values())
print(f"one step: wall {(en[b - 1] - st[a]) / 1e6:.2f} ms, kernel busy {busy / 1e6:.2f} ms, {b - a} dispatches")
for fam, t in tot.most_common():
    print(f"{fam:34s} {t / 1e6:8.3f} ms {100 * t / busy:5.1f} %  launches {cnt[fam]}")
