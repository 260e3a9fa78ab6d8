#!/usr/bin/env python3
"""from a rocprofv3 kernel trace of bench.py: per lock-step iteration, time in the sweep, in other kernels and in
host gaps; optionally the kernel sequence of chosen iterations (development aid)"""
import csv
import glob
import re
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
which = [int(a) for a in sys.argv[2:] if not a.startswith("sum=")]
sum_rng = [tuple(int(x) for x in a[4:].split(":")) for a in sys.argv[2:] if a.startswith("sum=")]
totals = defaultdict(float)
tot_gap = 0.0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = []
prev_fwd = False
for i, r in enumerate(rows):
    fwd = any(t in r["Kernel_Name"] for t in ("fwd_level", "fwd_thin", "fwd_wave"))
    if fwd and not prev_fwd:
        starts.append(i)
    prev_fwd = fwd or ("overflow" in r["Kernel_Name"])
print(f"{len(rows)} launches, {len(starts)} sweeps")
for w in range(len(starts)):
    a, b = starts[w], starts[w + 1] if w + 1 < len(starts) else len(rows)
    seq = rows[a:b]
    t0 = int(seq[0]["Start_Timestamp"])
    sweep = other = gaps = 0.0
    prev = t0
    by = defaultdict(float)
    kpt = re.search(r"_kernel<(\d+)", seq[0]["Kernel_Name"]).group(1)
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void eigd::", "").replace("eigd::", "")
        if any(t in name for t in ("level_kernel", "thin_kernel", "wave_kernel", "overflow")):
            sweep += (e - prev) / 1e3
        else:
            other += (e - s) / 1e3
            gaps += max(0, s - prev) / 1e3
            by[re.sub(r"<.*", "", name)] += (e - s) / 1e3
        prev = e
    if any(a <= w < b for a, b in sum_rng):
        totals["(sweep level kernels)"] += sweep
        for kname, v in by.items():
            totals[kname] += v
        tot_gap += gaps
    top = ", ".join(f"{k} {v:.0f}" for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:6])
    print(f"sweep {w:3d} width {kpt}: sweep {sweep:7.1f}  other {other:7.1f}  gaps {gaps:6.1f} us | {top}")
    if w in which or (w - len(starts)) in which:
        prev = t0
        for r in seq:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            name = r["Kernel_Name"].split("(")[0].replace("void eigd::", "")
            print(f"      gap {(s - prev) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f} us  wg {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}  {name}")
            prev = e

if sum_rng:
    print("--- totals over", sum_rng, "(us)")
    for kname, v in sorted(totals.items(), key=lambda kv: -kv[1]):
        print(f"  {v:10.1f}  {kname}")
    print(f"  {tot_gap:10.1f}  (host gaps)")
    print(f"  {sum(totals.values()) + tot_gap:10.1f}  total")
