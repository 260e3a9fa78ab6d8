#!/usr/bin/env python3
"""summarise a rocprofv3 kernel trace of tools/sweep_trace.py: per-launch durations of the last sweep of each width"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sw = [r for r in rows if any(t in r["Kernel_Name"] for t in ("level_kernel", "wave_kernel", "thin_kernel", "overflow"))]
# split into sweeps: a sweep starts at a forward kernel that follows a backward kernel
sweeps, cur, prev_bwd = [], [], True
for r in sw:
    fwd = "fwd_" in r["Kernel_Name"] or "overflow" in r["Kernel_Name"]
    if fwd and prev_bwd and cur:
        sweeps.append(cur)
        cur = []
    cur.append(r)
    prev_bwd = not fwd
sweeps.append(cur)
for s in sweeps[2::3]:  # tools/sweep_trace.py runs three sweeps per width: report the last of each
    t0, t1 = int(s[0]["Start_Timestamp"]), int(s[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s)
    print(f"sweep: {len(s)} launches, span {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
    prev = t0
    for r in s:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void eigd::", "")
        print(f"  gap {(a - prev) / 1e3:6.1f}  dur {(b - a) / 1e3:7.1f} us  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} wg  {name}")
        prev = b
