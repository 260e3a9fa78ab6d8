#!/usr/bin/env python3
"""per launch of one 32-column sweep: HBM bytes (PMC passes: FETCH_SIZE doubled + WRITE_SIZE, per dispatch) and the
kernel's duration in the same runs -> real traffic rate per tree level.  Usage: sweep_levels_report.py fetch.csv write.csv"""
import csv
import sys


def load(path):
    rows = [r for r in csv.DictReader(open(path))
            if any(t in r["Kernel_Name"] for t in ("level_kernel", "thin_kernel", "wave_kernel", "overflow_sum", "v1_assemble"))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


f, w = load(sys.argv[1]), load(sys.argv[2])
assert len(f) == len(w) and len(f) % 10 == 0, (len(f), len(w))
per = len(f) // 10  # tools/pmc_sweep.py runs ten sweeps
f, w = f[-per:], w[-per:]
tf = tw = td = 0.0
print(f"{'kernel':46s} {'workgroups':>10s} {'fetch MB':>9s} {'write MB':>9s} {'us':>7s} {'TB/s':>6s}")
for a, b in zip(f, w):
    assert a["Kernel_Name"] == b["Kernel_Name"]
    fe = float(a["Counter_Value"]) * 2 * 1024 / 1e6
    wr = float(b["Counter_Value"]) * 1024 / 1e6
    dur = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"]) + int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 2e3
    name = a["Kernel_Name"].split("(")[0].replace("void eigd::", "")
    wgs = int(a["Grid_Size"]) // int(a["Workgroup_Size"])
    tf, tw, td = tf + fe, tw + wr, td + dur
    print(f"{name:46s} {wgs:10d} {fe:9.1f} {wr:9.1f} {dur:7.1f} {(fe + wr) / dur:6.2f}")
print(f"{'one sweep':46s} {'':10s} {tf:9.1f} {tw:9.1f} {td:7.1f} {(tf + tw) / td:6.2f}")
