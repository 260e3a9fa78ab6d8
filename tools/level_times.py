#!/usr/bin/env python3
"""per launch of one 32-column sweep from a rocprofv3 kernel trace (development aid): level_times.py kernel_trace.csv [launches per sweep]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))
        if any(t in r["Kernel_Name"] for t in ("level_kernel", "thin_kernel", "wave_kernel", "overflow_sum", "subtree"))]
rows.sort(key=lambda r: int(r["Dispatch_Id"]) if "Dispatch_Id" in r else int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 10   # launches of one sweep
rows = rows[-per:]
tot = 0.0
for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    name = r["Kernel_Name"].split("(")[0].replace("void eigd::", "")
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    print(f"{name:46s} {wgs:7d} {dur:8.1f}  lds {r['LDS_Block_Size']} vgpr {r['VGPR_Count']}+{r['Accum_VGPR_Count']} scratch {r['Scratch_Size']}")
print(f"{'sum':46s} {'':7s} {tot:8.1f}")
