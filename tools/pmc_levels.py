#!/usr/bin/env python3
"""per launch of the last sweep in a set of rocprofv3 --pmc passes (development aid): pmc_levels.py dir_with_passes"""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

tab = OrderedDict()
names = []
for path in sorted(glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv")):
    per = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if not any(t in r["Kernel_Name"] for t in ("level_kernel", "thin_kernel", "wave_kernel")):
            continue
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        per[int(r["Dispatch_Id"])]["_k"] = r["Kernel_Name"].split("(")[0].replace("void eigd::", "")
        per[int(r["Dispatch_Id"])]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        per[int(r["Dispatch_Id"])]["_wg"] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    ids = sorted(per)
    n = len(ids) // 10
    for j, i in enumerate(ids[-n:]):
        row = tab.setdefault(j, {"k": per[i]["_k"], "wg": per[i]["_wg"], "us": per[i]["_us"]})
        for c, v in per[i].items():
            if not c.startswith("_"):
                row[c] = v
                if c not in names:
                    names.append(c)
print(f"{'kernel':40s} {'wgs':>6s} {'us':>6s} " + " ".join(f"{c[:22]:>22s}" for c in names))
for j, row in tab.items():
    print(f"{row['k'][:40]:40s} {row['wg']:6d} {row['us']:6.1f} " + " ".join(f"{row.get(c, float('nan')):22.4g}" for c in names))
