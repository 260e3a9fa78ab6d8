#!/usr/bin/env python3
"""sum a PMC counter per kernel family from a rocprofv3 --pmc run (counter_collection.csv)"""
import csv
import glob
import re
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
tot, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("eigd::", "")
    name = re.sub(r"<.*", "", name)
    tot[(name, r["Counter_Name"])] += float(r["Counter_Value"])
    cnt[(name, r["Counter_Name"])] += 1
for (name, ctr), v in sorted(tot.items()):
    print(f"{name:28s} {ctr:12s} launches {cnt[(name, ctr)]:6d}  sum {v:16.1f}  per launch {v / cnt[(name, ctr)]:14.1f}")
