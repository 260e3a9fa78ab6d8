#!/usr/bin/env python3
"""registers, occupancy and LDS of every kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line
per kernel.  Usage: tools/kernel_resources.py eigd_amd/csrc/factor.hip [substring of the kernel name ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
pats = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z \[\]/]+): (\d+))", line)
    if not m:
        continue
    if m.group(1):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name.split("(")[0].replace("void eigd::", "")}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(2).strip()] = int(m.group(3))
print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'occ':>4s} {'spill':>6s} {'lds':>7s}")
for r in rows:
    if pats and not any(p in r["name"] for p in pats):
        continue
    print(f"{r['name'][:70]:70s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} "
          f"{r.get('Occupancy [waves/SIMD]', 0):4d} {r.get('VGPRs Spill', 0):6d} {r.get('LDS Size [bytes/block]', 0):7d}")
