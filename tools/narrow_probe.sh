#!/bin/bash
# per-launch times of a 4- and an 8-column sweep (kernel trace)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/narrow
export TMPDIR=/tmp
rm -rf gpurun_out/narrow/p
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/narrow/p -- python3 tools/sweep_trace.py 8 4 > gpurun_out/narrow/run.log 2>&1 || { tail -5 gpurun_out/narrow/run.log; exit 1; }
f=$(find gpurun_out/narrow/p -name '*kernel_trace.csv' | head -1)
python3 tools/level_times.py $f 30 > gpurun_out/narrow/k4.txt
python3 - $f <<'PY' > gpurun_out/narrow/k8.txt
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if any(t in r["Kernel_Name"] for t in ("level_kernel","thin_kernel","wave_kernel","overflow_sum","subtree"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[60:90]
tot=0
for r in rows:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; tot+=d
    print(f'{r["Kernel_Name"].split("(")[0].replace("void eigd::",""):46s} {int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]):7d} {d:8.1f} vgpr {r["VGPR_Count"]}')
print("sum",tot)
PY
rm -rf gpurun_out/narrow/p
cat gpurun_out/narrow/k8.txt gpurun_out/narrow/k4.txt
