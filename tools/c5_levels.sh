#!/bin/bash
# per-launch times of the last sweep of the C5 bench (kernel trace)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c5l
export TMPDIR=/tmp
rm -rf gpurun_out/c5l/p
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c5l/p -- python3 bench.py --workload c5 --steps 1 --warmup 1 --cpu-sample none --no-fd-check --numpy-steps 0 > gpurun_out/c5l/run.json 2> gpurun_out/c5l/run.log || { tail -5 gpurun_out/c5l/run.log; exit 1; }
f=$(find gpurun_out/c5l/p -name '*kernel_trace.csv' | head -1)
python3 - $f <<'PY' > gpurun_out/c5l/levels.txt
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if any(t in r["Kernel_Name"] for t in ("level_kernel","thin_kernel","wave_kernel","overflow_sum","subtree","v1_assemble"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the roofline sweeps of the bench come last: find the last forward-leaf launch and print from there
starts=[i for i,r in enumerate(rows) if "fwd_thin_kernel<32" in r["Kernel_Name"] and ", 0, " in r["Kernel_Name"].split("(")[0]]
a=starts[-1]
tot=0
first=None
for i,r in enumerate(rows[a:]):
    if first is None and i>0 and "fwd_thin_kernel" in r["Kernel_Name"] and ", 0, " in r["Kernel_Name"].split("(")[0]:
        first=tot; print(f"-- the 32-column sweep: {tot:.1f} us of kernels")
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; tot+=d
    print(f'{r["Kernel_Name"].split("(")[0].replace("void eigd::",""):50s} {int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]):7d} {d:8.1f} vgpr {r["VGPR_Count"]} scratch {r["Scratch_Size"]}')
print("sum",tot)
PY
rm -rf gpurun_out/c5l/p
cat gpurun_out/c5l/levels.txt
