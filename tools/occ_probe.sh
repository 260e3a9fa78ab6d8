#!/bin/bash
# occupancy probe of the thin forward kernels: per-launch times of a 32-column sweep with unused dynamic LDS reserved
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/occ
export TMPDIR=/tmp
for lds in 0 81920 40960 26000; do
  rm -rf gpurun_out/occ/p$lds
  EIGD_THIN_LDS=$lds timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/occ/p$lds -- python3 tools/sweep_trace.py 32 > gpurun_out/occ/run$lds.log 2>&1 || { tail -5 gpurun_out/occ/run$lds.log; exit 1; }
  f=$(find gpurun_out/occ/p$lds -name '*kernel_trace.csv' | head -1)
  echo "== EIGD_THIN_LDS=$lds" >> gpurun_out/occ/summary.txt
  python3 tools/level_times.py $f 30 | grep -E "fwd_thin|sum" >> gpurun_out/occ/summary.txt
  rm -rf gpurun_out/occ/p$lds
done
cat gpurun_out/occ/summary.txt
