#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/l45
export TMPDIR=/tmp
for c in 32 48; do
  rm -rf gpurun_out/l45/p
  EIGD_THIN_NS_FWD_KIDS=$c timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/l45/p -- python3 tools/sweep_trace.py 32 > gpurun_out/l45/run.log 2>&1 || { tail -5 gpurun_out/l45/run.log; exit 1; }
  f=$(find gpurun_out/l45/p -name '*kernel_trace.csv' | head -1)
  echo "== EIGD_THIN_NS_FWD_KIDS=$c"
  python3 tools/level_times.py $f 30 | head -8
  rm -rf gpurun_out/l45/p
done
