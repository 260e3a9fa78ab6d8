#!/bin/bash
# occupancy probe of the thin kernels: per-launch times of a 32-column sweep with unused dynamic LDS reserved
# usage: occ_probe.sh fwd|bwd
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
dir=${1:-fwd}
var=EIGD_THIN_LDS; [ "$dir" = bwd ] && var=EIGD_THIN_LDS_BWD
mkdir -p gpurun_out/occ
export TMPDIR=/tmp
: > gpurun_out/occ/summary_$dir.txt
for lds in 0 81920 40960 26000 20000 13000 10000; do
  rm -rf gpurun_out/occ/p$lds
  env $var=$lds timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/occ/p$lds -- python3 tools/sweep_trace.py 32 > gpurun_out/occ/run$lds.log 2>&1 || { tail -5 gpurun_out/occ/run$lds.log; exit 1; }
  f=$(find gpurun_out/occ/p$lds -name '*kernel_trace.csv' | head -1)
  echo "== $var=$lds" >> gpurun_out/occ/summary_$dir.txt
  python3 tools/level_times.py $f 30 | grep -E "${dir}_thin|sum" >> gpurun_out/occ/summary_$dir.txt
  rm -rf gpurun_out/occ/p$lds
done
cat gpurun_out/occ/summary_$dir.txt
