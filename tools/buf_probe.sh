#!/bin/bash
# thin forward kernels with raw buffer accesses against the selected-address form: digests (bitwise), sweep times, per-launch times
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/buf
export TMPDIR=/tmp
: > gpurun_out/buf/summary.txt
for b in 0 1; do
  echo "== EIGD_THIN_BUF=$b" >> gpurun_out/buf/summary.txt
  EIGD_THIN_BUF=$b DIGEST=1 timeout -k 10 300 python3 tools/sweep_time.py >> gpurun_out/buf/summary.txt 2> gpurun_out/buf/err$b.log || { tail -5 gpurun_out/buf/err$b.log; exit 1; }
done
for b in 0 1; do
  rm -rf gpurun_out/buf/p$b
  EIGD_THIN_BUF=$b timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/buf/p$b -- python3 tools/sweep_trace.py 32 16 > gpurun_out/buf/run$b.log 2>&1 || { tail -5 gpurun_out/buf/run$b.log; exit 1; }
  f=$(find gpurun_out/buf/p$b -name '*kernel_trace.csv' | head -1)
  echo "== trace EIGD_THIN_BUF=$b (last sweep: 16 columns; the one before: 32)" >> gpurun_out/buf/summary.txt
  python3 tools/level_times.py $f 180 | grep -E "fwd_thin" | awk 'NR%30<=4 || 1' | tail -24 >> gpurun_out/buf/summary.txt
  rm -rf gpurun_out/buf/p$b
done
cat gpurun_out/buf/summary.txt
