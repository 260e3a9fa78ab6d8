#!/bin/bash
# timing experiment: thin forward kernels with every A-operand load redirected to one contiguous 512-byte piece
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/coal
export TMPDIR=/tmp
cp eigd_amd/lib/libeigd_hip.so /tmp/lib_ref.so
: > gpurun_out/coal/summary.txt
for v in ref exp ref exp; do
  if [ $v = exp ]; then cp gpurun_exp_lib.so eigd_amd/lib/libeigd_hip.so; else cp /tmp/lib_ref.so eigd_amd/lib/libeigd_hip.so; fi
  rm -rf gpurun_out/coal/p
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/coal/p -- python3 tools/sweep_trace.py 32 > gpurun_out/coal/run_$v.log 2>&1 || { tail -5 gpurun_out/coal/run_$v.log; cp /tmp/lib_ref.so eigd_amd/lib/libeigd_hip.so; exit 1; }
  f=$(find gpurun_out/coal/p -name '*kernel_trace.csv' | head -1)
  echo "== $v" >> gpurun_out/coal/summary.txt
  python3 tools/level_times.py $f 30 | grep -E "${PAT:-fwd_thin}|sum" >> gpurun_out/coal/summary.txt
  rm -rf gpurun_out/coal/p
done
cp /tmp/lib_ref.so eigd_amd/lib/libeigd_hip.so
cat gpurun_out/coal/summary.txt
