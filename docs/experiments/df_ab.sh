#!/bin/bash
# dataflow launches of the upper levels against one launch per level: digests (must be equal) and times
set -o pipefail
mkdir -p gpurun_out
for m in 0 1; do
  EIGD_SWEEP_DATAFLOW=$m timeout -k 10 240 python tools/sweep_digest.py > gpurun_out/df_digest_$m.txt 2>&1 || { echo "digest run $m failed"; tail -20 gpurun_out/df_digest_$m.txt; exit 1; }
done
tail -1 gpurun_out/df_digest_0.txt; tail -1 gpurun_out/df_digest_1.txt
for m in 0 1; do
  EIGD_SWEEP_DATAFLOW=$m DIGEST=1 timeout -k 10 240 python tools/sweep_time.py > gpurun_out/df_time_$m.txt 2>&1 || { echo "time run $m failed"; tail -20 gpurun_out/df_time_$m.txt; exit 1; }
  cat gpurun_out/df_time_$m.txt
done
