#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=gpurun_out/buf_bench.txt
: > $OUT
for b in 0 1 0 1; do
  echo "== EIGD_THIN_BUF=$b" | tee -a $OUT
  EIGD_THIN_BUF=$b timeout -k 10 400 python bench.py --steps 5 --warmup 2 --cpu-sample none --no-fd-check --numpy-steps 0 2>gpurun_out/bb_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step')}, d['roofline']['us_per_launch'], d['roofline']['frac'])" | tee -a $OUT || { tail -5 gpurun_out/bb_err.log | tee -a $OUT; exit 1; }
done
