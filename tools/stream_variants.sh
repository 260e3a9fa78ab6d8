#!/bin/bash
# step time of the headline case with the mode groups of the lock-step solver on one / several HIP streams
# (two-step form inside every group), cyclic and contiguous split; and with more extra pairs deflated
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=gpurun_out/stream_variants.txt
: > $OUT
run() {
  echo "== $*" | tee -a $OUT
  env "$@" timeout -k 10 400 python bench.py --steps 5 --warmup 2 --cpu-sample none --no-fd-check --numpy-steps 0 $EXTRA 2>gpurun_out/sv_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step')}, d.get('preamble_s',{}).get('eigensolve_repeat_s'), d.get('lock_step'))" | tee -a $OUT || { tail -5 gpurun_out/sv_err.log | tee -a $OUT; return 1; }
}
EXTRA="" run EIGD_NOP=1 &&
EXTRA="--streams 2" run EIGD_STREAM_SPLIT=cyclic &&
EXTRA="--streams 2" run EIGD_STREAM_SPLIT=block &&
EXTRA="--streams 4" run EIGD_STREAM_SPLIT=cyclic
