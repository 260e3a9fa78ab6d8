#!/bin/bash
# development aid: the C3 bench under several eigensolver plans (block size, extra converged pairs)
# usage: tools/iram_sweep.sh "1:0 8:0 8:16 8:32" outfile
out=$2
: > $out
for cfg in $1; do
  p=${cfg%%:*}; x=${cfg##*:}
  echo "== block $p extra $x" >> $out
  EIGD_IRAM_BLOCK=$p EIGD_IRAM_EXTRA=$x EIGD_TRACE_IRAM=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample none --no-fd-check --numpy-steps 0 --spmv-reps 20 2>> $out.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        b=json.loads(line)
        print(json.dumps({'ms_per_step':b['ms_per_step'],'eigensolve_s':b['preamble_s']['eigensolve_s'],'eig':b['eigensolver'],'it':b['sibk_iterations'],'sweeps_per_step':b['factor_sweeps_per_step'],'res':b['accuracy'].get('adjoint_residual_rel_max'),'design_point_s':b['design_point_s']}))
    else: print(line.rstrip())
" >> $out || exit 1
done
