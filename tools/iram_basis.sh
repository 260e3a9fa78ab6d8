#!/bin/bash
out=$1; shift
: > $out
for b in "$@"; do
  echo "== basis $b" >> $out
  EIGD_IRAM_BASIS=$b timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample none --no-fd-check --numpy-steps 0 --spmv-reps 10 2>> $out.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        b=json.loads(line)
        print(json.dumps({'ms_per_step':b['ms_per_step'],'eigensolve_s':b['preamble_s']['eigensolve_s'],'eig':b['eigensolver'],'it_max':max(b['sibk_iterations']),'it_sum':sum(b['sibk_iterations']),'design_point_s':b['design_point_s']}))
" >> $out || exit 1
done
cat $out
