#!/bin/bash
# development aid: the C3 step under environment variants; usage: tools/step_variants.sh out "VAR=val VAR2=val" "..." ...
out=$1; shift
: > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample none --no-fd-check --numpy-steps 0 --spmv-reps 10 2>> $out.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        b=json.loads(line)
        print(json.dumps({'ms_per_step':b['ms_per_step'],'lock_step':b['lock_step'],'it_max':max(b['sibk_iterations']),'it_sum':sum(b['sibk_iterations']),'res':b['accuracy'].get('adjoint_residual_rel_max')}))
" >> $out || exit 1
done
cat $out
