#!/bin/bash
# development aid: emulated rank shares under environment variants; usage: tools/rank_variants.sh out "r/P" "VAR=val" ...
out=$1; rp=$2; shift; shift
: > $out
for cfg in "$@"; do
  echo "== rank $rp $cfg" >> $out
  env $cfg timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-sample none --no-fd-check --numpy-steps 0 --spmv-reps 10 --emulate-rank $rp 2>> $out.err | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        b=json.loads(line); print(b['ms_per_step'], b['sibk_iterations'], b['lock_step'])
" >> $out || exit 1
done
cat $out
