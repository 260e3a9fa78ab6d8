#!/bin/bash
# development aid: the C4 "standard" full-size test under several solver settings (which one moves the adjoint residual)
out=$1; shift
: > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 200 python -m pytest tests/test_gpu_path.py -m gpu -q -x -k "c4_full_size and standard" 2>&1 | grep -E "passed|failed|AssertionError: \(array|^E  +[0-9]" | cut -c1-400 >> $out
done
cat $out
