#!/bin/bash
# kernel trace of sweeps (widths given as arguments) with the dataflow launches on
root=$(pwd); out=$root/gpurun_out/trace_${TAG:-df}
mkdir -p $out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/tools/sweep_trace.py "$@" > $out/run.log 2>&1 || { tail -20 $out/run.log; exit 1; }
python3 $root/tools/sweep_trace_report.py $out > $root/gpurun_out/trace_${TAG:-df}.txt
find $out -name "*.csv" -delete
cat $root/gpurun_out/trace_${TAG:-df}.txt
