#!/bin/bash
# Round evidence for profiles/: kernel-trace statistics of the bench command, the two PMC passes over the sweep and
# the SpMV (separate runs: --pmc with --kernel-trace only), the per-level table.  Run on the GPU box from the repo root:
#   tools/collect_profiles.sh r03 [stats|pmc|steppmc|all]
# Results land in gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
tag=${1:-r05}
what=${2:-all}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ "$what" = "stats" ] || [ "$what" = "all" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 3 --warmup 1 \
      --cpu-sample none --no-fd-check --numpy-steps 0 --no-arnoldi-leg --no-scaling-model --no-extras0-leg > $out/profiled_run.json \
      2> $out/profiled_run.log || exit 1
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/bench_c3_kernel_stats.csv
  python3 $root/tools/step_breakdown.py $out/stats > $out/step_breakdown.txt   # (the trace itself is too big to travel back)
  find $out/stats -name "*kernel_trace.csv" -delete
fi
if [ "$what" = "pmc" ] || [ "$what" = "all" ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_sweep_$ctr -- python3 $root/tools/pmc_sweep.py \
        > $out/pmc_sweep_$ctr.log 2>&1 || exit 1
    cp $(find $out/pmc_sweep_$ctr -name "*counter_collection.csv" | head -1) $out/pmc_${ctr}_sweep_coldot.csv
    (cd $root && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_spmv_$ctr -- python3 tools/pmc_spmv.py \
        > $out/pmc_spmv_$ctr.log 2>&1) || exit 1
    cp $(find $out/pmc_spmv_$ctr -name "*counter_collection.csv" | head -1) $out/pmc_${ctr}_spmv_coldot.csv
  done
  # matrix pipe of the sweep: fp64 MFMA instructions issued and the cycles they hold the pipe (compute ceiling of the roofline)
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_sweep_mfma -- \
      python3 $root/tools/pmc_sweep.py > $out/pmc_sweep_mfma.log 2>&1 || exit 1
  cp $(find $out/pmc_sweep_mfma -name "*counter_collection.csv" | head -1) $out/pmc_mfma_sweep.csv
fi
if [ "$what" = "steppmc" ] || [ "$what" = "all" ]; then
  # HBM traffic of a whole step: the bench command under the two counters (separate runs, --kernel-trace only)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_step_$ctr -- python3 $root/bench.py --steps 2 \
        --warmup 1 --cpu-sample none --no-fd-check --numpy-steps 0 --spmv-reps 4 --no-arnoldi-leg --no-scaling-model --no-extras0-leg \
        > $out/pmc_step_$ctr.json \
        2> $out/pmc_step_$ctr.log || exit 1
    cp $(find $out/pmc_step_$ctr -name "*counter_collection.csv" | head -1) $out/pmc_${ctr}_step.csv
  done
  python3 $root/tools/step_traffic_report.py $out/pmc_FETCH_SIZE_step.csv $out/pmc_WRITE_SIZE_step.csv > $out/step_traffic.txt
fi
ls -la $out | head -40
