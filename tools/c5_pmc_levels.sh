#!/bin/bash
# per launch of a 32-column C5 sweep: HBM bytes (two PMC passes) and duration -> gpurun_out/c5l/pmc_levels.txt
root=$(pwd)
out=$root/gpurun_out/c5l
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -- python3 $root/tools/pmc_sweep_c5.py > $out/pmc_$ctr.log 2>&1 || { tail -5 $out/pmc_$ctr.log; exit 1; }
  cp $(find $out/pmc_$ctr -name "*counter_collection.csv" | head -1) $out/pmc_${ctr}_c5.csv
  rm -rf $out/pmc_$ctr
done
python3 $root/tools/sweep_levels_report.py $out/pmc_FETCH_SIZE_c5.csv $out/pmc_WRITE_SIZE_c5.csv > $out/pmc_levels.txt
cat $out/pmc_levels.txt
