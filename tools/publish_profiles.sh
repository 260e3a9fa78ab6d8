#!/bin/bash
# Copy what tools/collect_profiles.sh left under gpurun_out/<tag>/ into profiles/ and rebuild profiles/<tag>_traffic.json
# (run here, after the gpurun call has merged gpurun_out/ back):   tools/publish_profiles.sh r03 [stats|pmc|steppmc|all]
tag=${1:-r05}
what=${2:-all}
src=gpurun_out/$tag
if [ "$what" = "stats" ] || [ "$what" = "all" ]; then
  cp $src/bench_c3_kernel_stats.csv profiles/${tag}_bench_c3_kernel_stats.csv
  cp $src/profiled_run.json profiles/${tag}_bench_c3_profiled_run.json
  cp $src/step_breakdown.txt profiles/${tag}_step_breakdown.txt
fi
if [ "$what" = "steppmc" ] || [ "$what" = "all" ]; then
  cp $src/step_traffic.txt profiles/${tag}_step_traffic.txt
fi
if [ "$what" = "pmc" ] || [ "$what" = "all" ]; then
  cp $src/pmc_FETCH_SIZE_sweep_coldot.csv profiles/${tag}_pmc_fetch_sweep_coldot.csv
  cp $src/pmc_WRITE_SIZE_sweep_coldot.csv profiles/${tag}_pmc_write_sweep_coldot.csv
  cp $src/pmc_FETCH_SIZE_spmv_coldot.csv profiles/${tag}_pmc_fetch_spmv_coldot.csv
  cp $src/pmc_WRITE_SIZE_spmv_coldot.csv profiles/${tag}_pmc_write_spmv_coldot.csv
  cp $src/pmc_mfma_sweep.csv profiles/${tag}_pmc_mfma_sweep.csv
  rm -f profiles/${tag}_traffic.json
  cal=511121408   # bytes one calibration launch reads: two n x 32 blocks of doubles, n = 998 284
  python3 tools/pmc_report.py --json profiles/${tag}_traffic.json --entry sweep_k32_c3 \
      --fetch profiles/${tag}_pmc_fetch_sweep_coldot.csv --write profiles/${tag}_pmc_write_sweep_coldot.csv \
      --kernels 'fwd_|bwd_|v1_assemble' --units 10 --sources factor.hip --calib-bytes $cal --mfma profiles/${tag}_pmc_mfma_sweep.csv > /dev/null
  python3 tools/pmc_report.py --json profiles/${tag}_traffic.json --entry spmv_c3 \
      --fetch profiles/${tag}_pmc_fetch_spmv_coldot.csv --write profiles/${tag}_pmc_write_spmv_coldot.csv \
      --kernels 'spmv_stream' --units 20 --sources sparse.hip --calib-bytes $cal > /dev/null
  python3 tools/pmc_report.py --json profiles/${tag}_traffic.json --entry spmm_k32_c3 \
      --fetch profiles/${tag}_pmc_fetch_spmv_coldot.csv --write profiles/${tag}_pmc_write_spmv_coldot.csv \
      --kernels 'spmm_tiled' --units 20 --sources sparse.hip --calib-bytes $cal > /dev/null
  python3 tools/sweep_levels_report.py profiles/${tag}_pmc_fetch_sweep_coldot.csv profiles/${tag}_pmc_write_sweep_coldot.csv \
      > profiles/${tag}_sweep_levels.txt
  python3 -c "
import json
d = json.load(open('profiles/${tag}_traffic.json'))
for k, v in d.items():
    print(k, v['traffic_bytes_per_launch'], v.get('calibration', {}).get('ratio_2x_fetch_to_known_bytes'))"
fi
