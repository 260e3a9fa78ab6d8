set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh2; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.log
echo c3 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --cpu-sample none --no-fd-check --numpy-steps 0 > $O/profiled_run.json 2> $O/profiled_run.log
echo stats done
python3 bench.py --workload c5 --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.log
echo c5 done
