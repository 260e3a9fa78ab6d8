#!/bin/bash
# round evidence from the final tree in one call: kernel statistics + per-step breakdown, the driver's bench line,
# the C5 line, emulated ranks.  usage (GPU box, repo root): tools/final_evidence.sh r03
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
bash tools/collect_profiles.sh $tag stats > $out/collect_stats.log 2>&1 || { tail -5 $out/collect_stats.log; exit 1; }
python3 tools/step_breakdown.py $out/stats > $out/step_breakdown.txt || exit 1
rm -rf $out/stats
cd $root
echo "[evidence] bench c3" 
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err || { tail -5 $out/bench_c3.err; exit 1; }
echo "[evidence] emulated ranks"
bash tools/rank_variants.sh $out/emulated_ranks_2.txt 1/2 EIGD_X=0 > /dev/null || exit 1
bash tools/rank_variants.sh $out/emulated_ranks_4.txt 3/4 EIGD_X=0 > /dev/null || exit 1
bash tools/rank_variants.sh $out/emulated_ranks_8.txt 7/8 EIGD_X=0 > /dev/null || exit 1
cat $out/emulated_ranks_2.txt $out/emulated_ranks_4.txt $out/emulated_ranks_8.txt > $out/emulated_ranks.txt
echo "[evidence] bench c5"
timeout -k 10 700 python3 bench.py --workload c5 --steps 3 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err || { tail -5 $out/bench_c5.err; exit 1; }
python3 - $out <<'PY'
import json, sys
for name in ("bench_c3", "bench_c5"):
    d = json.loads(open(f"{sys.argv[1]}/{name}.json").read().strip().splitlines()[-1])
    print(name, {k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["us_per_launch"], d["roofline"]["frac"], d["roofline"].get("traffic"))
PY
cat $out/emulated_ranks.txt | grep -E "^==|^[0-9]" | cut -c1-80
