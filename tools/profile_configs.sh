#!/bin/bash
# rocprofv3 evidence for bench.py's `configs` object (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the whole of tools/bench_configs.py  -> gpurun_out/prof_configs/stats (per-kernel average durations)
#   2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE / cache hit counters) of `bench_configs.py --kernels-only`
# The program stands directly after `--`; counters are collected with --kernel-trace only (no other trace domain).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_configs
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_configs.py > $O/line_under_rocprof.json 2> $O/stats.log
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc$i -- python3 $R/tools/bench_configs.py --kernels-only > $O/pmc$i.json 2> $O/pmc$i.log
done
python3 $R/tools/collect_config_traffic.py $O $R/gpurun_out/configs_traffic.json
find $O -name "*.csv" -size +2M -delete
