#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $out/bench_line.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-regimes --no-configs > $out/bench_under_rocprof.json 2> $out/stats.err
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $out/fetch --pmc FETCH_SIZE -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-regimes --no-configs > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/write --pmc WRITE_SIZE -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-regimes --no-configs > $out/write.log 2>&1
cd $R && python3 tools/collect_traffic.py $out/fetch $out/write $out/kernel_stats.csv $out/traffic.json 8192 128 bf16 fused > $out/traffic.log 2>&1
python3 tools/collect_traffic.py $out/fetch $out/write $out/kernel_stats.csv $out/traffic_fast.json 8192 128 bf16 fast >> $out/traffic.log 2>&1
python3 tools/collect_traffic.py $out/fetch $out/write $out/kernel_stats.csv $out/traffic_dual.json 8192 128 bf16 dual >> $out/traffic.log 2>&1
