#!/bin/bash
# round-5 evidence in one GPU session: bench line, rocprofv3 stats + PMC traffic of the same command, kernel trace of one step, the
# configs' stats + PMC passes, the captured-slot soak and the slot benchmark
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_bench.sh prof_r05
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r05/trace -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-regimes --no-configs > $R/gpurun_out/prof_r05/trace_line.json 2> $R/gpurun_out/prof_r05/trace.err
cd $R && python3 tools/step_trace.py gpurun_out/prof_r05/trace > gpurun_out/prof_r05/step_trace.txt 2>&1
bash $R/tools/profile_configs.sh
cd $R && timeout 600 python3 tools/soak_fresh.py --captured > gpurun_out/prof_r05/soak_captured.txt 2>&1
timeout 200 python3 tools/slot_bench.py > gpurun_out/prof_r05/slot_bench.json 2> /dev/null
find gpurun_out/prof_r05 -name "*.csv" -size +2M -delete
ls -la gpurun_out/prof_r05
