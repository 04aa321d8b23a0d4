#!/bin/bash
# round-6 evidence in one GPU session: bench line, rocprofv3 stats + PMC traffic of the same command (fused forward, fused backward),
# kernel trace of one step, PMC counters of the fused backward kernel, the fused backward A/B
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_bench.sh prof_r06
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r06/trace -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-regimes --no-configs > $R/gpurun_out/prof_r06/trace_line.json 2> $R/gpurun_out/prof_r06/trace.err
cd $R && python3 tools/step_trace.py gpurun_out/prof_r06/trace > gpurun_out/prof_r06/step_trace.txt 2>&1
bash $R/tools/pmc_one.sh dual_r06 seg_dual_kernel $R/tools/dual_one.py res 5 > /dev/null 2>&1
bash $R/tools/pmc_one.sh dual_tg_r06 seg_dual_kernel $R/tools/dual_one.py tg 5 > /dev/null 2>&1
bash $R/tools/dual_step_ab.sh > gpurun_out/prof_r06/dual_ab.txt 2>&1
python3 tools/dual_bwd_ab.py >> gpurun_out/prof_r06/dual_ab.txt 2>/dev/null
find gpurun_out/prof_r06 -name "*.csv" -size +2M -delete
ls -la gpurun_out/prof_r06
