#!/bin/bash
# kernel trace of one steady-state training step of bench.py (every launch in order): tools/step_trace.py --list
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-trace}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-regimes --no-configs ${@:2} > $out/trace_line.json 2> $out/trace.err
cd $R && python3 tools/step_trace.py $out/trace --list > $out/step_list.txt 2>&1
python3 tools/step_trace.py $out/trace > $out/step_trace.txt 2>&1
find $out -name "*.csv" -size +2M -delete
