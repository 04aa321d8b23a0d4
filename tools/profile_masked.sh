#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/masked_bench.py --reps 20 > $out/masked_bench.json 2> $out/masked_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/masked_bench.py --reps 10 > $out/stats.log 2>&1
run() { rocprofv3 --kernel-trace --output-format csv -d $out/$1 --pmc $2 -- python3 $R/tools/masked_bench.py --reps 3 --only "masked_bmm" > $out/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
run sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
cd $R && python3 tools/collect_pmc.py $out/pmc_summary.json masked_bmm $out/sq1 $out/sq2 $out/fetch $out/write $out/tcp > /dev/null
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python3 tools/bench_ops.py > $out/bench_ops.jsonl 2> $out/bench_ops.err
python3 tools/bench_layers.py > $out/bench_layers.jsonl 2> $out/bench_layers.err
