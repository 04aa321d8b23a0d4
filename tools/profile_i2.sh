#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rocprofv3 --kernel-trace --output-format csv -d $out/$1 --pmc $2 -- python3 $R/tools/bench_ops.py --i2-only > $out/$1.log 2>&1; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
run sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/bench_ops.py --i2-only > $out/stats.log 2>&1
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
cd $R && python3 tools/collect_pmc.py $out/pmc_summary.json seg_gmr_window $out/fetch $out/write $out/tcp $out/sq1 $out/sq2 > /dev/null
