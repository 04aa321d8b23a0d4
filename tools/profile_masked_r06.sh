#!/bin/bash
# round 6: PMC counters of the masked contraction kernel at the config-3 shape INCLUDING the matrix-core counters (SQ_INSTS_MFMA,
# SQ_VALU_MFMA_BUSY_CYCLES: MFMA utilisation from counters, north_star's wording) -- separate --pmc passes, --kernel-trace only
out=$GRAFT_REPO_ROOT/gpurun_out/masked_r06; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/$1 --pmc $2 -- python3 $R/tools/masked_bench.py --reps 3 --only "masked_bmm matrix-core kernel <bf16>" > $out/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
run sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
run sq3 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
cd $R && python3 tools/collect_pmc.py $GRAFT_REPO_ROOT/gpurun_out/masked_r06_pmc.json masked_bmm $out/sq1 $out/sq2 $out/sq3 $out/fetch $out/write $out/tcp > /dev/null
python3 $R/tools/masked_bench.py --reps 20 --only "masked_bmm" > $GRAFT_REPO_ROOT/gpurun_out/masked_r06_bench.json 2>/dev/null
find $out -name "*.csv" -size +1M -delete
tail -2 $out/sq3.log
