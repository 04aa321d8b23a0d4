#!/bin/bash
# PMC passes over the training step for the dense-side kernels (bn_bwd_linear_dw, rowblock_linear): tools/pmc_dense.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag; mkdir -p $R/gpurun_out/pmc_$tag
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-regimes --no-configs > $R/gpurun_out/pmc_$tag/p$i.log 2>&1
  tail -1 $R/gpurun_out/pmc_$tag/p$i.log | cut -c1-200
done
python3 $R/tools/collect_pmc.py $R/gpurun_out/pmc_$tag.json bn_bwd_linear,rowblock_linear,seg_gmr_fast $R/gpurun_out/pmc_$tag/p* > /dev/null
find $R/gpurun_out/pmc_$tag -name "*.csv" -size +1M -delete
