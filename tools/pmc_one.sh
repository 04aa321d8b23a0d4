#!/bin/bash
# PMC passes (separate rocprofv3 runs, --kernel-trace only) for one kernel of one command:
#   tools/pmc_one.sh <tag> <kernel-name-substring> <script.py> [args...]      -> gpurun_out/pmc_<tag>.json (per-kernel counter means)
tag=$1; sub=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$tag
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -- python3 "$@" > $R/gpurun_out/pmc_$tag/p$i.log 2>&1
  tail -1 $R/gpurun_out/pmc_$tag/p$i.log
done
python3 $R/tools/collect_pmc.py $R/gpurun_out/pmc_$tag.json "$sub" $R/gpurun_out/pmc_$tag/p* > /dev/null
find $R/gpurun_out/pmc_$tag -name "*.csv" -size +1M -delete
