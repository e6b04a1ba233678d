#!/bin/bash
# SQ stall / LDS counters of the attention kernels over tools/attn_bench.py (one rocprofv3 --pmc pass per counter group).
# usage (on the GPU box): bash tools/pmc_attn.sh <tag>   -> gpurun_out/pmc_attn_<tag>.txt
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_attn_${tag}_$i -o p --output-format csv -- python3 $R/tools/attn_bench.py --iters 3 --rounds 1 --shapes vit > $R/gpurun_out/pmc_attn_${tag}_$i.log 2>&1 || exit 1
done
python3 $R/tools/pmc_attn_sum.py $R/gpurun_out/pmc_attn_${tag}_* > $R/gpurun_out/pmc_attn_${tag}.txt
cat $R/gpurun_out/pmc_attn_${tag}.txt
