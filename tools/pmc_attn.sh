#!/bin/bash
# SQ stall / LDS / matrix-pipe counters of the attention kernels (one rocprofv3 --pmc pass per counter group, kernel trace only).
# usage (on the GPU box): bash tools/pmc_attn.sh <tag> [python script + args]   -> gpurun_out/pmc_attn_<tag>.txt
tag=${1:-x}
shift
cmd=${@:-tools/attn_bwd_ab.py --shapes vit --rounds 1 --iters 2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  ( cd $R && exec true )
  rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_attn_${tag}_$i -o p --output-format csv -- python3 $R/${cmd%% *} ${cmd#* } > $R/gpurun_out/pmc_attn_${tag}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_attn_${tag}_$i.log; exit 1; }
done
python3 $R/tools/pmc_attn_sum.py $R/gpurun_out/pmc_attn_${tag}_* > $R/gpurun_out/pmc_attn_${tag}.txt
cat $R/gpurun_out/pmc_attn_${tag}.txt
