#!/bin/bash
# SQ / GRBM counters of the NT GEMM kernels on one long launch per configuration (effective clock = GRBM_GUI_ACTIVE / 8 / duration
# is only trustworthy on dispatches of ~1 ms and more: M = 262 400 rows).  usage (GPU box): bash tools/pmc_gemm.sh <tag> [cfgs]
tag=${1:-x}
cfgs=${2:-13,15}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_gemm_${tag}_$i -o p --output-format csv -- python3 $R/tools/gemm_one.py 262400 2304 768 $cfgs 3 > $R/gpurun_out/pmc_gemm_${tag}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_gemm_${tag}_$i.log; exit 1; }
done
python3 $R/tools/pmc_gemm_sum.py $R/gpurun_out/pmc_gemm_${tag}_* > $R/gpurun_out/pmc_gemm_${tag}.txt
cat $R/gpurun_out/pmc_gemm_${tag}.txt
