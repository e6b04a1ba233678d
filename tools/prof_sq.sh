#!/bin/bash
# SQ / GRBM counter evidence for the step's kernel families at HEAD: rocprofv3 counter passes (kernel trace
# only beside --pmc; the program directly after `--`) over `bench.py --steps 2 --warmup 1 --no-cpu-baseline`.
# usage (GPU box): bash tools/prof_sq.sh <tag> [script and args: default bench.py --steps 2 --warmup 1 --no-cpu-baseline]   -> gpurun_out/sq_<tag>.txt (+ .json)
tag=${1:-x}
shift
if [ $# -gt 0 ]; then set -- $GRAFT_REPO_ROOT/"$@"; fi   # (the runs start in /tmp: a script path is taken relative to the repository)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/sq_${tag}_$i -o p --output-format csv -- python3 ${@:-$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline} > $R/gpurun_out/sq_${tag}_$i.log 2>&1 || { tail -5 $R/gpurun_out/sq_${tag}_$i.log; exit 1; }
done
python3 $R/tools/prof_sq_sum.py $R/gpurun_out/sq_${tag} $R/gpurun_out/sq_${tag}_* > $R/gpurun_out/sq_${tag}.txt
cat $R/gpurun_out/sq_${tag}.txt
