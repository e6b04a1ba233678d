#!/bin/bash
# usage (on the GPU box): bash tools/prof_roofline.sh <tag> <commit>
#   kernel traces of the full step and of the encoder alone -> gpurun_out/<tag>_roofline.json (+ kernel stats CSV of the full step)
tag=$1; commit=${2:-unrecorded}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_full -o p --output-format csv -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $root/gpurun_out/prof_${tag}_full.log 2>&1 || { tail -5 $root/gpurun_out/prof_${tag}_full.log; exit 1; }
rocprofv3 --kernel-trace -d $root/gpurun_out/prof_${tag}_enc -o p --output-format csv -- python3 $root/tools/encoder_only.py --steps 3 --warmup 2 > $root/gpurun_out/prof_${tag}_enc.log 2>&1 || { tail -5 $root/gpurun_out/prof_${tag}_enc.log; exit 1; }
pmc=""
[ -f $root/gpurun_out/pmc_${tag}_hbm.json ] && pmc="--pmc $root/gpurun_out/pmc_${tag}_hbm.json"
[ -f $root/gpurun_out/sq_${tag}.json ] && pmc="$pmc --sq $root/gpurun_out/sq_${tag}.json"
python3 $root/tools/roofline_sum.py $root/gpurun_out/prof_${tag}_full/p_kernel_trace.csv 5 $root/gpurun_out/prof_${tag}_enc/p_kernel_trace.csv 5 $commit $root/gpurun_out/${tag}_roofline.json $pmc
python3 $root/tools/prof_sum.py $root/gpurun_out/prof_${tag}_full/p_kernel_stats.csv 5 16 > $root/gpurun_out/${tag}_summary.txt
cp $root/gpurun_out/prof_${tag}_full/p_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv
cat $root/gpurun_out/${tag}_summary.txt
