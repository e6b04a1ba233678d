#!/bin/bash
# usage (on the GPU box): bash tools/prof_config5.sh <tag>  -> gpurun_out/<tag>_config5_{default,swin,pftn}_summary.txt (+ JSON lines)
#   rocprofv3 kernel stats of the three config-5 benches (synthetic stages / Swin-small end to end / the reference's PromptFTN)
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in default swin pftn; do
  case $v in default) flag="";; swin) flag="--swin";; pftn) flag="--prompt-ftn --batch 8";; esac
  rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_c5_$v -o p --output-format csv -- python3 $root/tools/bench_config5.py --steps 3 --warmup 2 $flag > $root/gpurun_out/${tag}_config5_$v.json 2> $root/gpurun_out/${tag}_config5_$v.err || { tail -5 $root/gpurun_out/${tag}_config5_$v.err; exit 1; }
  python3 $root/tools/prof_sum.py $root/gpurun_out/prof_${tag}_c5_$v/p_kernel_stats.csv 5 24 > $root/gpurun_out/${tag}_config5_${v}_summary.txt
  tail -1 $root/gpurun_out/${tag}_config5_$v.json | cut -c1-300
  cat $root/gpurun_out/${tag}_config5_${v}_summary.txt
done
