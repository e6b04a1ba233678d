#!/bin/bash
# usage (on the GPU box): bash tools/prof_bench.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/ + summary on stdout
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o p --output-format csv -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $root/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $root/gpurun_out/prof_$tag.log; exit 1; }
python3 $root/tools/prof_sum.py $root/gpurun_out/prof_$tag/p_kernel_stats.csv 5 12
