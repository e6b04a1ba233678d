#!/bin/bash
# Per-kernel time, HBM bytes and achieved bandwidth of ANY bench command (configs 4 and 5: VERDICT r4 item 7).
# usage (GPU box): bash tools/prof_config_hbm.sh <tag> <steps+warmup> <python script and args...>
#   three rocprofv3 runs of the same command: kernel trace (durations), --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, kernel trace only
#   beside --pmc; the program directly after `--`), then tools/hbm_table.py -> gpurun_out/<tag>_hbm_table.txt
tag=$1; n=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
script=$root/$1; shift   # (the runs start in /tmp: the script path is taken relative to the repository)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_t -o p --output-format csv -- python3 $script "$@" > $root/gpurun_out/${tag}_trace.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_trace.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $c --kernel-trace -d $root/gpurun_out/prof_${tag}_$d -o f --output-format csv -- python3 $script "$@" > $root/gpurun_out/${tag}_$d.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_$d.log; exit 1; }
done
python3 $root/tools/hbm_table.py $root/gpurun_out/prof_${tag}_t/p_kernel_stats.csv $root/gpurun_out/prof_${tag}_fetch/f_counter_collection.csv \
  $root/gpurun_out/prof_${tag}_write/f_counter_collection.csv $n "$script $*" > $root/gpurun_out/${tag}_hbm_table.txt
cp $root/gpurun_out/prof_${tag}_t/p_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv
tail -1 $root/gpurun_out/${tag}_trace.log | cut -c1-400
cat $root/gpurun_out/${tag}_hbm_table.txt
