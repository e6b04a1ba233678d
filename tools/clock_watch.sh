#!/bin/bash
# usage (on the GPU box): bash tools/clock_watch.sh <tag>
#   runs the default bench for a few hundred steps and samples the shader clock / socket power beside it (rocm-smi, read-only)
#   -> gpurun_out/<tag>_clock_watch.txt : what clock the chip holds under the step (the MFMA peak is quoted at 2.4 GHz)
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_clock_watch.txt
: > $out
timeout -k 10 300 python3 $root/bench.py --steps 400 --warmup 5 --no-cpu-baseline > $root/gpurun_out/${tag}_clock_watch_bench.json 2> $root/gpurun_out/${tag}_clock_watch_bench.err &
pid=$!
for i in $(seq 1 60); do
  kill -0 $pid 2>/dev/null || break
  echo "t=$i" >> $out
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> $out
  sleep 1
done
wait $pid
rc=$?
cat $root/gpurun_out/${tag}_clock_watch_bench.json >> $out
exit $rc
