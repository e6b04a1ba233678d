#!/bin/bash
# usage (on the GPU box): bash tools/prof_pmc.sh <tag> [commit]  (the box has no .git: pass `git rev-parse --short HEAD` from the build container)   -> gpurun_out/pmc_<tag>_{fetch,write}/ (two separate --pmc passes,
# kernel-trace only, as MI355X_MICROARCH.md prescribes) ; summarise with tools/pmc_sum.py
tag=$1
commit=${2:-unrecorded}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $c --kernel-trace -d $root/gpurun_out/pmc_${tag}_$d -o f --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/pmc_${tag}_$d.log 2>&1 || { tail -5 $root/gpurun_out/pmc_${tag}_$d.log; exit 1; }
done
python3 $root/tools/pmc_sum.py $root/gpurun_out/pmc_${tag}_fetch/f_counter_collection.csv $root/gpurun_out/pmc_${tag}_write/f_counter_collection.csv $root/gpurun_out/pmc_${tag} $commit
