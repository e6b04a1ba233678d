#!/bin/bash
# usage (on the GPU box): bash tools/tn_tower_ab.sh <tag>   -> gpurun_out/<tag>_tn_tower_ab.txt
#   whole-tower weight-gradient launch with and without the XCD lockstep: time (HIP events) and FETCH_SIZE (own --pmc pass)
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_tn_tower_ab.txt
: > $out
cd /tmp && export TMPDIR=/tmp
for ls in 0 1 0 1; do
  export LC2IS_TN_LOCKSTEP=$ls
  timeout -k 10 240 python3 $root/tools/tn_tower.py --iters 6 --check >> $out 2>&1 || { tail -5 $out; exit 1; }
done
for ls in 0 1; do
  export LC2IS_TN_LOCKSTEP=$ls
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $root/gpurun_out/pmc_${tag}_tower$ls -o f --output-format csv -- python3 $root/tools/tn_tower.py --iters 2 > $root/gpurun_out/pmc_${tag}_tower$ls.log 2>&1 || { tail -5 $root/gpurun_out/pmc_${tag}_tower$ls.log; exit 1; }
  python3 - $root/gpurun_out/pmc_${tag}_tower$ls/f_counter_collection.csv $ls >> $out <<'PY'
import csv, sys
tot = n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "gemm_tn_grouped_tbl_kernel" in r["Kernel_Name"]:
        tot += float(r["Counter_Value"]); n += 1
print(f"lockstep={sys.argv[2]}: FETCH_SIZE {tot / n:.0f} KB/launch over {n} launches = {2e3 * tot / n / 1e9:.2f} GB read per launch (x2: 128-B requests tallied at 64 B on gfx950)")
PY
done
cat $out
