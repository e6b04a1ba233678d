#!/bin/bash
# print "<label> <images/s> <ms/step>" for one bench.py run: bash tools/bench_val.sh <label> [bench.py args...]
label=$1; shift
python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'])"
