#!/bin/bash
# config 3 only, no CPU legs: the headline line's kernels_us, N times (A/B of a kernel change: run before and after)
for s in $(seq 1 ${1:-3}); do
  python bench.py --config4 0 --config5 0 --full-solve 0 --cpu-pivots 0 --same-alg-pivots 0 --long-window 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'pivots_per_s':d['value'],'ms_per_step':d['ms_per_step'],'kernels_us':d['kernels_us']}))"
done
