#!/bin/bash
# whole steps interleaved on one box: SLIC_WINO2_PERSIST = 0 (one block per workgroup) / 1 (persistent: every eligible launch) / 2 (forward-statistics launches only)
cd "$(dirname "$0")/../.."
for i in 1 2 3; do
  for m in 0 1 2; do
    SLIC_WINO2_PERSIST=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist=$m', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['ms_per_launch'],3))"
  done
done
