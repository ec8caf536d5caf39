#!/bin/bash
# round 6: the B = 8 training step (north_star's batch) with the persistent kernel (SLIC_WINO2_PERSIST=1, default) and without (=0), interleaved
cd "$(dirname "$0")/../.."
for i in 1 2 3; do
  for m in 0 1; do
    SLIC_WINO2_PERSIST=$m python bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=8 persist=$m', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
