#!/bin/bash
# round 6: the B = 8 step (north_star's batch): layer1's 784 blocks = 3.06 rounds end in column-half items (default) or in the K-split tail (SLIC_WINO2_HALFTAIL=0)
cd "$(dirname "$0")/../.."
for i in 1 2 3; do
  for m in 1 0; do
    SLIC_WINO2_HALFTAIL=$m python bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=8 halftail=$m', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
