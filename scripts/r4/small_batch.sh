#!/bin/bash
# round 4: the Winograd kernels at small batches (few workgroups): one- against two-dimensional, K-split piece counts
#   bash scripts/r4/small_batch.sh "B ..." "shapes" "pieces"
cd "$(dirname "$0")/../.."
for B in ${1:-8}; do
  for sh in ${2:-c4 c7 c10}; do
    for P in ${3:-0 2 4 8 16}; do
      if [ $P = 0 ]; then unset SLIC_WINO2_PIECES; else export SLIC_WINO2_PIECES=$P; fi
      echo "B=$B pieces=$P $sh: $(python scripts/bench_conv.py $B "$sh" 2>&1 | tail -1 | sed 's/.*| wino fwd/wino fwd/')"
    done
  done
done
