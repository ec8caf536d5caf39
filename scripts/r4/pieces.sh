#!/bin/bash
# round 4: K-split piece counts of the two-dimensional Winograd launches (layer4: all workgroups cut; layer2: the tail)
cd "$(dirname "$0")/../.."
for sh in "c10" "c4" "c7"; do
  for P in 0 2 3 4 6 8; do
    if [ $P = 0 ]; then unset SLIC_WINO2_PIECES; else export SLIC_WINO2_PIECES=$P; fi
    echo "pieces=$P $sh: $(python scripts/bench_conv.py 32 "$sh" 2>&1 | tail -1 | sed 's/.*| wino fwd/wino fwd/')"
  done
done
