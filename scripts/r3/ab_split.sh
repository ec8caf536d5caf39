#!/bin/bash
# Winograd weight gradient: workgroup budget sweep (tile slices = budget / blocks) at layers 1-3 (GPU box)
cd "$(dirname "$0")/../.."
for wgs in 512 768 1024 1536 2048; do
  export SLIC_WINO_WGRAD_WGS=$wgs
  for sh in "c2 l1" "c4 l2" "c7 l3"; do
    echo "wgs $wgs $sh: $(python scripts/bench_conv.py 32 "$sh" 2>/dev/null | sed 's/.*| wino/wino/')"
  done
done
