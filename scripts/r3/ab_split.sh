#!/bin/bash
# layer4 Winograd weight gradient: tile-slice count sweep (GPU box)
cd "$(dirname "$0")/../.."
export SLIC_WINO_SPLIT=4
for s in 1 2 3 4 5 6 7 8 14; do
  export SLIC_WINO_WGRAD_WGS=$((576 * s))
  echo "l4 wgrad slices $s: $(python scripts/bench_conv.py 32 'c10 l4' 2>/dev/null | sed 's/.*| wino/wino/')"
done
