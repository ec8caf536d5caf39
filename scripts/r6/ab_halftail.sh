#!/bin/bash
# round 6: layer2's 3.06 dispatch rounds (784 workgroups) — the persistent kernel's column-half items behind the last whole round (SLIC_WINO2_HALFTAIL=1,
# default) against the K-split tail (16 blocks x 6 pieces + finish pass; =0): per shape and whole steps interleaved
cd "$(dirname "$0")/../.."
for m in 1 0; do
  export SLIC_WINO2_HALFTAIL=$m
  for sh in c4; do echo "halftail=$m $sh: $(python scripts/bench_conv.py 32 "$sh" 2>/dev/null | sed 's/.*| wino2 fwd/wino2 fwd/')"; done
done
unset SLIC_WINO2_HALFTAIL
bash scripts/r6/ab_env_step.sh "SLIC_WINO2_HALFTAIL=0"
