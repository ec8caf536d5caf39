#!/bin/bash
# round 5: what the two-dimensional weight gradient's slice-sum and inverse-transform passes (conv_wgrad_wino2_sum / _reduce: 2.1 ms of
# kernel time per step on the side stream) cost a STEP: a diagnostic build without them (wrong gradients, right timing) against the tree
cd "$(dirname "$0")/../.."
bash scripts/r4/ab_wino2.sh build "nopasses:-DSLIC_W2_ABL=4096" 2>&1 | grep -E "error"
D=video_similarity_search_amd/csrc
for rep in 1 2; do
for name in base nopasses; do
  if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_w2_$name.so; fi
  echo "$name $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'])")"
done
done
