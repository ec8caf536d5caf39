#!/bin/bash
# round 6: the persistent data-gradient kernel's optional-operand loads, all eight passes of the second column half in flight at once (default) against
# batches of four (-DSLIC_W2P_LB1=4): the layer1 launches and whole steps interleaved
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
for name in base lb44; do
  if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_w2_$name.so; fi
  echo "== $name"; python scripts/r5/epilogue_parts.py 2>/dev/null | tail -2
done
unset SLIC_LIB_PATH
bash scripts/ab_lib.sh $PWD/$D/_exp/libslic_w2_lb44.so
