#!/bin/bash
# round 6: launches of <= 8 tile blocks (layer4 at B = 32) with the (n block, K piece) combinations dealt to the XCDs (SLIC_W2_UMAP=1, default) against the
# tile-block-major order (-DSLIC_W2_UMAP=0): per shape, then whole steps interleaved
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
for name in base noumap; do
  if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_w2_$name.so; fi
  for sh in c10 c7; do echo "$name $sh: $(python scripts/bench_conv.py 32 "$sh" 2>/dev/null | sed 's/.*| wino2 fwd/wino2 fwd/')"; done
done
unset SLIC_LIB_PATH
bash scripts/ab_lib.sh $PWD/$D/_exp/libslic_w2_noumap.so
