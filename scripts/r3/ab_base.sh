#!/bin/bash
# same-box A/B of a conv kernel change: csrc/_exp/libslic_base.so (built from the commit before) against the working tree's library
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export SLIC_LIB_PATH=$PWD/video_similarity_search_amd/csrc/_exp/libslic_base.so; else unset SLIC_LIB_PATH; fi
    for sh in "$@"; do echo "$lib $sh: $(python scripts/bench_conv.py 32 "$sh" 2>/dev/null | sed 's/.*| wino/wino/')"; done
  done
done
