#!/bin/bash
# diagnostic builds of the Winograd kernels (wrong results, right timing): where does the time go?
#   bash scripts/r3/ab_wino.sh build   (here: cross-compiles csrc/_exp/libslic_abl{1,2,3}.so)
#   bash scripts/r3/ab_wino.sh run     (GPU box: layer1 / layer2 shapes with each library)
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $D/_exp
  for abl in 1 2 3 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DSLIC_WINO_ABL=$abl -c $D/conv.hip -o $D/_exp/conv_abl$abl.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/_exp/libslic_abl$abl.so $D/_exp/conv_abl$abl.o $(ls $D/*.o | grep -v "/conv.o") -ldl
  done
else
  for abl in 0 1 2 3 4; do
    if [ $abl = 0 ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_abl$abl.so; fi
    echo "abl $abl"; python scripts/bench_conv.py 32 "l1" 2>/dev/null | sed 's/.*| wino/wino/'; python scripts/bench_conv.py 32 "c4" 2>/dev/null | sed 's/.*| wino/wino/'
  done
fi
