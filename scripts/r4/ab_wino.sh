#!/bin/bash
# round 4: diagnostic builds of the Winograd forward kernel (wrong results, right timing) — the ceiling of a kernel that reads a
# PRE-TRANSFORMED operand (SLIC_WINO_ABL 8: no transform, no validity VALU; 24: + record-contiguous layout; 9 / 25: + no traffic)
#   bash scripts/r4/ab_wino.sh build      (here: cross-compiles csrc/_exp/libslic_abl{8,24,9,25}.so)
#   bash scripts/r4/ab_wino.sh run        (GPU box)
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
ABLS="${ABLS:-8 24 9 25}"
if [ "$1" = build ]; then
  mkdir -p $D/_exp
  for abl in $ABLS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -Wno-unused-variable -DSLIC_WINO_ABL=$abl -c $D/conv.hip -o $D/_exp/conv_abl$abl.o &
  done
  wait
  for abl in $ABLS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/_exp/libslic_abl$abl.so $D/_exp/conv_abl$abl.o $(ls $D/*.o | grep -v "/conv.o") -ldl
  done
else
  for abl in 0 $ABLS; do
    if [ $abl = 0 ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_abl$abl.so; fi
    for sh in l1 c4 c7; do echo "abl $abl $sh: $(python scripts/bench_conv.py 32 "$sh" 2>/dev/null | sed 's/.*| wino/wino/')"; done
  done
fi
