#!/bin/bash
# round 4: diagnostic / alternative builds of the two-dimensional Winograd kernel (csrc/conv_wino2.hip)
#   bash scripts/r4/ab_wino2.sh build "NAME:FLAGS ..."   e.g. "abl1:-DSLIC_W2_ABL=1 notail:-DSLIC_W2_TAIL=0"
#   bash scripts/r4/ab_wino2.sh run "NAME ..." [shapes]
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $D/_exp
  for nf in $2; do
    name=${nf%%:*}; flags=${nf#*:}; flags=${flags//,/ }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include $flags -c $D/conv_wino2.hip -o $D/_exp/w2_$name.o &
  done
  wait
  for nf in $2; do
    name=${nf%%:*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/_exp/libslic_w2_$name.so $D/_exp/w2_$name.o $(ls $D/*.o | grep -v conv_wino2.o) -ldl
  done
else
  for name in base $2; do
    if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_w2_$name.so; fi
    for sh in ${3:-l1 c4 c7}; do echo "$name $sh: $(python scripts/bench_conv.py 32 "$sh" 2>/dev/null | sed 's/.*| wino fwd/wino fwd/')"; done
  done
fi
