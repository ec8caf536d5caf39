#!/bin/bash
# round 5: the two-dimensional weight gradient with Gw^T inside the kernel, grouped slice sums and batched loads in the reduce pass
# (the tree) against the previous form: per layer alone, and whole steps.
#   bash scripts/r5/ab_wgrad_passes_old_new.sh build     where the git history is (not on the GPU box): builds _exp/libslic_w2_old.so from
#                                                        conv_wino2.hip as of commit 14ec637; the library travels with the snapshot
#   bash scripts/r5/ab_wgrad_passes_old_new.sh           on the GPU box
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $D/_exp
  git show 14ec637:$D/conv_wino2.hip > $D/_exp/conv_wino2_old.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I $D -c $D/_exp/conv_wino2_old.hip -o $D/_exp/w2_old.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/_exp/libslic_w2_old.so $D/_exp/w2_old.o $(ls $D/*.o | grep -v conv_wino2.o) -ldl
  exit
fi
for name in old base; do
  if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_w2_$name.so; fi
  echo "== $name"; python scripts/r5/wgrad_passes.py 2>/dev/null | grep -E "^l[1-4]"
done
for rep in 1 2 3; do
for name in old base; do
  if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_w2_$name.so; fi
  echo "$name $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'])")"
done
done
