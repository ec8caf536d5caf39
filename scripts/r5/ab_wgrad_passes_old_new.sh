#!/bin/bash
# round 5: the two-dimensional weight gradient with Gw^T inside the kernel, grouped slice sums and batched loads in the reduce pass
# (the tree) against the previous form (SLIC_LIB_PATH = a library built from the earlier conv_wino2.hip): per layer alone, and whole steps
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -x -q -k "winograd_2d or chunk or untuned or wgrad" 2>&1 | tail -2
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
