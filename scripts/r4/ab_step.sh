#!/bin/bash
# round 4: the training step with the library in the tree against experiment builds (csrc/_exp/libslic_w2_NAME.so), interleaved
#   bash scripts/r4/ab_step.sh "NAME ..." [rounds]
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${2:-2}); do
  for name in base $1; do
    if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/video_similarity_search_amd/csrc/_exp/libslic_w2_$name.so; fi
    python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
