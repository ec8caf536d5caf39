#!/bin/bash
# round 5: topk_select with 4 / 2 / 1 queries (waves) per workgroup — 32 / 16 / 8 KB of LDS: how many sorts a CU holds at once
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
for v in base 2 1; do
  if [ $v = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_tks$v.so; fi
  echo "== waves per workgroup: $v"; bash scripts/r5/prof_topk.sh 2>&1 | grep -E "topk_select|topk_collect" | head -2
done
