#!/bin/bash
# round 5: what the BatchNorm statistics inside conv_wino2_kernel's epilogue cost (diagnostic builds: wrong statistics, right timing)
cd "$(dirname "$0")/../.."
bash scripts/r4/ab_wino2.sh build "nom2:-DSLIC_W2_ABL=1024 nostats:-DSLIC_W2_ABL=2048" 2>&1 | grep -E "error" 
D=video_similarity_search_amd/csrc
for name in base nom2 nostats; do
  if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_w2_$name.so; fi
  echo "== $name"; python scripts/r5/epilogue_parts.py 2>/dev/null | head -2
done
