#!/bin/bash
# round 6: the ceiling of a THIRD Winograd axis before anyone builds it.  F(2,3) along T would take the 3x3x3 stride-1 layers from 1/3 to
# 2/9 of the direct form's multiplies.  Diagnostic builds of the one-block kernel (wrong results, right timing; SLIC_WINO2_PERSIST=0):
#   kt2    = the K loop of two of the three kt: the MFMA count of a 3-D kernel at today's L2 -> LDS bytes per MFMA (the ceiling)
#   kt2x2  = the same with every DMA piece fetched twice: a 3-D tile (2 x 2 x 4 outputs from a 4 x 4 x 6 patch, 96 points) holds 32 tiles x 32 columns in
#            the accumulators where today's holds 64 x 64 — 48 + 48 KB of pixels + U per 4-channel stage instead of 24 + 24 for the same 24 MFMAs per wave
#   x2     = all three kt with every piece fetched twice (how much of today's loop is the L2 -> LDS path)
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
bash scripts/r4/ab_wino2.sh build "kt2:-DSLIC_W2_ABL=16384 kt2x2:-DSLIC_W2_ABL=49152 x2:-DSLIC_W2_ABL=32768" 2>&1 | grep -E "error"
export SLIC_WINO2_PERSIST=0
for name in base kt2 kt2x2 x2; do
  if [ $name = base ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_w2_$name.so; fi
  for sh in l1 c7; do echo "$name $sh: $(python scripts/bench_conv.py 32 "$sh" 2>/dev/null | sed 's/.*| wino2 fwd/wino2 fwd/')"; done
done
