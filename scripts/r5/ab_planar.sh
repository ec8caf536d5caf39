#!/bin/bash
# round 5: what a planar ([row][C/8][W][8]) activation layout could buy conv_wino2_kernel — diagnostic builds (wrong results, right timing)
#   planar   : pixel DMAs as 32 whole 1 KB runs per double stage (SLIC_W2_ABL=512)
#   planar_ne: the same without the epilogue (+32), base_ne: today's kernel without the epilogue (32), nopx: no pixel traffic at all would be abl1
cd "$(dirname "$0")/../.."
bash scripts/r4/ab_wino2.sh build "planar:-DSLIC_W2_ABL=512 planar_ne:-DSLIC_W2_ABL=544 base_ne:-DSLIC_W2_ABL=32 abl1:-DSLIC_W2_ABL=1"
bash scripts/r4/ab_wino2.sh run "planar planar_ne base_ne abl1" "${1:-l1 c4 c7}"
