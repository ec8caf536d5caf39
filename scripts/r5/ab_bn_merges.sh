#!/bin/bash
# round 5: (1) what the BatchNorm statistic merges (bn_merge_level / _final, sum_merge_level / _final: ~80 tiny launches per step, most
# of them alone on the forward's chain) cost a STEP — a diagnostic build that does not launch them (stale statistics, right timing);
# (a fused level + final launch with a last-workgroup ticket was measured with the same script: 36.70 ms against 36.71 — not kept)
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
mkdir -p $D/_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DSLIC_BN_DIAG=1 -c $D/bn.hip -o $D/_exp/bn_diag.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/_exp/libslic_bn_diag.so $D/_exp/bn_diag.o $(ls $D/*.o | grep -v "/bn.o") -ldl
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  unset SLIC_LIB_PATH
  echo "tree     $(run)"
  echo "none     $(SLIC_LIB_PATH=$PWD/$D/_exp/libslic_bn_diag.so run)"
done
