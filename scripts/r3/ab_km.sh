#!/bin/bash
# timing variants of the k-means M-step (some give wrong results): bash scripts/r3/ab_km.sh build | run
cd "$(dirname "$0")/../.."
D=video_similarity_search_amd/csrc
VARS=("-DKM_ACC_DEPTH=8" "-DKM_ACC_NOPIPE" "-DKM_ABL_NOSTATUS" "-DKM_ACC_DEPTH=32")
if [ "$1" = build ]; then
  mkdir -p $D/_exp
  (cd $D && make -j8 >/dev/null)
  i=0
  for v in "${VARS[@]}"; do
    i=$((i+1))
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I include $v -c $D/kmeans.hip -o $D/_exp/km_v$i.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/_exp/libslic_km$i.so $D/_exp/km_v$i.o $(ls $D/*.o | grep -v "/kmeans.o") -ldl
  done
else
  export FIT_ONLY=1
  for rep in 1 2; do
  i=0
  unset SLIC_LIB_PATH; echo "default"; python scripts/bench_kmeans.py 2>/dev/null | grep fit
  for v in "${VARS[@]}"; do
    i=$((i+1))
    export SLIC_LIB_PATH=$PWD/$D/_exp/libslic_km$i.so
    echo "$v"; python scripts/bench_kmeans.py 2>/dev/null | grep fit
  done
  done
fi
