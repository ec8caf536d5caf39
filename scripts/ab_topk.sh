#!/bin/bash
# same-box timing of cosine top-k (configs[4]) with ablation builds of the library: bash scripts/ab_topk.sh <lib> [<lib> ...]
python scripts/bench_topk_k.py 2>&1 | sed 's/^/cur /'
for l in "$@"; do SLIC_LIB_PATH="$l" python scripts/bench_topk_k.py 2>&1 | sed "s|^|$(basename $l) |"; done
