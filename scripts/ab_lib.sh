#!/bin/bash
# same-box A/B of two builds of the library: bash scripts/ab_lib.sh <other .so>   (prints clips/s of 3 alternating runs each)
for i in 1 2 3; do
  for tag in cur other; do
    if [ $tag = other ]; then export SLIC_LIB_PATH="$1"; else unset SLIC_LIB_PATH; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['ms_per_launch'],3))"
  done
done
