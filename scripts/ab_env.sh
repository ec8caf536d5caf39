#!/bin/bash
# same-box A/B of one environment switch: bash scripts/ab_env.sh VAR=value   (3 alternating runs each)
for i in 1 2 3; do
  for tag in default with; do
    if [ $tag = with ]; then export "$1"; else unset "${1%%=*}"; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1), round(d['ms_per_step'],2))"
  done
done
