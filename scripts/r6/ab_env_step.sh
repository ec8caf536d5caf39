#!/bin/bash
# whole steps interleaved on one box, default environment against one with extra variables:  bash scripts/r6/ab_env_step.sh "VAR=val VAR2=val2" [reps]
cd "$(dirname "$0")/../.."
for i in $(seq 1 ${2:-3}); do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['ms_per_launch'],3))"
  env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['ms_per_launch'],3))"
done
