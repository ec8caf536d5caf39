#!/bin/bash
# round 6: the persistent form of conv_wino2_kernel (SLIC_WINO2_PERSIST=1 all eligible launches / 2 forward-statistics launches only) against
# the one-block-per-workgroup kernel (=0): per launch at the layer1 shape, per conv shape, whole steps interleaved on one box.
cd "$(dirname "$0")/../.."
for m in 0 1; do
  export SLIC_WINO2_PERSIST=$m
  echo "== SLIC_WINO2_PERSIST=$m"; python scripts/r5/epilogue_parts.py 2>/dev/null
  for sh in l1 c4; do echo "persist=$m $sh: $(python scripts/bench_conv.py 32 "$sh" 2>/dev/null | sed 's/.*| wino2 fwd/wino2 fwd/')"; done
done
for i in 1 2 3; do
  for m in 0 1 2; do
    SLIC_WINO2_PERSIST=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist=$m', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['ms_per_launch'],3))"
  done
done
