#!/bin/bash
# does the partly filled last dispatch round cost what the model says?  Winograd forward / data gradient at batch sizes around whole rounds
cd "$(dirname "$0")/../.."
for b in 21 31 32 33 42; do echo "l2 B=$b: $(python scripts/bench_conv.py $b 'c4 l2' 2>/dev/null | sed 's/.*| wino/wino/')"; done
for b in 29 31 32 34; do echo "l1 B=$b: $(python scripts/bench_conv.py $b 'c2 l1' 2>/dev/null | sed 's/.*| wino/wino/')"; done
