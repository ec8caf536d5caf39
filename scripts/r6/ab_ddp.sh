#!/bin/bash
# round 6: what DistributedDataParallel costs a step at ONE rank (the code path of N = 2, 4, 8 before a byte crosses a link), interleaved on one box:
#   plain     — no process group
#   ddp_fast  — bench.py --force-dist with misc.distributed_helper.data_parallel (flat buffer broadcast, gradients into the bucket views, AVG hook)
#   ddp_plain — bench.py --force-dist with the reference's plain DistributedDataParallel call (SLIC_DDP_FAST=0)
cd "$(dirname "$0")/../.."
run() { "$@" 2>/dev/null | python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('$TAG', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  TAG=plain run python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-secondary
  TAG=ddp_fast run python bench.py --gpus 1 --force-dist --steps 10 --warmup 4 --no-cpu-baseline --no-secondary
  TAG=ddp_plain SLIC_DDP_FAST=0 run env SLIC_DDP_FAST=0 python bench.py --gpus 1 --force-dist --steps 10 --warmup 4 --no-cpu-baseline --no-secondary
done
