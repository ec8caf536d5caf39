#!/bin/bash
# round 6: do the HIP-runtime copies / fills of a DistributedDataParallel step at one rank (not torch ops: the profiler sees no aten::copy_ / zero_) scale with
# the number of collectives?  bucket_cap_mb 25 (6 buckets) against 200 (one bucket)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for mb in 25 200; do
  rm -rf /tmp/prof_b$mb
  SLIC_DDP_BUCKET_MB=$mb rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$mb -- python3 $R/bench.py --gpus 1 --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/prof_b$mb.log 2>&1
  echo "bucket_cap_mb=$mb: $(grep -o '"value": [0-9.]*, "unit": "clips/s"' /tmp/prof_b$mb.log | head -1)"
  f=$(find /tmp/prof_b$mb -name "*kernel_stats.csv" | head -1)
  grep -E "copyBuffer|fillBuffer|oneRank|Broadcast|ncclDev" $f | awk -F'","' '{printf "   %-70s calls/step %6.1f  ms/step %7.3f\n", substr($1,2,70), $2/8, $3/8/1e6}'
done
