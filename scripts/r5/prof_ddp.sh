#!/bin/bash
# round 5: what DistributedDataParallel costs the training step on ONE rank (bench.py --force-dist: 38.5-39.6 ms against 36.9-37.1 without a
# process group): kernel statistics + GPU busy fraction of the step under a one-rank RCCL group
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ddp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ddp -- python3 $R/bench.py --gpus 1 --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/prof_ddp.log 2>&1
grep -o '"value": [0-9.]*, "unit": "clips/s"' /tmp/prof_ddp.log | head -1
f=$(find /tmp/prof_ddp -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/prof_ddp -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$t" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("kernels not in the plain step's profile, or RCCL / copy kernels:")
for r in rows:
    n = r['Name']
    if any(k in n.lower() for k in ("rccl", "nccl", "copybuffer", "fillbuffer", "elementwise", "multi_tensor", "cat", "allreduce")):
        print(f"  {n[:90]:90s} calls/step {int(r['Calls'])/8:6.1f} avg_us {float(r['AverageNs'])/1e3:8.1f} ms/step {float(r['TotalDurationNs'])/8/1e6:7.3f}")
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[2])))
stems = [i for i, e in enumerate(ev) if e[2].startswith('void conv_gemm_kernel<64, 64, 2, 2>')]
a, b = stems[-3], stems[-2]
step = ev[a:b]; t0, t1 = step[0][0], ev[b][0]
cs, ce = step[0][0], step[0][1]; busy = 0; gaps = []
for s, e, n in step[1:]:
    if s > ce: busy += ce - cs; gaps.append((s - ce, n)); cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"one step: wall {(t1-t0)/1e6:.3f} ms, GPU busy {busy/1e6:.3f} ms, idle {(t1-t0-busy)/1e6:.3f} ms in {len(gaps)} gaps; launches {len(step)}")
for d, n in sorted(gaps, reverse=True)[:8]: print(f"   gap {d/1e3:7.1f} us before {n[:70]}")
PY
