#!/bin/bash
# round 4: kernel trace of the training step at batch B (GPU box): per-kernel ms per step + the timeline's busy fraction
#   bash scripts/r4/trace_step.sh B  ->  gpurun_out/trace_b${B}.txt
cd "$(dirname "$0")/../.."
R=$PWD; B=${1:-8}; O=$R/gpurun_out/tr_b$B; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/log.txt 2>&1 < /dev/null
python - "$O" > $R/gpurun_out/trace_b$B.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['n'] = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
rows.sort(key=lambda r: r['s'])
stems = [i for i, r in enumerate(rows) if r['n'].startswith('conv_gemm_kernel')]
i0, i1 = stems[-2], stems[-1]
step = rows[i0:i1]
t0 = step[0]['s']
ev = sorted((r['s'], r['e']) for r in step)
busy = 0; cs, ce = ev[0]
for s, e in ev[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"step {(rows[i1]['s'] - t0) / 1e6:.3f} ms, {len(step)} kernels, union of kernel intervals {busy / 1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    agg[r['n']][0] += 1; agg[r['n']][1] += (r['e'] - r['s']) / 1e6
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{n:50s} {c:4d} launches {ms:7.3f} ms")
PY
rm -rf $O
cat $R/gpurun_out/trace_b$B.txt
