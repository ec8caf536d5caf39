#!/bin/bash
# round 6: kernels of ONE training step (between two stem launches, rocprofv3 kernel trace) without a process group, under data_parallel and under the
# plain DistributedDataParallel wrapper at one rank: launch counts and kernel time by name where they differ, and the GPU-idle time of the step.
# (A division of a whole run's kernel statistics by its step count — scripts/r5/prof_ddp.sh, round 5's "160 copies + 65 fills per step" — charges the
#  wrapper's ONE-OFF construction to the steps: DistributedDataParallel's parameter / buffer broadcast at construction is ~600 copies and fills.)
R=$PWD
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; rm -rf /tmp/prof_$name; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/prof_$name.log 2>&1; }
prof plain
prof fast --gpus 1 --force-dist
SLIC_DDP_FAST=0 prof slow --gpus 1 --force-dist
python3 - <<'PY'
import csv, glob, re
from collections import Counter
def step(name):
    t = glob.glob(f"/tmp/prof_{name}/**/*kernel_trace.csv", recursive=True)[0]
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(t)))
    stems = [i for i, e in enumerate(ev) if e[2].startswith('void conv_gemm_kernel<64, 64, 2, 2>')]
    out = []
    for a, b in ((stems[4], stems[5]), (stems[5], stems[6])):
        w = ev[a:b]
        cnt, tim = Counter(), Counter()
        for s, e, n in w:
            k = re.sub(r"<.*", "", n.replace("void ", "").split("(")[0])[:70] if "multi_tensor" not in n else ("multi_tensor_apply " + ("add" if "plus" in n else "mul" if "multiplies" in n else "other"))
            cnt[k] += 1; tim[k] += e - s
        cs, ce = w[0][0], w[0][1]; busy = 0
        for s, e, n in w[1:]:
            if s > ce: busy += ce - cs; cs, ce = s, e
            else: ce = max(ce, e)
        busy += ce - cs
        out.append((cnt, tim, (ev[b][0] - w[0][0]) / 1e6, busy / 1e6, len(w)))
    return out
base = step("plain")
print(f"plain: steps of {base[0][2]:.3f} / {base[1][2]:.3f} ms, GPU busy {base[0][3]:.3f} / {base[1][3]:.3f} ms, {base[0][4]} / {base[1][4]} launches")
for name in ("fast", "slow"):
    cur = step(name)
    print(f"{name}: steps of {cur[0][2]:.3f} / {cur[1][2]:.3f} ms, GPU busy {cur[0][3]:.3f} / {cur[1][3]:.3f} ms, {cur[0][4]} / {cur[1][4]} launches")
    c, t = cur[1][0], cur[1][1]
    bc, bt = base[1][0], base[1][1]
    for k in sorted(set(c) | set(bc), key=lambda k: -(t.get(k, 0) - bt.get(k, 0))):
        if c.get(k, 0) != bc.get(k, 0):
            print(f"     {k:72s} {bc.get(k, 0):4d} -> {c.get(k, 0):4d} launches   {bt.get(k, 0) / 1e6:7.3f} -> {t.get(k, 0) / 1e6:7.3f} ms")
PY
