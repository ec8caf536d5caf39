#!/bin/bash
# round 6: the kernels a step gains under DistributedDataParallel at ONE rank — plain step vs bench.py --force-dist with data_parallel (SLIC_DDP_FAST=1) and
# with the plain wrapper (SLIC_DDP_FAST=0): per-step launch counts / time by kernel name where they differ from the plain step, GPU-idle gaps of one step
R=$PWD
cd /tmp && export TMPDIR=/tmp
prof() {   # name, extra args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/prof_$name.log 2>&1
  grep -o '"value": [0-9.]*, "unit": "clips/s"' /tmp/prof_$name.log | head -1
}
prof plain
prof fast --gpus 1 --force-dist
SLIC_DDP_FAST=0 prof slow --gpus 1 --force-dist
python3 - <<'PY'
import csv, glob
def load(name):
    f = glob.glob(f"/tmp/prof_{name}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]) / 8.0, float(r["TotalDurationNs"]) / 8 / 1e6) for r in csv.DictReader(open(f))}
def gaps(name):
    t = glob.glob(f"/tmp/prof_{name}/**/*kernel_trace.csv", recursive=True)[0]
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(t)))
    stems = [i for i, e in enumerate(ev) if e[2].startswith('void conv_gemm_kernel<64, 64, 2, 2>')]
    a, b = stems[-3], stems[-2]
    step = ev[a:b]; t0, t1 = step[0][0], ev[b][0]
    cs, ce = step[0][0], step[0][1]; busy = 0; g = []
    for s, e, n in step[1:]:
        if s > ce: busy += ce - cs; g.append((s - ce, n)); cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    print(f"{name}: one step wall {(t1-t0)/1e6:.3f} ms, GPU busy {busy/1e6:.3f} ms, idle {(t1-t0-busy)/1e6:.3f} ms in {len(g)} gaps; launches {len(step)}")
    for d, n in sorted(g, reverse=True)[:6]: print(f"     gap {d/1e3:7.1f} us before {n[:80]}")
base = load("plain")
for name in ("fast", "slow"):
    cur = load(name)
    print(f"== {name}: kernels whose per-step launch count differs from the plain step (launches/step, ms/step)")
    extra_n = extra_t = 0.0
    for k in sorted(set(cur) | set(base), key=lambda k: -(cur.get(k, (0, 0))[1] - base.get(k, (0, 0))[1])):
        c, b = cur.get(k, (0, 0)), base.get(k, (0, 0))
        if abs(c[0] - b[0]) >= 0.5:
            print(f"   {k[:100]:100s} {b[0]:6.1f} -> {c[0]:6.1f}   {b[1]:7.3f} -> {c[1]:7.3f}")
            extra_n += c[0] - b[0]; extra_t += c[1] - b[1]
    print(f"   extra launches per step {extra_n:.1f}, extra kernel time {extra_t:.3f} ms")
for name in ("plain", "fast", "slow"):
    gaps(name)
PY
