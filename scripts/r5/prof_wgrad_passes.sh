#!/bin/bash
# round 5: the weight gradient's three kernels in isolation, per layer (rocprofv3 kernel trace of scripts/r5/wgrad_passes.py)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wgp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_wgp -- python3 "$R/scripts/r5/wgrad_passes.py" 2>/dev/null | grep -E "^l[1-4]"
python3 - <<'P'
import csv,glob,collections
f=glob.glob('/tmp/prof_wgp/**/*kernel_trace.csv',recursive=True)[0]
d=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'wgrad_wino2' not in n: continue
    key=(n.split('(')[0][:40], r['Grid_Size_X'], r['Grid_Size_Y'])
    d.setdefault(key,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    v=sorted(v); print(f"  {k[0]:40s} grid {k[1]:>8s} x {k[2]:>3s}  n {len(v):3d}  median {v[len(v)//2]:8.1f} us")
P
