#!/bin/bash
# GPU box: rocprofv3 kernel stats of any python script -> top kernels   (usage: bash scripts/quick_stats_py.sh scripts/bench_kmeans.py)
R="$(pwd)"; S="$R/gpurun_out/qs2"; rm -rf "$S"; mkdir -p "$S"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$S/stats" -- python "$R/$1" > "$S/stats.log" 2>&1 < /dev/null
f=$(find "$S/stats" -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg us {float(r['AverageNs'])/1e3:9.1f} total ms {float(r['TotalDurationNs'])/1e6:8.2f}")
P
rm -rf "$S/stats"
