#!/bin/bash
# GPU box: rocprofv3 kernel stats of a short bench run -> gpurun_out/qs/stats.csv  (usage: bash scripts/quick_stats.sh [bench args])
R="$(pwd)"; S="$R/gpurun_out/qs"; rm -rf "$S"; mkdir -p "$S"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$S/stats" -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" > "$S/stats.log" 2>&1 < /dev/null
f=$(find "$S/stats" -name "*kernel_stats.csv" | head -1); cp "$f" "$S/stats.csv"
find "$S/stats" -type f ! -name "*kernel_stats.csv" -delete
grep "^{" "$S/stats.log" | cut -c1-150
