#!/bin/bash
# GPU box: rocprofv3 kernel trace of a short bench run, then scripts/step_timeline.py on it  (usage: bash scripts/quick_timeline.sh [bench args])
R="$(pwd)"; S="$R/gpurun_out/qt"; rm -rf "$S"; mkdir -p "$S"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$S/tr" -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary "$@" > "$S/tr.log" 2>&1 < /dev/null
f=$(find "$S/tr" -name "*kernel_trace.csv" | head -1)
grep "^{" "$S/tr.log" | cut -c1-120
python "$R/scripts/step_timeline.py" "$f"
rm -rf "$S/tr"
