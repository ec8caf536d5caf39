#!/bin/bash
# GPU box: per-launch durations (last training step) of the kernels whose name contains $1   (usage: bash scripts/quick_launches.sh bn_)
R="$(pwd)"; S="$R/gpurun_out/ql"; rm -rf "$S"; mkdir -p "$S"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$S/tr" -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > "$S/tr.log" 2>&1 < /dev/null
f=$(find "$S/tr" -name "*kernel_trace.csv" | head -1)
python - "$f" "$1" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))) for r in rows)
marks = [i for i, e in enumerate(ev) if e[2].startswith("ncdhw_to_ndhwc")]
win = ev[marks[-2]:marks[-1]] if len(marks) >= 2 else ev
for s, e, n, g in win:
    if sys.argv[2] in n:
        print(f"{n[:44]:44s} grid {g:9d}  {(e - s) / 1e3:8.1f} us")
P
rm -rf "$S/tr"
