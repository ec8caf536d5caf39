#!/bin/bash
# round 5: kernel statistics of the B = 8 training step (north_star's batch), 20 steps
R=$PWD
cat > /tmp/b8.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch, bench
from video_similarity_search_amd.loss import OnlineTripletLoss
model, _ = bench.build_model(); model = model.cuda().train()
crit = OnlineTripletLoss(0.2, 'cosine'); opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.5)
x8 = torch.randn(8, 3, 16, 112, 112, device="cuda"); lab8 = torch.arange(4).repeat(2).cuda()
for _ in range(23):
    l8, _ = crit(model(x8), lab8, sampling_strategy='noise_contrastive'); opt.zero_grad(set_to_none=True); l8.backward(); opt.step()
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -- python3 /tmp/b8.py > /dev/null 2>&1
f=$(find /tmp/prof_b8 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel ms per step {tot/23/1e6:.3f}")
for r in rows[:28]:
    print(f"{r['Name'][:80]:80s} calls/step {int(r['Calls'])/23:6.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} ms/step {float(r['TotalDurationNs'])/23/1e6:7.3f}")
PY
t=$(find /tmp/prof_b8 -name "*kernel_trace.csv" | head -1)
cp "$t" $R/gpurun_out/r5_b8_kernel_trace.csv
