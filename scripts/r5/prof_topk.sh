#!/bin/bash
# round 5: per-kernel times of one retrieval call (configs[4], k = 50) on the collect path and on the streaming path
R=$PWD
cat > /tmp/topk50.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch
from video_similarity_search_amd.evaluate import cosine_topk
rng = np.random.default_rng(5)
Q = torch.from_numpy(rng.standard_normal((10000, 512)).astype(np.float32)).cuda()
G = torch.from_numpy(rng.standard_normal((100000, 512)).astype(np.float32)).cuda()
for _ in range(6): cosine_topk(Q, G, k=50)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf /tmp/prof_topk_$mode
  SLIC_TOPK_COLLECT=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_topk_$mode -- python3 /tmp/topk50.py > /dev/null 2>&1
  echo "== SLIC_TOPK_COLLECT=$mode"
  f=$(find /tmp/prof_topk_$mode -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
done
