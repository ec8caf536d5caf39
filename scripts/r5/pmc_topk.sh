#!/bin/bash
# round 5: matrix-pipe busy fraction of the retrieval kernels (collect pass vs streaming pass) from PMC counters — a separate --pmc pass with
# --kernel-trace only.  busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
R=$PWD
cat > /tmp/topk50.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch
from video_similarity_search_amd.evaluate import cosine_topk
rng = np.random.default_rng(5)
Q = torch.from_numpy(rng.standard_normal((10000, 512)).astype(np.float32)).cuda()
G = torch.from_numpy(rng.standard_normal((100000, 512)).astype(np.float32)).cuda()
for _ in range(4): cosine_topk(Q, G, k=50)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf /tmp/pmc_topk_$mode
  SLIC_TOPK_COLLECT=$mode rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_topk_$mode -- python3 /tmp/topk50.py > /dev/null 2>&1
  f=$(find /tmp/pmc_topk_$mode -name "*counter_collection.csv" | head -1)
  echo "== SLIC_TOPK_COLLECT=$mode"
  python3 - "$f" <<PY
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "topk_collect_qreg" in n or ("topk_partial_qreg" in n and int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) > 100000):
        per[(n[:34], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
agg = collections.defaultdict(list)
for (n, d), c in per.items():
    if c.get("GRBM_GUI_ACTIVE"):
        agg[n].append((c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
for n, v in agg.items():
    print(f"{n:36s} launches {len(v)}  matrix pipe busy {sum(x[0] for x in v)/len(v):.3f}  waves issue-stalled {sum(x[1] for x in v)/len(v):.3f}  issuing {sum(x[2] for x in v)/len(v):.3f}")
PY
done
