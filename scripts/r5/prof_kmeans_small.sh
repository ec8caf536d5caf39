#!/bin/bash
# round 5: per-kernel times of the sharded Lloyd iteration on a 12 500-row shard (one rank of an 8-GPU strong-scaled run)
R=$PWD
cat > /tmp/kms.py <<PY
import os, sys
sys.path.insert(0, "$R")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0); dist.init_process_group("nccl")
from video_similarity_search_amd.clustering import KMeans
rows = int(os.environ.get("ROWS", "12500"))
N, D, K, iters = 100000, 512, 500, 20
rng = np.random.default_rng(1)
X = rng.standard_normal((N, D)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
init = X[rng.choice(N, K, replace=False)].copy()
Xd = torch.from_numpy(X[:rows]).cuda()
km = KMeans(n_clusters=K, init=init, n_init=1, max_iter=iters, tol=0.0, fixed_iters=True, process_group=dist.group.WORLD, exchange=os.environ.get("EX", "oneshot"))
for _ in range(6): km.fit(Xd)
torch.cuda.synchronize(); dist.destroy_process_group()
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_kms
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kms -- python3 /tmp/kms.py > /dev/null 2>&1
f=$(find /tmp/prof_kms -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
