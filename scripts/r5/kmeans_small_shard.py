"""round 5: what ONE rank of an 8-GPU strong-scaled k-means run does per iteration — the sharded Lloyd iteration on 12 500 of the 100k rows
(K = 500, D = 512) over a one-rank process group (the exchange's wire time is absent: this is the rank-local cost the E-step's 1 / 8 must be
held against).  Also 25k and 50k rows (4 and 2 GPUs).      python scripts/r5/kmeans_small_shard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl")
from video_similarity_search_amd.clustering import KMeans
N, D, K, iters = 100000, 512, 500, 20
rng = np.random.default_rng(1)
X = rng.standard_normal((N, D)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
init = X[rng.choice(N, K, replace=False)].copy()
for rows in (100000, 50000, 25000, 12500):
    Xd = torch.from_numpy(X[:rows]).cuda()
    for ex in ("allreduce", "oneshot", None):
        pg = dist.group.WORLD if ex else None
        km = KMeans(n_clusters=K, init=init, n_init=1, max_iter=iters, tol=0.0, fixed_iters=True, process_group=pg, exchange=ex)
        for _ in range(4): km.fit(Xd)
        ts = []
        for _ in range(3):
            km.fit(Xd); ts.append(km.lloyd_seconds_ / iters * 1e6)
        print(f"rows {rows:6d}  {'unsharded (one fused call)' if ex is None else 'sharded, exchange=' + ex:28s}  {min(ts):7.1f} us / iteration", flush=True)
dist.destroy_process_group()
