"""E-step (assign_perm + combine) timing vs N — diagnostic for km_assign_creg: tiles per workgroup, startup cost"""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from video_similarity_search_amd.clustering.kmeans_hip import HipKernels
hk = HipKernels()
rng = np.random.default_rng(1)
D, K = 512, 500
for N in (98304, 100000, 106496, 49152, 196608, 8192 * 4):
    X = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32)).cuda()
    C = X[:K].clone()
    cn = torch.empty(K, device="cuda"); hk.cnorm(C, cn)
    lab = torch.empty(N, dtype=torch.int32, device="cuda")
    f = lambda: hk.assign_perm(X, C, cn, lab, None, None)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(N, "tiles/wg %.2f" % (N / 128 / 64), round(e0.elapsed_time(e1) / 10 * 1e3, 1), "us")
