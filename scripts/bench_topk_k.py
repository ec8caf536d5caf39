import sys, os; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from video_similarity_search_amd.evaluate import cosine_topk
rng = np.random.default_rng(5)
Q = torch.from_numpy(rng.standard_normal((10000, 512)).astype(np.float32)).cuda()
G = torch.from_numpy(rng.standard_normal((100000, 512)).astype(np.float32)).cuda()
for k in (1, 10, 50, 88):
    cosine_topk(Q, G, k=k); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): cosine_topk(Q, G, k=k)
    e1.record(); torch.cuda.synchronize()
    print(k, e0.elapsed_time(e1) / 3, "ms")
