"""Counters of the top-k partial kernel (diagnostic build: -DTK_COUNT, SLIC_LIB_PATH=.../libcnt.so)."""
import sys, ctypes; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from video_similarity_search_amd import _lib
from video_similarity_search_amd.evaluate import cosine_topk
rng = np.random.default_rng(5)
Q = torch.from_numpy(rng.standard_normal((10000, 512)).astype(np.float32)).cuda()
G = torch.from_numpy(rng.standard_normal((100000, 512)).astype(np.float32)).cuda()
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
for k in (1, 10, 50, 88):
    cosine_topk(Q, G, k=k); torch.cuda.synchronize()
    lib.slic_debug_topk_counters(out, 1)
    cosine_topk(Q, G, k=k); torch.cuda.synchronize()
    lib.slic_debug_topk_counters(out, 1)
    c = list(out)
    print(k, dict(flushes=c[0], rounds=c[1], groups_entered=c[2], wave_elem_hits=c[3], appended=c[4], inserted=c[5], flush_cycles=c[6], wave_cycles=c[7]))
