"""where a km_assign_creg wave spends its time (diagnostic build -DKM_STAMPS, SLIC_LIB_PATH=.../libkmst.so)"""
import sys, ctypes; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from video_similarity_search_amd import _lib
from video_similarity_search_amd.clustering.kmeans_hip import HipKernels
hk = HipKernels(); lib = _lib.load()
rng = np.random.default_rng(1)
N, D, K = 100000, 512, 500
X = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32)).cuda()
C = X[:K].clone(); cn = torch.empty(K, device="cuda"); hk.cnorm(C, cn)
lab = torch.empty(N, dtype=torch.int32, device="cuda")
out = (ctypes.c_ulonglong * 8)()
hk.assign_perm(X, C, cn, lab, None, None); torch.cuda.synchronize(); lib.slic_debug_km_counters(out, 1)
hk.assign_perm(X, C, cn, lab, None, None); torch.cuda.synchronize(); lib.slic_debug_km_counters(out, 1)
c = list(out); W = 1024
print("per wave (us at 100 MHz): prologue %.1f  k loops %.1f  epilogues %.1f  total %.1f" % tuple(x / W / 100 for x in c[:4]))
